#!/bin/bash
# A/B of experiment builds (make -C rust-brotli_amd variant NAME=... EXTRA=...): bench.py headline for the default
# library and for every variant named on the command line.  Run on the GPU box: gpurun -- bash tools/ab_variants.sh w64 occ6
cd "$(dirname "$0")/.."
B="python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
for v in "" "$@"; do
  if [ -n "$v" ]; then export BROTLI_MI355X_LIB=$PWD/rust-brotli_amd/libbrotli_mi355x_$v.so; else unset BROTLI_MI355X_LIB; fi
  echo "== variant [$v]"
  timeout 120 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
