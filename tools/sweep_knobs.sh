#!/bin/bash
# tools/sweep_knobs.sh -- headline bench under a few settings of the tuning knobs (warm-up length, segment size); run on
# the GPU box through gpurun.  Prints "setting value ms_per_step rounds".
cd "$(dirname "$0")/.."
run() {
  local tag="$1"; shift
  timeout 120 env "$@" python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['config'].get('lz77_rounds_per_step'))"
}
EXTRA=""
for w in 192 256 320 384 512; do run "warmup=$w" BROTLI_MI355X_WARMUP=$w; done
for s in 1024 1536 3072 4096; do EXTRA="--segment-bytes $s" run "segment=$s" X=1; done
