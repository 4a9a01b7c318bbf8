#!/bin/bash
# tools/ab_small.sh v1 v2 ... -- same-box A/B of library builds ("-" = the default library) on one small call (alice29, quality 5) and the headline step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
libof() { if [ "$1" = "-" ]; then echo $ROOT/rust-brotli_amd/libbrotli_mi355x.so; else echo $ROOT/rust-brotli_amd/libbrotli_mi355x_$1.so; fi; }
for r in 1 2 3; do for v in "$@"; do
  A=$(BROTLI_MI355X_LIB=$(libof $v) python tools/small_trace.py 40 | tail -1)
  H=$(BROTLI_MI355X_LIB=$(libof $v) python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.3f ms, metablock stage %.2f' % (d['ms_per_step'], d['config']['stage_ms_last_step']['metablock']))")
  echo "[$v] $A | $H"
done; done
