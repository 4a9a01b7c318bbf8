#!/bin/bash
# tools/ab_env.sh VAR v1 v2 ... -- same-box A/B of one environment knob on the headline workload (ms per step, 3 alternating rounds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then R=$(python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"lz77_rounds_per_step": [0-9.]*' | tr '\n' ' ');
    else R=$(env $VAR=$v python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"lz77_rounds_per_step": [0-9.]*' | tr '\n' ' '); fi
    echo "$VAR=$v $R"
  done
done
