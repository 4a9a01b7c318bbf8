#!/bin/bash
# tools/ab_kernel.sh <kernel substring> v1 v2 ... -- average duration of the kernels matching the substring under rocprofv3, per library build, headline workload
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; K=$1; shift
libof() { if [ "$1" = "-" ]; then echo $ROOT/rust-brotli_amd/libbrotli_mi355x.so; else echo $ROOT/rust-brotli_amd/libbrotli_mi355x_$1.so; fi; }
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  D=/tmp/abk_$v; rm -rf $D
  BROTLI_MI355X_LIB=$(libof $v) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $D.log 2>&1
  F=$(find $D -name '*kernel_stats.csv' | head -1)
  echo "[$v]"; grep -i "$K" $F | cut -c1-200 | head -6
done
