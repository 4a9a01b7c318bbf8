#!/bin/bash
# ordered kernel trace of a few steps of the headline workload: where the gaps are
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r05_trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/r05_trace -o t -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r05_trace.log 2>&1
tail -2 $OUT/r05_trace.log | cut -c1-300
find $OUT/r05_trace -name "*.csv" | head; du -sh $OUT/r05_trace
