#!/bin/bash
# tools/small_trace.sh [tag] -- what a 152 KB call is made of: host timeline, ordered kernel + copy trace, HIP API statistics.
TAG=${1:-r06s}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/small_trace.py 30 > $OUT/${TAG}_small.txt 2>&1
BROTLI_MI355X_TIMELINE=1 python $ROOT/tools/small_trace.py 2 2> $OUT/${TAG}_small_timeline.err >> $OUT/${TAG}_small.txt
rm -rf $OUT/${TAG}_small_trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d $OUT/${TAG}_small_trace -o t -- python $ROOT/tools/small_trace.py 10 > $OUT/${TAG}_small_trace.log 2>&1
cat $OUT/${TAG}_small.txt; tail -2 $OUT/${TAG}_small_timeline.err | cut -c1-3000
find $OUT/${TAG}_small_trace -name '*stats*' | head; 
