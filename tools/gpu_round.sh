#!/bin/bash
# tools/gpu_round.sh <what> [tag] -- what the builder runs on the GPU box through gpurun; everything goes to gpurun_out/<tag>_*.
#   probe     headline workload: timing, host-side timeline (BROTLI_MI355X_TIMELINE), launch list
#   trace     ordered rocprofv3 kernel + copy trace of three steps of the headline workload (where the gaps are)
#   wtrace    ordered kernel trace of one large workload's last call (tools/step_trace.py)
#   q01       qualities 0 / 1: -m gpu tests of test_quality_0_1.py and tools/q01_probe.py
#   suite     the whole -m gpu suite
#   bench     python bench.py with the driver's defaults
#   profile   tools/profile_round.sh (kernel stats + PMC passes of the headline workload)
# (one script instead of one file per call; the committed summaries live under profiles/)
WHAT=${1:-probe}; TAG=${2:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
case $WHAT in
  probe)
    python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/${TAG}_probe_bench.json 2> $OUT/${TAG}_probe_bench.err
    BROTLI_MI355X_TIMELINE=1 BROTLI_MI355X_PROFILE=1 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_probe_timeline.err
    grep -o '"ms_per_step": [0-9.]*' $OUT/${TAG}_probe_bench.json; grep timeline $OUT/${TAG}_probe_timeline.err | tail -1 | cut -c1-1200;;
  trace)
    cd /tmp && export TMPDIR=/tmp
    rm -rf $OUT/${TAG}_trace
    timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_trace.log 2>&1
    tail -2 $OUT/${TAG}_trace.log | cut -c1-300;;
  wtrace)  # wtrace <tag> <case>: ordered kernel trace of the last call of one large workload (tools/ab_workload.py)
    CASE=${3:-c5_xorshift_1GiB_q5}
    cd /tmp && export TMPDIR=/tmp
    rm -rf $OUT/${TAG}_wtrace
    if [ "$CASE" = "silesia_256MiB_q5" ]; then RUN="python $ROOT/tools/prof_workload.py $CASE"; else RUN="python $ROOT/tools/ab_workload.py $CASE"; fi
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_wtrace -o t -- $RUN > $OUT/${TAG}_wtrace.log 2>&1
    python $ROOT/tools/step_trace.py $OUT/${TAG}_wtrace 300 > $OUT/${TAG}_wtrace_${CASE}.txt; rm -rf $OUT/${TAG}_wtrace
    tail -3 $OUT/${TAG}_wtrace.log | cut -c1-300; tail -30 $OUT/${TAG}_wtrace_${CASE}.txt;;
  q01)
    timeout 900 python -m pytest tests/test_quality_0_1.py -x -q -m gpu > $OUT/${TAG}_q01_tests.log 2>&1; tail -3 $OUT/${TAG}_q01_tests.log
    timeout 600 python tools/q01_probe.py > $OUT/${TAG}_q01_probe.jsonl 2> $OUT/${TAG}_q01_probe.err; cat $OUT/${TAG}_q01_probe.jsonl; tail -3 $OUT/${TAG}_q01_probe.err;;
  suite)
    timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/${TAG}_gpu_suite.log 2>&1; tail -5 $OUT/${TAG}_gpu_suite.log;;
  bench)
    timeout 1500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; head -c 700 $OUT/${TAG}_bench.json; echo; tail -3 $OUT/${TAG}_bench.err;;
  profile)
    bash tools/profile_round.sh $TAG $(git -C $ROOT rev-parse --short HEAD 2>/dev/null || echo unknown); tail -5 $OUT/${TAG}_summary.err;;
  *) echo "unknown: $WHAT"; exit 2;;
esac
