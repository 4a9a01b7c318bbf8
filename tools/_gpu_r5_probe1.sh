#!/bin/bash
# round 5, first look: headline timing, host-side timeline and per-launch parse durations of one step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r05_p1_bench.json 2> gpurun_out/r05_p1_bench.err
BROTLI_MI355X_TIMELINE=1 BROTLI_MI355X_PROFILE=1 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r05_p1_timeline.json 2> gpurun_out/r05_p1_timeline.err
BROTLI_MI355X_DEBUG_LAUNCH=1 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> gpurun_out/r05_p1_launch.err
head -c 400 gpurun_out/r05_p1_bench.json; echo
tail -5 gpurun_out/r05_p1_timeline.err
