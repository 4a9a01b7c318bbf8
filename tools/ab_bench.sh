#!/bin/bash
# same-box A/B of two builds of the library: put them in gpurun_ab/lib_old.so and gpurun_ab/lib_new.so, then
# gpurun -- "bash tools/ab_bench.sh > gpurun_out/ab.log" (box-to-box variance is larger than most single changes)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in old new; do
    cp gpurun_ab/lib_$v.so rust-brotli_amd/libbrotli_mi355x.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | sed "s/^/$v /"
  done
done
