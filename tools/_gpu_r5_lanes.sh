#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 12288 4000000000; do
  BROTLI_MI355X_LANES_MIN=$m BROTLI_MI355X_TIMELINE=1 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_lanes_$m.json 2> gpurun_out/r05_lanes_$m.err
  echo "LANES_MIN=$m"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05_lanes_$m.json; grep -o '"identical_to_cpu_oracle": [a-z]*' gpurun_out/r05_lanes_$m.json | head -1; grep -o '"roofline": {[^}]*}' gpurun_out/r05_lanes_$m.json | cut -c1-400
  grep timeline gpurun_out/r05_lanes_$m.err | tail -1 | cut -c1-900
done
timeout 600 python -m pytest tests/test_lz77_gpu.py -x -q -m gpu 2>&1 | tail -3
