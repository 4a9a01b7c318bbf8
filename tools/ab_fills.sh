cd $GRAFT_REPO_ROOT
for r in 1 2; do
 for v in 0 1; do
  if [ $v = 1 ]; then export BROTLI_MI355X_DIRECT_FILLS=1; else unset BROTLI_MI355X_DIRECT_FILLS; fi
  A=$(python tools/small_trace.py 40 | tail -1)
  H=$(python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.3f ms, identical %s' % (d['ms_per_step'], d['config']['identical_to_cpu_oracle']))")
  echo "[direct=$v] $A | $H"
 done
done
unset BROTLI_MI355X_DIRECT_FILLS
python tools/small_probe.py
timeout 900 python -m pytest tests/test_cabi.py tests/test_lz77_gpu.py tests/test_quality_2_4.py tests/test_streaming.py -x -q -m gpu 2>&1 | tail -3
