mkdir -p gpurun_out
timeout 80 python -m pytest tests/test_quality_0_1.py tests/test_fuzz_smoke.py::test_api_sweep_quality_0_1_device tests/test_oneshot_streamed.py -x -q -m gpu --durations=5 > gpurun_out/r04_f3_gpu5.log 2>&1
tail -9 gpurun_out/r04_f3_gpu5.log
