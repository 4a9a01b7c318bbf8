# qualities 0..4 (row f3) on the device for the first time: the new -m gpu tests, the C ABI suite, then a speed probe
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_quality_2_4.py::test_streamed_in_pieces_gpu tests/test_quality_0_1.py tests/test_fuzz_smoke.py::test_api_sweep_quality_2_4_device tests/test_fuzz_smoke.py::test_api_sweep_quality_0_1_device -x -q -m gpu --durations=12 > gpurun_out/r04_f3_gpu2.log 2>&1
tail -22 gpurun_out/r04_f3_gpu2.log
