mkdir -p gpurun_out
timeout 35 python -m pytest tests/test_quality_0_1.py::test_catable_streams_dictionaries_and_shards_gpu -x -q -m gpu > gpurun_out/r04_f3_gpu6.log 2>&1
tail -4 gpurun_out/r04_f3_gpu6.log
