# regression subset of the -m gpu suite after the changes to the shared meta-block items (mode of the code jobs, one-block-type split)
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_lz77_gpu.py tests/test_stream_gpu.py tests/test_quality_9_5.py::test_reference_kat_130036_gpu tests/test_quality_9_5.py::test_quality_11_with_q9_5_gpu tests/test_quality_10_11.py::test_reference_kats_47488_46493_gpu tests/test_multi_gpu_plumbing.py tests/test_small_windows.py tests/test_large_window.py -x -q -m gpu --durations=8 > gpurun_out/r04_regress_gpu.log 2>&1
tail -14 gpurun_out/r04_regress_gpu.log
