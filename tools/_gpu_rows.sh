mkdir -p gpurun_out
(BROTLI_MI355X_SELFTEST=1 timeout 600 python tools/rows_gpu_check.py > gpurun_out/r04_rows_selftest.log 2>&1; echo rc=$? >> gpurun_out/r04_rows_selftest.log)
(timeout 300 python tools/prof_inputs.py random silesia binary hex > gpurun_out/r04_rows_inputs.log 2>&1)
(BROTLI_MI355X_NO_POTENTIAL_MASK=1 timeout 300 python tools/prof_inputs.py random silesia > gpurun_out/r04_rows_inputs_nopot.log 2>&1)
(timeout 600 python tools/prof_workload.py c5_xorshift_1GiB_q5 > gpurun_out/r04_c5.log 2>&1)
(timeout 900 python -m pytest tests/test_large_gpu.py -x -q -m gpu -k "xorshift or zero" > gpurun_out/r04_large_c5.log 2>&1)
tail -3 gpurun_out/r04_rows_selftest.log; cat gpurun_out/r04_rows_inputs.log gpurun_out/r04_rows_inputs_nopot.log; tail -2 gpurun_out/r04_c5.log; tail -3 gpurun_out/r04_large_c5.log
