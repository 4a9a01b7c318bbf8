mkdir -p gpurun_out
(BROTLI_MI355X_SELFTEST=1 timeout 600 python tools/rows_gpu_check.py > gpurun_out/r04_rows_selftest.log 2>&1; echo rc=$? >> gpurun_out/r04_rows_selftest.log)
(timeout 300 python tools/prof_inputs.py random silesia binary text zero > gpurun_out/r04_rows_inputs.log 2>&1)
bash tools/_gpu_prof_c5.sh > gpurun_out/r04_c5_prof.log 2>&1
bash tools/_gpu_prof_c5.sh silesia_256MiB_q5 > gpurun_out/r04_sil_prof.log 2>&1
tail -2 gpurun_out/r04_rows_selftest.log; cat gpurun_out/r04_rows_inputs.log; grep MB_per gpurun_out/r04_c5_xorshift_1GiB_q5_kt.log gpurun_out/r04_silesia_256MiB_q5_kt.log
