mkdir -p gpurun_out
bash tools/profile_round.sh r04 ff8a0f9 > gpurun_out/r04_profile_round.log 2>&1
cp gpurun_out/r04_pmc_parse.json profiles/ 2>/dev/null
bash tools/profile_workloads.sh r04 c5_xorshift_1GiB_q5 c3_enwik_256MiB_q9 silesia_256MiB_q5 > gpurun_out/r04_profile_workloads.log 2>&1
cp gpurun_out/r04_c5_xorshift_1GiB_q5.json gpurun_out/r04_c3_enwik_256MiB_q9.json gpurun_out/r04_silesia_256MiB_q5.json profiles/ 2>/dev/null
rm -rf gpurun_out/r04_*_kt gpurun_out/r04_*_pmc_FETCH_SIZE gpurun_out/r04_*_pmc_WRITE_SIZE gpurun_out/r04_pmc_* gpurun_out/r04_kt
timeout 600 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
timeout 400 python -m pytest tests/test_quality_10_11.py tests/test_fuzz_smoke.py -x -q -m gpu -k "10_11" --durations=6 > gpurun_out/r04_q10_11_gpu3.log 2>&1
tail -3 gpurun_out/r04_profile_round.log; tail -3 gpurun_out/r04_profile_workloads.log; head -c 400 gpurun_out/r04_bench.json; echo; tail -8 gpurun_out/r04_q10_11_gpu3.log
