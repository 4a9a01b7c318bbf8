mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
head -c 600 gpurun_out/r04_bench_final.json; echo; tail -3 gpurun_out/r04_bench_final.err
