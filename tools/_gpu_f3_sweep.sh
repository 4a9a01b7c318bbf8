# wider randomised sweeps of qualities 0..4 on the device (tests/fuzz_api.py), bounded
mkdir -p gpurun_out
cd tests
(FUZZ_QUICK=1 FUZZ_MAXN=400000 timeout 100 python fuzz_api.py 120 41; FUZZ_QUICK=1 FUZZ_TINY=1 timeout 40 python fuzz_api.py 150 42;
 FUZZ_FRAGMENT=1 FUZZ_MAXN=400000 timeout 100 python fuzz_api.py 120 43; FUZZ_FRAGMENT=1 FUZZ_TINY=1 timeout 40 python fuzz_api.py 150 44) 2>&1 | grep -v "reference encoder fails" > ../gpurun_out/r04_f3_sweep.log
tail -12 ../gpurun_out/r04_f3_sweep.log
