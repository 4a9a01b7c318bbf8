# tools/fuzz_round.sh <seed> -- randomised sweeps against the oracle on the GPU box (not tests): the API sweep at qualities 5-9, 2-4 and 0 / 1, the one-shot sweep
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT/tests; S=${1:-901}
timeout 700 python fuzz_api.py 220 $S > $ROOT/gpurun_out/fuzz_api_$S.log 2>&1; tail -2 $ROOT/gpurun_out/fuzz_api_$S.log
FUZZ_QUICK=1 timeout 700 python fuzz_api.py 300 $((S+1)) > $ROOT/gpurun_out/fuzz_quick_$S.log 2>&1; tail -2 $ROOT/gpurun_out/fuzz_quick_$S.log
FUZZ_FRAGMENT=1 timeout 500 python fuzz_api.py 120 $((S+2)) > $ROOT/gpurun_out/fuzz_frag_$S.log 2>&1; tail -2 $ROOT/gpurun_out/fuzz_frag_$S.log
timeout 700 python fuzz_gpu.py 150 $((S+3)) > $ROOT/gpurun_out/fuzz_gpu_$S.log 2>&1; tail -2 $ROOT/gpurun_out/fuzz_gpu_$S.log
