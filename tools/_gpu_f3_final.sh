# qualities 0..4 on the device: the larger tests, the torch plumbing test that failed once on a box, kernel statistics of the probe
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_quality_2_4.py::test_past_the_ring_and_shards_gpu tests/test_quality_2_4.py::test_identity_with_the_oracle_gpu tests/test_quality_0_1.py::test_several_fragments_gpu tests/test_multi_gpu_plumbing.py -x -q -m gpu --durations=6 > gpurun_out/r04_f3_gpu3.log 2>&1
tail -9 gpurun_out/r04_f3_gpu3.log
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/r04_f3_kt
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_f3_kt -o r04 -- python $ROOT/tools/f3_probe.py > $ROOT/gpurun_out/r04_f3_probe.jsonl 2> $ROOT/gpurun_out/r04_f3_kt.err
cp $(find $ROOT/gpurun_out/r04_f3_kt -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r04_f3_kernel_stats.csv
rm -rf $ROOT/gpurun_out/r04_f3_kt
head -8 $ROOT/gpurun_out/r04_f3_kernel_stats.csv | cut -c1-120
grep -c identical_to_cpu_oracle...true $ROOT/gpurun_out/r04_f3_probe.jsonl
