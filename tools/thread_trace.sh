ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
python $ROOT/tools/thread_trace.py 1 40
python $ROOT/tools/thread_trace.py 4 40
# four processes side by side
for i in 1 2 3 4; do python $ROOT/tools/thread_trace.py 1 40 & done; wait
for t in 1 4; do
rm -rf $OUT/tt$t
timeout 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $OUT/tt$t -o t -- python $ROOT/tools/thread_trace.py $t 30 > $OUT/tt$t.log 2>&1
tail -1 $OUT/tt$t.log
head -12 $(find $OUT/tt$t -name '*hip_api_stats*' | head -1) | cut -d, -f1-4,6,7
find $OUT/tt$t -name '*trace.csv' -delete
done
