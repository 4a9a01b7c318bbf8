cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
NAME=${1:-c5_xorshift_1GiB_q5}
rm -rf $OUT/r04_${NAME}_kt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_${NAME}_kt -o r04 -- python $ROOT/tools/prof_workload.py $NAME > $OUT/r04_${NAME}_kt.log 2>&1
cp $(find $OUT/r04_${NAME}_kt -name "*kernel_stats.csv" | head -1) $OUT/r04_${NAME}_kernel_stats.csv
rm -rf $OUT/r04_${NAME}_kt
head -25 $OUT/r04_${NAME}_kernel_stats.csv | cut -c1-160
tail -2 $OUT/r04_${NAME}_kt.log
