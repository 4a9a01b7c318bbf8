#!/bin/bash
# tools/profile_round.sh <tag> -- run on the GPU box (through gpurun): kernel trace + separate PMC passes of the
# bench command; raw output under gpurun_out/<tag>_*; tools/summarize_profile.py turns it into profiles/.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
CMD="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pmc_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o $TAG -- $CMD > $OUT/${TAG}_kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"; do
  N=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$N -o $TAG -- $CMD > $OUT/${TAG}_pmc_$N.log 2>&1
done
python $ROOT/tools/summarize_profile.py $TAG $OUT ${2:-unknown} > $OUT/${TAG}_summary.json 2> $OUT/${TAG}_summary.err
cp $(find $OUT/${TAG}_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
tail -1 $OUT/${TAG}_kt.log
