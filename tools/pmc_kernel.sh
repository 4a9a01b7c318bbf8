#!/bin/bash
# tools/pmc_kernel.sh <tag> <kernel substring> -- PMC passes (one counter group per run, no tracing alongside) of a short
# bench run; prints the per-launch maxima of every counter for the named kernel.  Run on the GPU box through gpurun.
TAG=$1; KERNEL=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  rm -rf $OUT/${TAG}_pmc_$N
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$N -o $TAG -- $CMD > $OUT/${TAG}_pmc_$N.log 2>&1
done
python3 - "$OUT" "$TAG" "$KERNEL" <<'PY'
import csv, glob, collections, sys
out, tag, kernel = sys.argv[1:4]
for d in sorted(glob.glob('%s/%s_pmc_*/' % (out, tag))):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, 'max', max(v), 'launches', len(v))
PY
