#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  rm -rf $OUT/r05l_pmc_$N
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/r05l_pmc_$N -o t -- $CMD > $OUT/r05l_pmc_$N.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in sorted(glob.glob('%s/r05l_pmc_*/' % out)):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'k_parse_lanes' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, [int(x) for x in v])
PY
