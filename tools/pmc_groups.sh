#!/bin/bash
# tools/pmc_groups.sh -- PMC passes of the headline workload, per parse kernel (k_parse_groups / k_parse_segments), totals per step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC_SETS=1: the first counter set only
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS")
for C in "${SETS[@]:0:${PMC_SETS:-4}}"; do
  N=$(echo $C | tr ' ' '_'); D=$OUT/pg_$N; rm -rf $D
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -o pg -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D.log 2>&1
  python3 - "$D" <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_parse' in n:
            acc[n.split('(')[0].replace('void brotli_mi355x::', '')[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k, {c: round(sum(x) / 3.0 / 1e6, 2) for c, x in cs.items()}, 'M per step; launches', len(next(iter(cs.values()))))
PY
  rm -rf $D
done
