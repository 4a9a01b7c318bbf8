#!/bin/bash
# tools/f3_pmc.sh -- what bounds the single-wavefront kernels of qualities 0..4: two PMC passes (one counter group per run, no
# tracing alongside) of 512 KiB of text at qualities 0, 1 and 2; per kernel the sums over its launches.  Through gpurun.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/f3_mini.py <<PY
import sys
sys.path.insert(0, "$ROOT/tests"); sys.path.insert(0, "$ROOT/rust-brotli_amd")
import torch  # noqa
import synth, test_cabi
lib = test_cabi._load("gpu")
d = synth.markov_text(1 << 19)
for q in (0, 1, 2):
    lib.compress(d, q, 22)
PY
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  rm -rf $OUT/r04_f3_pmc_$N
  timeout 45 rocprofv3 --pmc $C --output-format csv -d $OUT/r04_f3_pmc_$N -o r04 -- python /tmp/f3_mini.py > $OUT/r04_f3_pmc_$N.log 2>&1
done
python3 - "$OUT" <<'PY' > $OUT/r04_f3_pmc.json
import csv, glob, collections, json, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(out + '/r04_f3_pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('brotli_mi355x::', '').split('(')[0]
        if name in ('k_quick_block', 'k_fragment'):
            a = acc[(name, r['Counter_Name'])]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    for (name, c), (v, n) in acc.items():
        res[name][c] = {"sum": v, "launches": n}
print(json.dumps({"input": "512 KiB of text at qualities 0, 1 (k_fragment) and 2 (k_quick_block), one BrotliEncoderCompress each", "counters": res}, indent=1))
PY
rm -rf $OUT/r04_f3_pmc_SQ_*
cat $OUT/r04_f3_pmc.json | head -60
