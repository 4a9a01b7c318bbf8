#!/bin/bash
# tools/ab_parse.sh [--pmc] [--check] v1 v2 ... -- same-box A/B of library builds (rust-brotli_amd/libbrotli_mi355x_<v>.so; "-" = the default
# library) on the headline workload: ms per step (3 alternating rounds), the parse kernel's HIP-event time per step, and with --pmc
# SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INST_CYCLES_SALU per launch of k_parse_segments; --check runs tests/test_lz77_gpu.py on each.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out; mkdir -p $OUT
PMC=0; CHECK=0
while [ "${1:0:2}" = "--" ]; do [ "$1" = "--pmc" ] && PMC=1; [ "$1" = "--check" ] && CHECK=1; shift; done
libof() { if [ "$1" = "-" ]; then echo $ROOT/rust-brotli_amd/libbrotli_mi355x.so; else echo $ROOT/rust-brotli_amd/libbrotli_mi355x_$1.so; fi; }
B="python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
for r in 1 2 3; do
  for v in "$@"; do
    R=$(BROTLI_MI355X_LIB=$(libof $v) timeout 200 $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step %.3f parse_ms_per_step %.3f launches %.1f frac %.5f lz77 %.2f mb %.2f' % (d['ms_per_step'], r['avg_launch_ms']*r['launches_per_step'], r['launches_per_step'], r['frac'], d['config']['stage_ms_last_step']['lz77'], d['config']['stage_ms_last_step']['metablock']))")
    echo "[$v] $R"
  done
done
if [ $CHECK = 1 ]; then
  for v in "$@"; do
    echo "== check [$v]"; BROTLI_MI355X_LIB=$(libof $v) timeout 900 python -m pytest tests/test_lz77_gpu.py tests/test_stream_gpu.py -x -q -m gpu 2>&1 | tail -2
  done
fi
if [ $PMC = 1 ]; then
  cd /tmp && export TMPDIR=/tmp
  for v in "$@"; do
    for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
      N=$(echo $C | tr ' ' '_'); D=$OUT/ab_${v}_$N; rm -rf $D
      BROTLI_MI355X_LIB=$(libof $v) timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -o ab -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D.log 2>&1
      python3 - "$D" "$v" <<'PY'
import csv, glob, collections, sys
d, v = sys.argv[1:3]
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'k_parse_segments' in r['Kernel_Name']:
            acc['all'][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print('[%s] per step:' % v, {c: round(sum(x) / 3.0 / 1e6, 1) for c, x in cs.items()}, '(M, 3 calls averaged; launches %d)' % len(next(iter(cs.values()))))
PY
      rm -rf $D
    done
  done
fi
