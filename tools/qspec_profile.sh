#!/bin/bash
# tools/qspec_profile.sh [MiB] [q] -- rocprofv3 kernel statistics of quality q on text of that size (the speculative quick path), three calls
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; MIB=${1:-64}; Q=${2:-4}
cd /tmp && export TMPDIR=/tmp
D=/tmp/qsprof; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $ROOT/tools/qspec_rounds.py $MIB $Q > $D.log 2>&1
F=$(find $D -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time of two calls: %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print("%-70s calls %5s total %8.2f ms avg %8.1f us  %4.1f%%" % (r['Name'].replace('brotli_mi355x::','')[:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
T=$(find $D -name '*kernel_trace.csv' | head -1)
grep -m1 "k_qs_parse" $T | awk -F, '{print "k_qs_parse: LDS", $12, "scratch", $13, "VGPR", $14, "AGPR", $15, "SGPR", $16}'
