#!/bin/bash
# tools/profile_workloads.sh <tag> [names...] -- run on the GPU box (through gpurun): for every workload a kernel trace and the
# FETCH_SIZE / WRITE_SIZE passes (one counter per run, nothing traced alongside) of `python tools/prof_workload.py <name>`.
# Leaves gpurun_out/<tag>_<name>_kernel_stats.csv and gpurun_out/<tag>_<name>.json (library counters + per-kernel HBM traffic).
set -u
TAG=${1:-r04}
shift
NAMES=${@:-"c3_enwik_256MiB_q9 silesia_256MiB_q5 c5_xorshift_1GiB_q5"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for NAME in $NAMES; do
  CMD="python $ROOT/tools/prof_workload.py $NAME"
  rm -rf $OUT/${TAG}_${NAME}_kt $OUT/${TAG}_${NAME}_pmc_*
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${NAME}_kt -o $TAG -- $CMD > $OUT/${TAG}_${NAME}_kt.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_${NAME}_pmc_$C -o $TAG -- $CMD > $OUT/${TAG}_${NAME}_pmc_$C.log 2>&1
  done
  cp $(find $OUT/${TAG}_${NAME}_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${NAME}_kernel_stats.csv 2>/dev/null
  python3 - "$OUT" "$TAG" "$NAME" <<'PY'
import csv, glob, json, os, sys, collections
out, tag, name = sys.argv[1:4]
sys.path.insert(0, os.path.dirname(out.rstrip("/")))
import bench
res = {"workload": name, "source_fingerprint": bench.source_fingerprint(), "kernel_fingerprint": bench.kernel_fingerprint()}
for line in open(os.path.join(out, "%s_%s_kt.log" % (tag, name))):
    if line.startswith("{"):
        res["library"] = json.loads(line)
kern = {}
for f in glob.glob(os.path.join(out, "%s_%s_kt" % (tag, name), "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kern[r["Name"].replace("brotli_mi355x::", "").split("(")[0].replace("void ", "")] = {
            "calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])}
res["kernels"] = dict(sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"])[:12])
traffic = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "%s_%s_pmc_%s" % (tag, name, c), "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"].replace("brotli_mi355x::", "").split("(")[0].replace("void ", "")]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (s, n) in acc.items():
            traffic[k][c] = {"kib_total_both_calls": s, "launches": n}
res["hbm_counters_kib_raw"] = {k: traffic[k] for k in res["kernels"] if k in traffic}
json.dump(res, open(os.path.join(out, "%s_%s.json" % (tag, name)), "w"), indent=1)
print(name, res.get("library", {}).get("MB_per_s"), list(res["kernels"].items())[:3])
PY
done
