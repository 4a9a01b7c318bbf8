#!/usr/bin/env python3
"""tools/summarize_profile.py <tag> <gpurun_out dir> -- condense rocprofv3 csv output (kernel stats + PMC passes made by
tools/profile_round.sh) into one JSON: per-kernel call counts / average durations, and per-kernel average counter values.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950
counts 64 B per 128-B request for wide streaming reads, so both the raw and the doubled figure are given."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("brotli_mi355x::", "")
    name = name.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name  # template instances keep their arguments: k_parse_segments<...,8,false> and <...,8,true> are two rows


def family(name):
    """base name of a template instance (what bench.py's roofline block covers: every launch of every instance)"""
    return short(name).split("<")[0]


def main():
    tag, out = sys.argv[1], sys.argv[2]
    res = {"tag": tag, "kernels": {}, "counters": {}}
    res["families"] = {}
    for f in glob.glob(os.path.join(out, tag + "_kt", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            one = {"calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4),
                   "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])}
            k = short(r["Name"])
            if k in res["kernels"]:  # (same short name from two namespaces: add up instead of overwriting)
                old = res["kernels"][k]
                one = {"calls": old["calls"] + one["calls"], "total_ms": round(old["total_ms"] + one["total_ms"], 3), "pct": old["pct"] + one["pct"]}
                one["avg_ms"] = round(one["total_ms"] / one["calls"], 4)
            res["kernels"][k] = one
            fam = res["families"].setdefault(family(r["Name"]), {"calls": 0, "total_ms": 0.0, "pct": 0.0, "instances": 0})
            fam["calls"] += int(r["Calls"])
            fam["total_ms"] = round(fam["total_ms"] + float(r["TotalDurationNs"]) / 1e6, 3)
            fam["pct"] = round(fam["pct"] + float(r["Percentage"]), 3)
            fam["instances"] += 1
    for fam in res["families"].values():
        fam["avg_ms"] = round(fam["total_ms"] / max(1, fam["calls"]), 4)
    for d in sorted(glob.glob(os.path.join(out, tag + "_pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
            for r in csv.DictReader(open(f)):
                names = {short(r["Kernel_Name"]), family(r["Kernel_Name"])}  # per instance AND per family (all launches)
                for nm in names:
                    a = acc[nm][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
            for k, cs in acc.items():
                for c, (s, n) in cs.items():
                    res["counters"].setdefault(k, {})[c] = {"avg_per_launch": s / n, "launches": n}
    p = res["counters"].get("k_parse_segments", {})
    if "FETCH_SIZE" in p:
        kib = p["FETCH_SIZE"]["avg_per_launch"]
        wr = p.get("WRITE_SIZE", {}).get("avg_per_launch", 0.0)
        res["k_parse_segments_hbm"] = {
            "fetch_bytes_raw": kib * 1024, "fetch_bytes_x2_streaming_correction": kib * 2048, "write_bytes_raw": wr * 1024,
            # the parse kernel's reads are dominated by 4..16-byte gathers, not 16 B/lane streams: the x2 correction of the
            # guide applies to wide coalesced reads only, so the raw figure is the lower bound and x2 the upper bound
            "hbm_bytes_per_launch": kib * 1024 + wr * 1024,
        }
    # FETCH_SIZE + WRITE_SIZE of the WHOLE step: every launch of every kernel of the profiled command (5 steps + 1 warm-up = 6 calls)
    calls = int(os.environ.get("PROFILE_CALLS", "6"))
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    per_kernel = {}
    for k, cs in res["counters"].items():
        if "<" not in k and any(kk.startswith(k + "<") for kk in res["counters"]):
            continue  # (a family row repeats its template instances)
        for c in tot:
            if c in cs:
                b = cs[c]["avg_per_launch"] * cs[c]["launches"] * 1024.0
                tot[c] += b
                per_kernel.setdefault(k, {})[c] = b / calls
    if tot["FETCH_SIZE"] > 0:
        res["step_traffic"] = {"what": "FETCH_SIZE + WRITE_SIZE (raw, KiB -> bytes) summed over every launch of every kernel, per call of the library",
                               "calls": calls, "fetch_bytes_per_step": tot["FETCH_SIZE"] / calls, "write_bytes_per_step": tot["WRITE_SIZE"] / calls,
                               "bytes_per_step": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / calls,
                               "top_kernels_bytes_per_step": dict(sorted(((k, round(v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0))) for k, v in per_kernel.items()),
                                                                         key=lambda kv: -kv[1])[:12])}
    print(json.dumps(res, indent=1))
    # the numbers bench.py quotes as `roofline.traffic` (per launch of the dominant kernel), as a file of their own
    if "FETCH_SIZE" in p:
        c = res["counters"]["k_parse_segments"]

        def avg(name):
            return c.get(name, {}).get("avg_per_launch")
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        pm = {"kernel": "k_parse_segments", "commit": sys.argv[3] if len(sys.argv) > 3 else "?", "source_fingerprint": bench.source_fingerprint(), "kernel_fingerprint": bench.kernel_fingerprint(),
              "source": "rocprofv3 --pmc passes (one counter group per run) of `python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras`, tools/profile_round.sh " + tag,
              "launches": c["FETCH_SIZE"]["launches"], "fetch_kib_per_launch": avg("FETCH_SIZE"), "write_kib_per_launch": avg("WRITE_SIZE"),
              "hbm_bytes_per_launch": res["k_parse_segments_hbm"]["hbm_bytes_per_launch"],
              "note": "FETCH_SIZE / WRITE_SIZE are KiB, raw.  MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request for wide coalesced "
                      "streams (x2 correction); this kernel's reads are candidate gathers of 32 bytes and 64-byte row loads, which are counted "
                      "per 64-byte line, so the raw figure is used and the x2 figure is the upper bound.",
              "tcc_hit_per_launch": avg("TCC_HIT_sum"), "tcc_miss_per_launch": avg("TCC_MISS_sum")}
        if avg("SQ_WAVE_CYCLES"):
            pm["sq_wait_any_over_wave_cycles"] = avg("SQ_WAIT_ANY") / avg("SQ_WAVE_CYCLES")
        if avg("SQ_LDS_IDX_ACTIVE"):
            pm["lds_bank_conflict_cycles_over_lds_active"] = avg("SQ_LDS_BANK_CONFLICT") / avg("SQ_LDS_IDX_ACTIVE")
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVES"):
            if avg(k) is not None:
                pm[k.lower() + "_per_launch"] = avg(k)
        json.dump(pm, open(os.path.join(out, tag + "_pmc_parse.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
