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
    if name.startswith("k_parse_segments<"):
        name = "k_parse_segments"  # the two instantiations (H9 / the others) never run in the same call
    return name


def main():
    tag, out = sys.argv[1], sys.argv[2]
    res = {"tag": tag, "kernels": {}, "counters": {}}
    for f in glob.glob(os.path.join(out, tag + "_kt", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            res["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4),
                                                 "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])}
    for d in sorted(glob.glob(os.path.join(out, tag + "_pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
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
