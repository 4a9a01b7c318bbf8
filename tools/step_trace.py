"""tools/step_trace.py <rocprofv3 output dir> [min_us] -- ordered list of the launches and copies of the LAST call in a
rocprofv3 --kernel-trace [--memory-copy-trace] CSV trace (a call starts with k_compute_keys): start, idle gap in front, duration,
name.  Launches shorter than min_us (default 0: list everything) are folded into one line per run of them."""
import csv, glob, os, sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("brotli_mi355x::", "").replace("void ", "")
        name = name.split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_compute_keys")]
if not starts:
    sys.exit("no k_compute_keys launch in the trace")
first = starts[-1]
rows = rows[first:]
t0 = rows[0][0]
prev_end = t0
busy = 0
small_n, small_t, small_from = 0, 0.0, 0.0
def flush():
    global small_n, small_t
    if small_n:
        print("%10.1f %8s %8.1f  (%d short launches / copies)" % (small_from, "", small_t, small_n))
    small_n, small_t = 0, 0.0
totals = {}
for s, e, name in rows:
    dur = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    busy += max(0, e - max(s, prev_end))
    totals[name] = totals.get(name, [0, 0.0])
    totals[name][0] += 1
    totals[name][1] += dur
    if dur < min_us and gap < min_us:
        if small_n == 0:
            small_from = (s - t0) / 1e3
        small_n += 1
        small_t += dur
    else:
        flush()
        print("%10.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, gap, dur, name[:90]))
    prev_end = max(prev_end, e)
flush()
span = (prev_end - t0) / 1e3
print("# span %.1f us, busy %.1f us, %d launches and copies" % (span, busy / 1e3, len(rows)))
print("# totals by name (us):")
for name, (n, t) in sorted(totals.items(), key=lambda kv: -kv[1][1])[:25]:
    print("#   %10.1f  %5d  %s" % (t, n, name[:90]))
