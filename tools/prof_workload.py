"""tools/prof_workload.py <name> -- two compressions (warm-up + measured) of one large workload on the GPU, for rocprofv3
(tools/profile_workloads.sh).  name: a key of tests/large_cases.py CASES (one-shot cases), or silesia_256MiB_q5.
Prints one JSON line: what the library's own counters say about the measured call."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu  # noqa: E402
import gpulib  # noqa: E402
import large_cases  # noqa: E402
import synth  # noqa: E402

name = sys.argv[1]
if name == "silesia_256MiB_q5":
    data, q, w = synth.silesia_like(256 << 20), 5, 22
else:
    case = large_cases.CASES[name]
    data, q, w = case["make"](), case["quality"], case["lgwin"]
L = gpulib.lib()
work_fn = L.brotli_mi355x_last_parse_work
work_fn.argtypes = [ctypes.POINTER(ctypes.c_double)]
work_fn.restype = None
params = [(1, q), (2, w), (5, min(len(data), 1 << 30))]
emu.encode_stream(L, data, params)  # warm the pools
t = time.time()
out, st = emu.encode_stream(L, data, params)
dt = time.time() - t
work = (ctypes.c_double * 4)()
work_fn(work)
print(json.dumps({"workload": name, "input_bytes": len(data), "compressed_bytes": len(out), "quality": q, "lgwin": w, "ms": round(dt * 1e3, 2),
                  "MB_per_s": round(len(data) / dt / 1e6, 1), "lz77_rounds": st["lz77_rounds"], "ms_lz77": round(st["ms_lz77"], 2),
                  "ms_metablock": round(st["ms_metablock"], 2), "searches_final_parse": st["searches"], "commands": st["commands"],
                  "positions_walked_all_launches": work[0], "searches_all_launches": work[1], "commands_all_launches": work[2]}))
