"""tools/ab_workload.py <case> [VAR=value ...] -- device-resident timing of one large workload of tests/large_cases.py (input in HBM,
output to page-locked host memory, like bench.py's other_workloads), best of 3; environment knobs for a same-box A/B are given
on the command line and applied before the library loads."""
import ctypes, hashlib, json, os, sys, time
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import torch
import brotli_mi355x as bm
from brotli_mi355x import multi
import large_cases
name = sys.argv[1]
frozen = json.load(open(os.path.join(ROOT, "tests", "golden", "large_hashes.json")))
case = large_cases.CASES[name]
data = large_cases.make_input(name, frozen)
lib = bm.default_library()
enc = multi.ShardEncoder(lib.lib, 0)
dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = len(data) + len(data) // 4 + 4096
pinned = torch.empty(cap, dtype=torch.uint8).pin_memory()
enc.use_output_buffer(pinned.data_ptr(), cap, pinned)
params = [(bm.BROTLI_PARAM_QUALITY, case["quality"]), (bm.BROTLI_PARAM_LGWIN, case["lgwin"]), (bm.BROTLI_PARAM_SIZE_HINT, min(len(data), 1 << 30))]
enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=False)
best = None
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    out = enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    best = dt if best is None else min(best, dt)
print(json.dumps({"workload": name, "env": sys.argv[2:], "ms": round(best * 1e3, 1), "MBps": round(len(data) / best / 1e6, 1), "rounds": enc.stats[0],
                  "ms_lz77": round(enc.stats[7], 1), "ms_metablock": round(enc.stats[8], 1),
                  "identical": hashlib.sha256(bytes(out)).hexdigest() == frozen[name]["stream_sha256"]}))
