"""tools/size_sweep.py [quality] -- BrotliEncoderCompress(quality, 22) on the text generator at sizes from 64 KiB to 64 MiB, host buffers
in and out (what a drop-in caller sees), next to the oracle (liborc_fast.so) on one pinned core of the same host: where the device
starts to pay.  One JSON line per size."""
import json, os, sys, time
os.environ.setdefault("ORC_FAST", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth
import orc
import brotli_mi355x
q = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = brotli_mi355x.default_library()
text = synth.markov_text(64 << 20, 11)
try:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[-1]})
except Exception:
    pass
for kib in (64, 256, 1024, 4096, 16384, 65536):
    d = text[:kib << 10]
    lib.compress(d, q, 22)
    reps = max(2, min(20, (32 << 20) // len(d)))
    t0 = time.time()
    for _ in range(reps):
        got = lib.compress(d, q, 22)
    gpu = (time.time() - t0) / reps
    creps = max(1, min(5, (16 << 20) // len(d)))
    t0 = time.time()
    for _ in range(creps):
        want = orc.compress(d, q, 22)
    cpu = (time.time() - t0) / creps
    print(json.dumps({"quality": q, "KiB": kib, "device_ms": round(gpu * 1e3, 3), "device_MBps": round(len(d) / gpu / 1e6, 1), "oracle_ms": round(cpu * 1e3, 2),
                      "oracle_MBps": round(len(d) / cpu / 1e6, 1), "ratio": round(cpu / gpu, 2), "identical": got == want}), flush=True)
