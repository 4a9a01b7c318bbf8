"""tools/small_trace.py [calls] -- alice29.txt through BrotliEncoderCompress(5, 22), `calls` times after three warm-up calls: the
workload tools/small_trace.sh runs under BROTLI_MI355X_TIMELINE and under rocprofv3 (what one small call is made of)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth
import brotli_mi355x

lib = brotli_mi355x.default_library()
data = synth.alice()
q = int(os.environ.get("SMALL_Q", "5"))
for _ in range(3):
    lib.compress(data, q, 22)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.time()
for _ in range(n):
    lib.compress(data, q, 22)
dt = (time.time() - t0) / n
print("alice29 q%d: %.3f ms per call, %.1f MB/s" % (q, dt * 1e3, len(data) / dt / 1e6))
