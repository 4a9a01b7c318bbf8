"""tools/thread_trace.py <threads> <calls> -- alice29 at quality 5 from several host threads (for rocprofv3 --hip-runtime-trace --stats:
what the runtime's entry points cost when threads compete)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth, brotli_mi355x
lib = brotli_mi355x.default_library()
data = synth.alice()
threads, calls = int(sys.argv[1]), int(sys.argv[2])
for _ in range(3):
    lib.compress(data, 5, 22)
def work():
    for _ in range(calls):
        lib.compress(data, 5, 22)
ts = [threading.Thread(target=work) for _ in range(threads)]
t0 = time.time()
for t in ts: t.start()
for t in ts: t.join()
dt = time.time() - t0
print("%d threads x %d calls: %.1f MB/s aggregate, %.2f ms per call per thread" % (threads, calls, threads * calls * len(data) / dt / 1e6, dt / calls * 1e3))
