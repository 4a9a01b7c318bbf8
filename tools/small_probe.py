"""SURVEY section 8(d) input C1: alice29.txt through BrotliEncoderCompress(5, 22) -- what one small call costs on the device (host
buffers in and out), one call at a time and several host threads at once; the oracle (liborc_fast.so, one pinned core) beside it."""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth
import orc
import brotli_mi355x

lib = brotli_mi355x.default_library()
data = synth.alice()
want = orc.compress(data, 5, 22)
for _ in range(3):
    got = lib.compress(data, 5, 22)
assert got == want
n = 30
t0 = time.time()
for _ in range(n):
    lib.compress(data, 5, 22)
one = (time.time() - t0) / n
print(json.dumps({"input": "alice29.txt", "bytes": len(data), "ms_per_call": round(one * 1e3, 3), "MBps": round(len(data) / one / 1e6, 1)}), flush=True)
for threads in (4, 16, 64):
    def work():
        for _ in range(6):
            lib.compress(data, 5, 22)
    ts = [threading.Thread(target=work) for _ in range(threads)]
    t0 = time.time()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.time() - t0
    print(json.dumps({"threads": threads, "calls": threads * 6, "aggregate_MBps": round(threads * 6 * len(data) / dt / 1e6, 1), "ms_per_call_per_thread": round(dt / 6 * 1e3, 2)}), flush=True)
t0 = time.time()
for _ in range(5):
    orc.compress(data, 5, 22)
cpu = (time.time() - t0) / 5
print(json.dumps({"cpu_oracle_ms": round(cpu * 1e3, 2), "cpu_MBps": round(len(data) / cpu / 1e6, 1)}))
