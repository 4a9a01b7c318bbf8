"""concurrent BrotliEncoderCompress calls from many host threads (each on its own HIP stream): identity, aggregate rate, and a
clean exit (thread-local pools are torn down when the threads end)"""
import faulthandler, json, os, sys, threading, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth
import orc
import brotli_mi355x
lib = brotli_mi355x.default_library()
q = int(sys.argv[1]) if len(sys.argv) > 1 else 5
data = synth.alice() if len(sys.argv) < 3 else synth.markov_text(int(sys.argv[2]), 9)
want = orc.compress(data, q, 22)
assert lib.compress(data, q, 22) == want
bad = []
def work(n):
    for _ in range(n):
        if lib.compress(data, q, 22) != want:
            bad.append(1)
for threads in (2, 8, 32):
    ts = [threading.Thread(target=work, args=(4,)) for _ in range(threads)]
    t0 = time.time()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.time() - t0
    print(json.dumps({"quality": q, "bytes": len(data), "threads": threads, "aggregate_MBps": round(threads * 4 * len(data) / dt / 1e6, 1), "bad": len(bad)}), flush=True)
print("threads done", flush=True)
