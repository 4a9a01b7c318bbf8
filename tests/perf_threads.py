"""Aggregate throughput of independent calls issued from several host threads (one HIP stream each) -- not a test."""
import sys, time, threading
import synth
sys.path.insert(0, "../rust-brotli_amd")
import brotli_mi355x
lib = brotli_mi355x.default_library()
size = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 20)
text = synth.markov_text(size * 8)
parts = [text[i * size:(i + 1) * size] for i in range(8)]
for nthreads in (1, 2, 4, 8, 16):
    reps = 6
    def work(i):
        for _ in range(reps):
            lib.compress(parts[i % 8], 5, 22)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    work(0)  # warm
    t0 = time.time()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.time() - t0
    print("%2d threads x %d calls of %d KiB: %.1f MB/s aggregate, %.2f ms per call per thread" % (nthreads, reps, size >> 10, nthreads * reps * size / dt / 1e6, dt / reps * 1e3), flush=True)
