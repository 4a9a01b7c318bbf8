"""Throughput / round-count probe over several data distributions on the GPU (not a test)."""
import sys, time, ctypes
import numpy as np
import synth, emu, gpulib, orc
L = gpulib.lib()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
check = len(sys.argv) > 2 and sys.argv[2] == "check"
n = mb << 20
rng = np.random.default_rng(5)
def periodic(n, period):
    base = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
    return (base * (n // period + 1))[:n]
cases = {
    "text": lambda: synth.markov_text(n),
    "random": lambda: synth.random_bytes(n),
    "zeros": lambda: bytes(n),
    "mixed": lambda: synth.mixed(n),
    "period7": lambda: periodic(n, 7),
    "period100k": lambda: periodic(n, 100003),
    "lowentropy": lambda: rng.integers(0, 4, n, dtype=np.uint8).tobytes(),
    "text_x4": lambda: (synth.markov_text(n // 4) * 4)[:n],
    "enwik": lambda: synth.enwik_like(n),
    "silesia": lambda: synth.silesia_like(n),
    "binary": lambda: synth.silesia_like(n, only=60),
    "hex": lambda: synth.silesia_like(n, only=85),
    "silesia_fine": lambda: synth.silesia_like(n, min_segment=64 << 10, max_segment=2 << 20),
}
sel = sys.argv[3].split(",") if len(sys.argv) > 3 else list(cases)
for name in sel:
    d = cases[name]()
    best = None
    for it in range(2):
        t = time.time()
        out, st = emu.encode_stream(L, d, [(1, 5), (2, 22), (5, len(d))])
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
    ok = ""
    if check:
        ok = " identical=%s" % (orc.compress(d, 5, 22) == out)
    print("%-12s %4d MiB  %7.1f ms  %7.1f MB/s  rounds %2d  out %9d  lz77 %.1f ms  mb %.1f ms%s" % (name, mb, best * 1e3, len(d) / best / 1e6, st["lz77_rounds"], len(out), st["ms_lz77"], st["ms_metablock"], ok), flush=True)
