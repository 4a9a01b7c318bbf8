"""Throughput over qualities / window sizes / input sizes on the GPU (not a test)."""
import sys, time
import synth, emu, gpulib, orc
L = gpulib.lib()
text = synth.markov_text(64 << 20)
for q, w, n in [(5, 22, 64 << 20), (6, 22, 64 << 20), (7, 22, 64 << 20), (8, 22, 64 << 20), (9, 22, 64 << 20), (5, 24, 64 << 20), (5, 18, 64 << 20),
                (5, 22, 3 << 20), (5, 22, 1 << 20), (5, 22, 152089)]:
    d = text[:n]
    best = None
    for it in range(3):
        t = time.time()
        out, st = emu.encode_stream(L, d, [(1, q), (2, w), (5, len(d))])
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
    ok = ""
    if n <= (3 << 20) or q >= 8:
        ok = " identical=%s" % (orc.compress(d, q, w) == out)
    print("q%d w%d %9d B  %7.1f MB/s  rounds %2d  lz77 %.1f ms  mb %.1f ms  total %.1f ms%s" % (q, w, n, n / best / 1e6, st["lz77_rounds"], st["ms_lz77"], st["ms_metablock"], st["ms_total"], ok), flush=True)
