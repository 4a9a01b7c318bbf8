import sys, numpy as np, synth, emu, orc, gpulib
import cmp_lz77
L = gpulib.lib()
n = 1 << 20
rng = np.random.default_rng(5)
def periodic(n, period):
    base = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
    return (base * (n // period + 1))[:n]
periodic(n, 7)
d = periodic(n, 100003)
cmp_lz77.check("period100k", d, lib=L)
cmp_lz77.check("period100k-300k", d[:300000], lib=L)
