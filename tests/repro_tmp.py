import sys, numpy as np, synth, emu, orc
import gpulib
L = gpulib.lib() if len(sys.argv) > 3 else emu.lib()
mb = int(sys.argv[1]); name = sys.argv[2]
n = mb << 20
rng = np.random.default_rng(5)
def periodic(n, period):
    base = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
    return (base * (n // period + 1))[:n]
# keep rng call order identical to perf_types for period100k: period7 first
if name == "period100k":
    periodic(n, 7)
    d = periodic(n, 100003)
elif name == "mixed":
    d = synth.mixed(n)
out, st = emu.encode_stream(L, d, [(1, 5), (2, 22), (5, len(d))])
exp = orc.compress(d, 5, 22)
print(name, mb, len(out), len(exp), out == exp, st["lz77_rounds"])
if out != exp:
    i = next(k for k in range(min(len(out), len(exp))) if out[k] != exp[k])
    print("first diff at", i)
