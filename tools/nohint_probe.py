"""tools/nohint_probe.py [MiB] -- the CompressorWriter pattern (4 KiB writes, no size hint, quality 5, lgwin 22) on text: where the time of the
live chain goes (BROTLI_MI355X_PROFILE=1 BROTLI_MI355X_DEBUG=1 for the library's own stage timers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth, brotli_mi355x
lib = brotli_mi355x.default_library()
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) << 20
data = synth.markov_text(n, 5)
lib.compress(data[:1 << 16], 5, 22)
t = time.time()
enc = brotli_mi355x.Encoder(lib, [(1, 5), (2, 22)])
for i in range(0, n, 4096):
    enc.write(data[i:i + 4096])
out = enc.finish()
print("writer pattern %d MiB: %.0f ms (%.1f MB/s), %d bytes" % (n >> 20, (time.time() - t) * 1e3, n / (time.time() - t) / 1e6, len(out) if out else -1))
