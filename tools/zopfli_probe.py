"""tools/zopfli_probe.py -- quality 10 / 11 on the GPU: time and identity for a few inputs (run through gpurun)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu  # noqa: E402
import gpulib  # noqa: E402
import orc  # noqa: E402
import synth  # noqa: E402

L = gpulib.lib()
a = synth.alice()
cases = [("alice", a, 22), ("markov 1 MiB", synth.markov_text(1 << 20), 22), ("mixed 192 KiB", synth.mixed(3 << 16), 22)]
for name, d, w in cases:
    for q in (10, 11):
        params = [(1, q), (2, w), (5, len(d))]
        t = time.time()
        out, st = emu.encode_stream(L, d, params)
        dt = time.time() - t
        t = time.time()
        want, _ = orc.stream_compress(d, params)
        cpu = time.time() - t
        print("%-14s q%d: %d -> %d bytes  gpu %.2f s (%.3f MB/s)  oracle %.2f s (%.3f MB/s)  identical %s  lz77 %.1f ms  metablock %.1f ms" %
              (name, q, len(d), len(out), dt, len(d) / dt / 1e6, cpu, len(d) / cpu / 1e6, out == want, st["ms_lz77"], st["ms_metablock"]), flush=True)
