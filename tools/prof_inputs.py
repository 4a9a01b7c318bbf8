"""tools/prof_inputs.py [names...] -- one compression of each 64 MiB test distribution on the GPU with the stage timers
on (run through gpurun); prints the library's own phase split.  names: text random zero silesia binary hex enwik"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu  # noqa: E402
import gpulib  # noqa: E402
import synth  # noqa: E402

N = int(os.environ.get("PROF_MIB", "64")) << 20
Q = int(os.environ.get("PROF_Q", "5"))
GEN = {
    "text": lambda: synth.markov_text(N),
    "random": lambda: synth.random_bytes(N),
    "zero": lambda: bytes(N),
    "silesia": lambda: synth.silesia_like(N),
    "binary": lambda: synth.silesia_like(N, only=60),
    "hex": lambda: synth.silesia_like(N, only=85),
    "enwik": lambda: synth.enwik_like(N),
}
L = gpulib.lib()
for name in (sys.argv[1:] or list(GEN)):
    data = GEN[name]()
    emu.encode_stream(L, data, [(1, Q), (2, 22), (5, len(data))])  # warm the pools
    t = time.time()
    out, st = emu.encode_stream(L, data, [(1, Q), (2, 22), (5, len(data))])
    dt = time.time() - t
    print("== %s: %d -> %d bytes, %.1f ms (%.0f MB/s), rounds %d, lz77 %.1f ms, metablock %.1f ms; final parse: %d searches, %d commands, %d literals" %
          (name, len(data), len(out), dt * 1e3, len(data) / dt / 1e6, st["lz77_rounds"], st["ms_lz77"], st["ms_metablock"],
           st.get("searches", -1), st.get("commands", -1), st.get("literals", -1)), flush=True)
