"""Qualities 0 / 1 with the fragments of a call side by side (round 5): MB/s of BrotliEncoderCompress on the text generator for a few
sizes and windows, identity with the oracle (liborc_fast.so, one pinned core, timed beside it).  Run on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import synth
import orc
import brotli_mi355x

lib = brotli_mi355x.default_library()
out = []
for mib, w in ((2, 22), (64, 18), (64, 22), (256, 18)):
    data = synth.markov_text(mib << 20, 5)
    for q in (0, 1):
        lib.compress(data[:1 << 20], q, w)
        best = None
        for _ in range(2):
            t0 = time.time()
            got = lib.compress(data, q, w)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        t0 = time.time()
        want = orc.compress(data, q, w) if mib <= 64 else None
        cpu = time.time() - t0
        rec = {"mib": mib, "lgwin": w, "quality": q, "gpu_MBps": round(len(data) / best / 1e6, 1), "ms": round(best * 1e3, 1),
               "identical": (got == want) if want is not None else None, "cpu_MBps": round(len(data) / cpu / 1e6, 1) if want is not None else None}
        print(json.dumps(rec), flush=True)
        out.append(rec)
