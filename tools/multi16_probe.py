"""tools/multi16_probe.py -- BrotliEncoderCompressMulti, 16 MiB of text as 16 shards at quality 2 (light jobs) and 64 MiB as 8 shards at quality 5: wall time per call"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import synth
import brotli_mi355x as bm
lib = bm.default_library()
big = synth.markov_text(16 << 20, 77)
for rep in range(3):
    t0 = time.time(); out = bytes(lib.BrotliCompress(big, {bm.BROTLI_PARAM_QUALITY: 2, bm.BROTLI_PARAM_LGWIN: 22}, 16)); print("q2 16 shards %.0f ms (%d bytes)" % ((time.time() - t0) * 1e3, len(out)), flush=True)
big = synth.markov_text(64 << 20, 5)
for rep in range(2):
    t0 = time.time(); out = bytes(lib.BrotliCompress(big, {bm.BROTLI_PARAM_QUALITY: 5, bm.BROTLI_PARAM_LGWIN: 22, bm.BROTLI_PARAM_SIZE_HINT: len(big)}, 8)); print("q5 8 shards %.0f ms (%d bytes)" % ((time.time() - t0) * 1e3, len(out)), flush=True)
