"""tools/c4_workers.py -- BrotliEncoderCompressMulti over the 1 GiB Silesia-like input (8 shards) on one GPU, timed host to
host; run once per BROTLI_MI355X_SHARD_WORKERS setting (the library reads it once per process)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import brotli_mi355x  # noqa: E402
import json  # noqa: E402

import large_cases  # noqa: E402

frozen = json.load(open(os.path.join(ROOT, "tests", "golden", "large_hashes.json")))
data = large_cases.make_input("c4_silesia_1GiB_multi8_hinted", frozen)  # (a seed on which the reference encoder does not fail)
n = len(data)
lib = brotli_mi355x.default_library()
best = None
for _ in range(3):
    t = time.time()
    out = lib.BrotliCompress(data, {1: 5, 2: 22, 5: 1 << 30}, 8)
    dt = time.time() - t
    best = dt if best is None else min(best, dt)
print("workers %s: %d -> %d bytes, best of 3 %.1f ms (%.0f MB/s)" % (os.environ.get("BROTLI_MI355X_SHARD_WORKERS", "default"), n, len(out), best * 1e3, n / best / 1e6))
