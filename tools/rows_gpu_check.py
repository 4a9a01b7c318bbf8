"""tools/rows_gpu_check.py -- GPU check of the candidate-row maintenance on literal-heavy input (run through gpurun): byte identity
with the oracle at sizes it finishes quickly, with BROTLI_MI355X_SELFTEST=1 (rows recomputed on the host after every update) on the
small ones; then the phase split of the 64 MiB distributions."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpulib  # noqa: E402
import synth  # noqa: E402
from cmp_stream import check_bytes  # noqa: E402

L = gpulib.lib()
ok = True
for name, data in (("random 3 MiB", synth.random_bytes(3 << 20)), ("mixed 4 MiB", synth.mixed(4 << 20)),
                   ("silesia-like 6 MiB", synth.silesia_like(6 << 20, min_segment=1 << 16, max_segment=1 << 20)),
                   ("stretches 3 MiB", synth.stretches(3 << 20, 5)), ("random 24 MiB", synth.random_bytes(24 << 20, 77)),
                   ("hex 8 MiB", synth.silesia_like(8 << 20, only=85))):
    ok &= bool(check_bytes(L, name, data, [(1, 5), (2, 22), (5, len(data))]))
print("IDENTITY", "OK" if ok else "FAILED", flush=True)
sys.exit(0 if ok else 1)
