"""tools/rows_selftest_gpu.py -- GPU check of the candidate-row maintenance with the rows compared against a rebuild from the flags
after EVERY update (BROTLI_MI355X_SELFTEST_ROWS, Lz77Stage::SelfTestRows): literal-heavy and mixed inputs small enough for the
host rebuild, byte identity with the oracle on top."""
import os, sys
os.environ["BROTLI_MI355X_SELFTEST_ROWS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpulib, synth  # noqa: E402
from cmp_stream import check_bytes  # noqa: E402
L = gpulib.lib()
ok = True
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [("random 2 MiB", synth.random_bytes(2 << 20)), ("mixed 2 MiB", synth.mixed(2 << 20)),
         ("silesia-like 2 MiB", synth.silesia_like(2 << 20, min_segment=1 << 15, max_segment=1 << 18)), ("text 2 MiB", synth.markov_text(2 << 20))]
if not quick:
    cases += [("random 6 MiB", synth.random_bytes(6 << 20, 99)), ("mixed 3 MiB", synth.mixed(3 << 20)),
              ("silesia-like 4 MiB", synth.silesia_like(4 << 20, min_segment=1 << 16, max_segment=1 << 20)), ("stretches 2 MiB", synth.stretches(2 << 20, 5))]
for name, data in cases:
    ok &= bool(check_bytes(L, name, data, [(1, 5), (2, 22), (5, len(data))]))
print("ROWS SELFTEST", "OK" if ok else "FAILED", flush=True)
sys.exit(0 if ok else 1)
