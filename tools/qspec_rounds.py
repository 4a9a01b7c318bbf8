"""tools/qspec_rounds.py MiB q -- one call of quality q on text of that size with BROTLI_MI355X_DEBUG: the rounds of the speculative quick path"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, test_cabi
lib = test_cabi._load("gpu")
mib, q = int(sys.argv[1]), int(sys.argv[2])
d = synth.markov_text(mib << 20)
lib.compress(d, q, 22)
os.environ["BROTLI_MI355X_DEBUG"] = "1"
t = time.time()
lib.compress(d, q, 22)
print("q%d %d MiB: %.2f ms" % (q, mib, (time.time() - t) * 1e3))
