"""tools/c4_after_big.py -- the 1 GiB / 8 hinted shards call in a process that has run 1 GiB one-shot calls before (what bench.py's extras do)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import brotli_mi355x, large_cases, synth
frozen = json.load(open(os.path.join(ROOT, "tests", "golden", "large_hashes.json")))
lib = brotli_mi355x.default_library()
data = large_cases.make_input("c4_silesia_1GiB_multi8_hinted", frozen)
def multi(tag):
    t = time.time()
    out = lib.BrotliCompress(data, {1: 5, 2: 22, 5: 1 << 30}, 8)
    print("%s: multi8 %.0f ms" % (tag, (time.time() - t) * 1e3), flush=True)
multi("fresh"); multi("fresh")
big = bytes(1 << 30) if len(sys.argv) < 2 or sys.argv[1] == "zero" else synth.random_bytes(1 << 30)
for _ in range(2):
    t = time.time()
    lib.compress(big, 5, 22)
    print("one-shot 1 GiB %.0f ms" % ((time.time() - t) * 1e3), flush=True)
for i in range(4):
    multi("after")
print("trim:", lib.lib.BrotliMi355xTrimPool() >> 20, "MiB")
multi("trimmed"); multi("trimmed")
