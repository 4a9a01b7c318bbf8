"""tools/c4_probe.py [pre] -- BASELINE configs[3]-like: 1 GiB Silesia-like through BrotliEncoderCompressMulti in 8 hinted shards, host to host; a warm-up call, then
two timed ones.  `pre`: a 64 MiB quality-4 call and a 1 GiB zero-fill call first (the pool state bench.py's side workloads leave behind)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import brotli_mi355x as bm
import large_cases, synth
frozen = json.load(open(os.path.join(ROOT, "tests", "golden", "large_hashes.json")))
name = "c4_silesia_1GiB_multi8_hinted"
case = large_cases.CASES[name]
lib = bm.default_library()
if len(sys.argv) > 1:
    lib.compress(synth.markov_text(64 << 20, 5), 4, 22)
    if sys.argv[1] == "pre2":
        big = synth.markov_text(16 << 20, 77)
        lib.BrotliCompress(big, {bm.BROTLI_PARAM_QUALITY: 2, bm.BROTLI_PARAM_LGWIN: 22}, 16)
        lib.compress(synth.markov_text(256 << 20, 3), 9, 22)
        lib.compress(synth.random_bytes(1 << 30), 5, 22)
    lib.compress(bytes(1 << 30), 5, 22)
    print("pre done", flush=True)
data = large_cases.make_input(name, frozen)
params = {bm.BROTLI_PARAM_QUALITY: 5, bm.BROTLI_PARAM_LGWIN: 22, bm.BROTLI_PARAM_SIZE_HINT: case["hint"]}
for i in range(3):
    t0 = time.time()
    out = bytes(lib.BrotliCompress(data, params, 8))
    print("call %d: %.0f ms, %d bytes" % (i, (time.time() - t0) * 1e3, len(out)), flush=True)
