"""tools/dirty_probe.py <kind> <MiB> [VAR=value ...] -- one quality-5 call on generated input (random | text | mixed) with
BROTLI_MI355X_DEBUG_DIRTY: per round, how many segments are dirty and why, and how the dirty stretches lie inside their blocks."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
kind, mib = sys.argv[1], int(sys.argv[2])
path = "/tmp/dirty_%d.txt" % os.getpid()
os.environ["BROTLI_MI355X_DEBUG_DIRTY"] = path
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1); os.environ[k] = v
import torch, synth
import brotli_mi355x as bm
from brotli_mi355x import multi
lib = bm.default_library(); enc = multi.ShardEncoder(lib.lib, 0)
n = mib << 20
data = synth.random_bytes(n, 0x5EED000000000005) if kind == "random" else (synth.markov_text(n) if kind == "text" else synth.silesia_like(n, 7))
dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
params = [(bm.BROTLI_PARAM_QUALITY, 5), (bm.BROTLI_PARAM_LGWIN, 22), (bm.BROTLI_PARAM_SIZE_HINT, len(data))]
torch.cuda.synchronize(); t0 = time.time()
out = enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=True)
torch.cuda.synchronize(); print(kind, mib, "MiB", round((time.time() - t0) * 1e3, 1), "ms", len(bytes(out)), "bytes, rounds", enc.stats[0], flush=True)
lines = open(path).read().split("\n")
os.remove(path)
for r, line in enumerate(l for l in lines if l):
    m, why = line.split(" ")
    nseg = len(m)
    per_block = 32 if nseg >= 32 else nseg
    c = collections.Counter(m)
    # dirty stretches: length histogram and where in the block they start
    runs = collections.Counter(); starts = collections.Counter()
    i = 0
    while i < nseg:
        if m[i] == '.':
            i += 1; continue
        j = i
        while j < nseg and m[j] != '.' and (j == i or j % per_block != 0): j += 1
        runs[min(j - i, 33)] += 1; starts[(i % per_block) // 4] += 1
        i = j
    print("round %2d: %s | why %s | stretch lengths %s | start octant %s" % (r, dict(c), dict(collections.Counter(why)), sorted(runs.items())[:12], sorted(starts.items())))
    if r <= 3:
        # the first few blocks that have anything dirty
        shown = 0
        for b in range(0, nseg, per_block):
            if any(ch != '.' for ch in m[b:b + per_block]):
                print("     block %5d  %s  %s" % (b // per_block, m[b:b + per_block], why[b:b + per_block])); shown += 1
                if shown >= 6: break
