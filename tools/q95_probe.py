"""tools/q95_probe.py [lgwin] -- quality 10 + BROTLI_PARAM_Q9_5 on 8 MiB of the text generator (bench.py's q9_5_text_8MiB entries), two calls:
for rocprofv3 --kernel-trace --stats (where a lone meta-block's time goes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import torch, synth
import brotli_mi355x as bm
from brotli_mi355x import multi
lgwin = int(sys.argv[1]) if len(sys.argv) > 1 else 22
lib = bm.default_library(); enc = multi.ShardEncoder(lib.lib, 0)
data = synth.markov_text(8 << 20)
dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
params = [(bm.BROTLI_PARAM_QUALITY, 10), (150, 1), (bm.BROTLI_PARAM_LGWIN, lgwin), (bm.BROTLI_PARAM_SIZE_HINT, len(data))]
for i in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=True)
    torch.cuda.synchronize(); print("call", i, round((time.time() - t0) * 1e3, 1), "ms", len(bytes(out)), "bytes", flush=True)
