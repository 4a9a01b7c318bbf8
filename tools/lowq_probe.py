"""tools/lowq_probe.py <quality> <KiB> [lgwin] -- one BrotliEncoderCompressStream-equivalent call on the text generator at a quality whose
device path is one sequential kernel per stream (2..4, 10, 11), for rocprofv3 --kernel-trace --stats: is anything but that kernel in the way?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import torch, synth
import brotli_mi355x as bm
from brotli_mi355x import multi
q, kib = int(sys.argv[1]), int(sys.argv[2])
lgwin = int(sys.argv[3]) if len(sys.argv) > 3 else 22
lib = bm.default_library(); enc = multi.ShardEncoder(lib.lib, 0)
data = synth.markov_text(kib << 10)
dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
params = [(bm.BROTLI_PARAM_QUALITY, q), (bm.BROTLI_PARAM_LGWIN, lgwin), (bm.BROTLI_PARAM_SIZE_HINT, len(data))]
torch.cuda.synchronize(); t0 = time.time()
out = enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=True)
torch.cuda.synchronize(); print("quality", q, "KiB", kib, round((time.time() - t0) * 1e3, 1), "ms", len(bytes(out)), "bytes", flush=True)
