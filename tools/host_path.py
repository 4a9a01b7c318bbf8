"""tools/host_path.py -- BrotliEncoderCompress through the C ABI with the input and output in ordinary (pageable) host
memory, 64 MiB text at quality 5 / lgwin 22: what a drop-in caller sees, transfers included."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import brotli_mi355x  # noqa: E402
import synth  # noqa: E402

n = int(os.environ.get("HOST_MIB", "64")) << 20
data = synth.markov_text(n)
lib = brotli_mi355x.default_library()
L = lib.lib
L.BrotliEncoderCompress.restype = ctypes.c_int
L.BrotliEncoderCompress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
src = ctypes.create_string_buffer(data, n)
cap = n + n // 4 + 4096
dst = ctypes.create_string_buffer(cap)
times = []
for _ in range(8):
    out_size = ctypes.c_size_t(cap)
    t = time.time()
    ok = L.BrotliEncoderCompress(5, 22, 0, n, src, ctypes.byref(out_size), dst)
    times.append(time.time() - t)
    assert ok == 1
print("BrotliEncoderCompress host to host: %d -> %d bytes; ms per call: %s" % (n, out_size.value, " ".join("%.1f" % (x * 1e3) for x in times)))
