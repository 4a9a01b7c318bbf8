"""ctypes binding of the product library rust-brotli_amd/libbrotli_mi355x.so (HIP backed)."""
import ctypes
import os

import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BROTLI_MI355X_LIB") or os.path.join(ROOT, "rust-brotli_amd", "libbrotli_mi355x.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("product library not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = emu.bind_trace(ctypes.CDLL(LIB_PATH))
    return _lib
