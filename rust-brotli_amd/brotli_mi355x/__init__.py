"""brotli_mi355x -- Python host-side binding of the MI355X brotli encoder library (C ABI via ctypes).

Mirrors the reference's own ctypes binding (c/py/brotli.py:78-180).  The library is HIP only: there
is no CPU fallback, importing fails loudly if the shared object has not been built."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libbrotli_mi355x.so")
if not os.path.exists(LIB_PATH):
    raise ImportError("libbrotli_mi355x.so is missing; build it with `make -C rust-brotli_amd` "
                      "(or __graft_entry__.build()). There is no CPU fallback.")
lib = ctypes.CDLL(LIB_PATH)
