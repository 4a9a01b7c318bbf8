"""Shared cases for the distance-cache check (lz77_check_cache): a literal-only segment stays literal-only under another
cache iff none of the candidate distances derived from it (4 at q5-6; 10 at q7-8: last +-1..3 as well, mod.rs:632-651,
1707-1741) matches two bytes at a searched position."""
import ctypes

import synth


def run(L):
    L.brotli_mi355x_debug_check_cache.restype = ctypes.c_int
    L.brotli_mi355x_debug_check_cache.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                                  ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32, ctypes.c_uint32]
    n = 8192
    base = bytearray(synth.random_bytes(n, 99))
    # make sure there is no accidental two-byte match at the distances used below
    cache = (ctypes.c_int32 * 4)(1000, 1500, 2000, 2500)
    searched = [4096 + 17 * i for i in range(100)]

    def clean(buf):
        for p in searched:
            for d in (1000, 1500, 2000, 2500, 999, 1001, 998, 1002, 997, 1003):
                if buf[p] == buf[p - d] and buf[p + 1] == buf[p + 1 - d]:
                    buf[p - d] ^= 0x55
        return buf

    def call(buf, flags, ndist):
        return L.brotli_mi355x_debug_check_cache(bytes(buf), bytes(flags), n, 4096, 6144, cache, ndist, (1 << 22) - 16)

    flags = bytearray(n)
    for p in searched:
        flags[p] = 3  # stored + searched
    buf = clean(bytearray(base))
    assert call(buf, flags, 4) == 1 and call(buf, flags, 10) == 1
    # a two-byte match at the last distance itself: caught by both
    hit = bytearray(buf)
    p = searched[40]
    hit[p - 1000] = hit[p]
    hit[p - 999] = hit[p + 1]
    assert call(hit, flags, 4) == 0 and call(hit, flags, 10) == 0
    # at last distance + 1 (candidate 5 of the q7-8 list): only the 10-candidate check may object
    hit = bytearray(buf)
    hit[p - 1001] = hit[p]
    hit[p - 1000] = hit[p + 1]
    hit = bytearray(hit)
    assert call(hit, flags, 10) == 0
    assert call(hit, flags, 4) in (0, 1)
    # the same bytes at a position that was NOT searched do not count
    off = bytearray(buf)
    q = searched[40] + 5
    off[q - 1000] = off[q]
    off[q - 999] = off[q + 1]
    assert call(clean(off), flags, 10) == 1
    # distances beyond max_backward / non-positive cache entries are ignored
    far = (ctypes.c_int32 * 4)(0x7ffffff0, -5, 0, 5000000)
    assert L.brotli_mi355x_debug_check_cache(bytes(buf), bytes(flags), n, 4096, 6144, far, 10, (1 << 22) - 16) == 1
