"""ctypes binding of tests/emu/libbrotli_emu.so: the product's host driver + chain code compiled for
the CPU (one lane per wave).  Test infrastructure only."""
import ctypes, os, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


class Command(ctypes.Structure):
    _fields_ = [("insert_len_", ctypes.c_uint32), ("copy_len_", ctypes.c_uint32), ("dist_extra_", ctypes.c_uint32),
                ("cmd_prefix_", ctypes.c_uint16), ("dist_prefix_", ctypes.c_uint16)]


class MetaBlockInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("start", "end", "cmd_offset", "n_cmds", "n_literals", "uncompressed",
                                               "is_last", "pad")] + [("dist_cache_after", ctypes.c_int32 * 4)]


def build():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])


def bind_trace(L):
    L.brotli_mi355x_lz77_trace.restype = ctypes.c_long
    L.brotli_mi355x_lz77_trace.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p,
                                           ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.POINTER(Command), ctypes.c_size_t, ctypes.POINTER(MetaBlockInfo),
                                           ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                           ctypes.POINTER(ctypes.c_uint32), ctypes.c_char_p, ctypes.c_size_t]
    return L


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = bind_trace(ctypes.CDLL(os.path.join(EMU_DIR, "libbrotli_emu.so")))
    return _lib


def lz77_trace(L, data, quality=5, lgwin=22, size_hint=None, catable=False, prefix=b"", segment_bytes=4096):
    """returns (metablocks, stats): metablocks = list of dict(start,end,uncompressed,cmds=[tuples],dist_cache_after)"""
    if size_hint is None:
        size_hint = len(data)
    cap = len(data) // 2 + len(data) // 65536 * 4 + 1024
    cmds = (Command * cap)()
    mbs = (MetaBlockInfo * 4096)()
    nmb = ctypes.c_size_t(0)
    stats = (ctypes.c_uint32 * 12)()
    err = ctypes.create_string_buffer(512)
    n = L.brotli_mi355x_lz77_trace(quality, lgwin, size_hint, 1 if catable else 0, prefix, len(prefix), data, len(data),
                                   segment_bytes, cmds, cap, mbs, 4096, ctypes.byref(nmb), stats, err, 512)
    if n < 0:
        raise RuntimeError(err.value.decode())
    out = []
    for i in range(nmb.value):
        m = mbs[i]
        cl = [(cmds[j].insert_len_, cmds[j].copy_len_, cmds[j].dist_extra_, cmds[j].cmd_prefix_, cmds[j].dist_prefix_)
              for j in range(m.cmd_offset, m.cmd_offset + m.n_cmds)]
        out.append(dict(start=m.start, end=m.end, uncompressed=bool(m.uncompressed), is_last=bool(m.is_last), cmds=cl,
                        n_literals=m.n_literals, dist_cache_after=tuple(m.dist_cache_after)))
    names = ('keys', 'sort', 'init', 'rank', 'parse', 'resolve', 'gather', 'total')
    return out, dict(rounds=stats[0], segments_parsed=stats[1], searches=stats[2], total_cmds=stats[3],
                     ms={n: stats[4 + i] / 1000.0 for i, n in enumerate(names)})


def bind_encode(L):
    L.brotli_mi355x_encode_stream.restype = ctypes.c_long
    L.brotli_mi355x_encode_stream.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32),
                                              ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32,
                                              ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double),
                                              ctypes.c_char_p, ctypes.c_size_t]
    return L


def encode_stream(L, data, params, prefix=b"", continuation=True, segment_bytes=0):
    """params: list of (BrotliEncoderParameter id, value).  Returns (bytes, stats dict)."""
    bind_encode(L)
    keys = (ctypes.c_int * len(params))(*[k for k, _ in params])
    vals = (ctypes.c_uint32 * len(params))(*[v for _, v in params])
    cap = len(data) + len(data) // 4 + 4096
    out = ctypes.create_string_buffer(cap)
    st = (ctypes.c_double * 32)()
    err = ctypes.create_string_buffer(512)
    buf = ctypes.create_string_buffer(data, len(data) if data else 1)
    n = L.brotli_mi355x_encode_stream(keys, vals, len(params), prefix, len(prefix), 1 if continuation else 0,
                                      ctypes.cast(buf, ctypes.c_void_p), len(data), 0, segment_bytes, out, cap, st, err, 512)
    if n < 0:
        raise RuntimeError(err.value.decode())
    names = ["lz77_rounds", "searches", "commands", "literals", "metablocks", "uncompressed_metablocks",
             "fallback_retries", "ms_lz77", "ms_metablock", "ms_total"]
    d = {k: st[i] for i, k in enumerate(names)}
    d["ms_phase"] = [st[10 + i] for i in range(16)]
    d["chains_launched"] = st[28]  # over all launches: warm-up dry runs, round 0, re-parses
    d["num_segments"] = st[29]
    return out.raw[:n], d
