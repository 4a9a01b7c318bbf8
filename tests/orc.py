"""ctypes binding of the CPU oracle (oracle/liborc.so) and of the system brotli decoder.
Test infrastructure only."""
import ctypes, os, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class OrcStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("positions_searched", "positions_stored", "commands", "literals",
                                               "metablocks", "uncompressed_metablocks", "dict_lookups",
                                               "dict_matches")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OrcCommand(ctypes.Structure):
    _fields_ = [("insert_len_", ctypes.c_uint32), ("copy_len_", ctypes.c_uint32), ("dist_extra_", ctypes.c_uint32),
                ("cmd_prefix_", ctypes.c_uint16), ("dist_prefix_", ctypes.c_uint16)]


TRACE_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_size_t,
                            ctypes.POINTER(OrcCommand), ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32))

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        # ORC_FAST=1: the -O3 -march=native build of the same sources (bench baseline / hash freezing)
        fast = os.environ.get("ORC_FAST")
        path = os.path.join(ORACLE_DIR, "liborc_fast.so" if fast else "liborc.so")
        if not os.path.exists(path):
            if fast:
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liborc_fast.so"])
            else:
                build()
        L = ctypes.CDLL(path)
        L.orc_max_compressed_size.restype = ctypes.c_size_t
        L.orc_max_compressed_size.argtypes = [ctypes.c_size_t]
        L.orc_max_compressed_size_multi.restype = ctypes.c_size_t
        L.orc_max_compressed_size_multi.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
        L.orc_encoder_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p,
                                           ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p,
                                           ctypes.POINTER(OrcStats)]
        L.orc_writer_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
                                          ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p,
                                          ctypes.POINTER(OrcStats), ctypes.c_void_p, ctypes.c_void_p]
        L.orc_compress_multi.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32),
                                         ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p,
                                         ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_size_t]
        L.orc_log2f_restated.restype = ctypes.c_float
        L.orc_log2f_restated.argtypes = [ctypes.c_float]
        L.orc_bits_entropy.restype = ctypes.c_float
        L.orc_bits_entropy.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t]
        _lib = L
    return _lib


class ReferencePanics(Exception):
    """the restated code reached a state in which the Rust reference panics (and its FFI reports failure)"""


def _check_panic():
    flag = ctypes.c_int.in_dll(lib(), "orc_reference_would_panic")
    if flag.value:
        flag.value = 0
        raise ReferencePanics("the reference encoder fails on this input")


def compress(data, quality=5, lgwin=22, mode=0, with_stats=False):
    """one-shot BrotliEncoderCompress equivalent (size_hint = len)"""
    L = lib()
    cap = L.orc_max_compressed_size(len(data)) + 64
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    st = OrcStats()
    ok = L.orc_encoder_compress(quality, lgwin, mode, len(data), data, ctypes.byref(n), out, ctypes.byref(st))
    if not ok:
        raise RuntimeError("oracle compress failed")
    res = out.raw[:n.value]
    _check_panic()
    return (res, st.as_dict()) if with_stats else res


def writer_compress(data, quality=5, lgwin=22, chunk=0, with_stats=False, trace=None):
    """CompressorWriter feeding pattern (size_hint derived from the first write)"""
    L = lib()
    cap = L.orc_max_compressed_size(len(data)) + len(data) // 16 + 4096  # (room: tiny fragments of quality 0 / 1 outgrow the bound)
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    st = OrcStats()
    cb = TRACE_CB(trace) if trace else None
    ok = L.orc_writer_compress(quality, lgwin, chunk, len(data), data, ctypes.byref(n), out, ctypes.byref(st),
                               ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
    if not ok:
        raise RuntimeError("oracle writer compress failed")
    res = out.raw[:n.value]
    _check_panic()
    return (res, st.as_dict()) if with_stats else res


def reader_compress(data, params, chunk=4096, with_stats=False):
    """BrotliCompressCustomIo feeding pattern (what the reference's integration tests use): PROCESS per `chunk` bytes,
    then FINISH without input.  params: list of (ORC_PARAM_*, value)."""
    L = lib()
    L.orc_reader_compress.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t,
                                      ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p,
                                      ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.POINTER(OrcStats),
                                      ctypes.c_void_p, ctypes.c_void_p]
    keys = (ctypes.c_int * len(params))(*[k for k, _ in params])
    vals = (ctypes.c_uint32 * len(params))(*[v for _, v in params])
    cap = L.orc_max_compressed_size(len(data)) + len(data) // 16 + 4096
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    st = OrcStats()
    ok = L.orc_reader_compress(keys, vals, len(params), chunk, len(data), data, ctypes.byref(n), out, ctypes.byref(st),
                               None, None)
    if not ok:
        raise RuntimeError("oracle reader compress failed")
    res = out.raw[:n.value]
    _check_panic()
    return (res, st.as_dict()) if with_stats else res


def compress_multi(data, params, num_threads):
    L = lib()
    keys = (ctypes.c_int * len(params))(*[k for k, _ in params])
    vals = (ctypes.c_uint32 * len(params))(*[v for _, v in params])
    cap = L.orc_max_compressed_size_multi(len(data), num_threads)  # (what the reference's own binding hands over, c/py/brotli.py)
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    ok = L.orc_compress_multi(keys, vals, len(params), len(data), data, ctypes.byref(n), out, num_threads)
    _check_panic()
    if not ok:
        # compress_part gives every shard BrotliEncoderMaxCompressedSize(its length) of room and fails the call when the shard does not
        # fit (threading/mod.rs:337-411) -- which tiny fragments of quality 0 / 1 on incompressible input do not
        raise ReferencePanics("the reference's compress_multi reports an error")
    return out.raw[:n.value]


# ---- independent decoder (Google libbrotlidec) for round trips
_dec = None


def decompress(comp, expected_size):
    global _dec
    if _dec is None:
        _dec = ctypes.CDLL("libbrotlidec.so.1")
        _dec.BrotliDecoderDecompress.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t),
                                                 ctypes.c_char_p]
    cap = expected_size + 16
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    r = _dec.BrotliDecoderDecompress(len(comp), comp, ctypes.byref(n), out)
    if r != 1:
        raise RuntimeError("decoder rejected the stream (result %d)" % r)
    return out.raw[:n.value]


def stream_compress(data, params, prefix=None, collect_trace=False, continuation=True):
    """Generic path through the oracle's stream API: set params, optional custom dictionary
    (multi-thread continuation semantics), one FINISH call.  Returns (bytes, trace) where trace is a
    list of (kind, start, nbytes, [commands as tuples], dist_cache_after)."""
    L = lib()
    L.orc_encoder_create.restype = ctypes.c_void_p
    L.orc_encoder_destroy.argtypes = [ctypes.c_void_p]
    L.orc_encoder_set_parameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    L.orc_encoder_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_encoder_set_custom_dictionary.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int]
    L.orc_encoder_compress_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.orc_encoder_is_finished.argtypes = [ctypes.c_void_p]
    s = L.orc_encoder_create()
    trace = []

    def cb(opaque, kind, start, nbytes, cmds, n, dc):
        trace.append((kind, start, nbytes,
                      [(cmds[i].insert_len_, cmds[i].copy_len_, cmds[i].dist_extra_, cmds[i].cmd_prefix_,
                        cmds[i].dist_prefix_) for i in range(n)], tuple(dc[i] for i in range(4))))

    cbo = TRACE_CB(cb)
    for k, v in params:
        L.orc_encoder_set_parameter(s, k, v)
    if collect_trace:
        L.orc_encoder_set_trace(s, ctypes.cast(cbo, ctypes.c_void_p), None)
    if prefix is not None:
        L.orc_encoder_set_custom_dictionary(s, len(prefix), prefix, 1 if continuation else 0)
    cap = L.orc_max_compressed_size(len(data)) + len(data) // 16 + 4096
    out = ctypes.create_string_buffer(cap)
    inbuf = ctypes.create_string_buffer(data, len(data) if len(data) else 1)
    avail_in = ctypes.c_size_t(len(data))
    next_in = ctypes.c_void_p(ctypes.addressof(inbuf))
    avail_out = ctypes.c_size_t(cap)
    next_out = ctypes.c_void_p(ctypes.addressof(out))
    total = ctypes.c_size_t(0)
    ok = L.orc_encoder_compress_stream(s, 2, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                       ctypes.byref(next_out), ctypes.byref(total))
    fin = L.orc_encoder_is_finished(s)
    L.orc_encoder_destroy(s)
    if not ok or not fin:
        raise RuntimeError("oracle stream compress failed")
    _check_panic()
    return out.raw[:cap - avail_out.value], trace


def stream_with_flushes(data, params, cuts, write_size=0, dictionary=None):
    """Oracle stream API with BROTLI_OPERATION_FLUSH after the bytes up to each offset in `cuts` (ascending) and FINISH
    at the end; between flushes the input is handed over with PROCESS in pieces of `write_size` bytes (0: all at once).
    Returns the list of output pieces (one per flush + the final one)."""
    L = lib()
    L.orc_encoder_create.restype = ctypes.c_void_p
    L.orc_encoder_destroy.argtypes = [ctypes.c_void_p]
    L.orc_encoder_set_parameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    L.orc_encoder_compress_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.orc_encoder_is_finished.argtypes = [ctypes.c_void_p]
    L.orc_encoder_has_more_output.argtypes = [ctypes.c_void_p]
    s = L.orc_encoder_create()
    for k, v in params:
        L.orc_encoder_set_parameter(s, k, v)
    if dictionary is not None:  # BrotliEncoderSetCustomDictionary (encode.rs:1196-1270)
        L.orc_encoder_set_custom_dictionary.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int]
        L.orc_encoder_set_custom_dictionary(s, len(dictionary), dictionary, 0)
    cap = L.orc_max_compressed_size(len(data)) + len(data) // 16 + 4096 + 16 * len(cuts)
    out = ctypes.create_string_buffer(cap)
    inbuf = ctypes.create_string_buffer(data, len(data) if len(data) else 1)
    base = ctypes.addressof(inbuf)
    avail_out = ctypes.c_size_t(cap)
    next_out = ctypes.c_void_p(ctypes.addressof(out))
    total = ctypes.c_size_t(0)
    pieces = []
    pos = 0
    done_out = 0

    def call(op, lo, hi):
        avail_in = ctypes.c_size_t(hi - lo)
        next_in = ctypes.c_void_p(base + lo)
        while True:
            ok = L.orc_encoder_compress_stream(s, op, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                               ctypes.byref(next_out), ctypes.byref(total))
            if not ok:
                raise RuntimeError("oracle compress_stream failed (op %d)" % op)
            if avail_in.value == 0 and not L.orc_encoder_has_more_output(s):
                break

    for item in list(cuts) + [len(data)]:
        # an entry of `cuts` is an offset (flush there) or (offset, metadata bytes): BROTLI_OPERATION_EMIT_METADATA after
        # the input up to the offset has been handed over with PROCESS
        meta = None
        cut = item
        if isinstance(item, tuple):
            cut, meta = item
        final = meta is None and cut == len(data) and len(pieces) == len(cuts)
        if write_size:
            while cut - pos > write_size:
                call(0, pos, pos + write_size)
                pos += write_size
        if meta is None:
            call(2 if final else 1, pos, cut)
        else:
            if cut > pos:
                call(0, pos, cut)
            mbuf = ctypes.create_string_buffer(bytes(meta), max(1, len(meta)))
            avail_in = ctypes.c_size_t(len(meta))
            next_in = ctypes.c_void_p(ctypes.addressof(mbuf))
            while True:
                ok = L.orc_encoder_compress_stream(s, 3, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                                   ctypes.byref(next_out), ctypes.byref(total))
                if not ok:
                    raise RuntimeError("oracle EMIT_METADATA failed")
                if avail_in.value == 0 and not L.orc_encoder_has_more_output(s):
                    break
        pos = cut
        produced = cap - avail_out.value
        pieces.append(out.raw[done_out:produced])
        done_out = produced
    fin = L.orc_encoder_is_finished(s)
    L.orc_encoder_destroy(s)
    if not fin:
        raise RuntimeError("oracle stream did not finish")
    _check_panic()
    return pieces


def concat(files, window=None, bs=4096):
    """The oracle's BroCatli (oracle/orc_multi.c: concat/mod.rs:274-608 restated) driven like the reference's own test helper
    (src/bin/test_broccoli.rs:28-132): reads and writes of `bs` bytes.  Returns (result code, bytes): 0 = success, >= 124 = the
    BroCatliResult failure the reference would report."""
    L = lib()
    n = len(files)
    bufs = [ctypes.create_string_buffer(f, len(f) if len(f) else 1) for f in files]
    ptrs = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(b) for b in bufs])
    sizes = (ctypes.c_size_t * max(n, 1))(*[len(f) for f in files])
    cap = sum(len(f) for f in files) + 16 * n + 64
    out = ctypes.create_string_buffer(cap)
    got = ctypes.c_size_t(0)
    L.orc_concat.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p,
                             ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    r = L.orc_concat(n, ptrs, sizes, -1 if window is None else window, bs, out, cap, ctypes.byref(got))
    assert r >= 0, "oracle concat: output buffer too small / unexpected state (%d)" % r
    return r, out.raw[:got.value]
