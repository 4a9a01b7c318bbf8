"""The BroCatli streaming C API (c/brotli/broccoli.h, src/ffi/broccoli.rs): catable / appendable streams fed in small
pieces and drained through a small output buffer must give the bytes of the whole-chunk stitcher (which the multi-shard
tests compare with the oracle's compress_multi), and must decode to the concatenation of the inputs."""
import ctypes
import os

import emu
import orc
import synth

CAT, APP, Q, W = 167, 168, 1, 2


class State(ctypes.Structure):
    _fields_ = [("unused", ctypes.c_void_p), ("data", ctypes.c_ubyte * 248)]


def _bind():
    emu.build()
    L = ctypes.CDLL(os.path.join(emu.EMU_DIR, "libbrotli_emu.so"))
    L.BroccoliCreateInstance.restype = State
    L.BroccoliCreateInstanceWithWindowSize.restype = State
    L.BroccoliCreateInstanceWithWindowSize.argtypes = [ctypes.c_uint8]
    L.BroccoliDestroyInstance.argtypes = [State]
    L.BroccoliNewBrotliFile.argtypes = [ctypes.POINTER(State)]
    L.BroccoliConcatStreaming.argtypes = [ctypes.POINTER(State), ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p,
                                          ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
    L.BroccoliConcatFinished.argtypes = [ctypes.POINTER(State), ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
    return L


def _concat(L, files, in_piece, out_piece, window=None):
    st = L.BroccoliCreateInstance() if window is None else L.BroccoliCreateInstanceWithWindowSize(window)
    out = bytearray()
    buf = ctypes.create_string_buffer(out_piece)
    for f in files:
        L.BroccoliNewBrotliFile(ctypes.byref(st))
        for i in range(0, len(f), in_piece):
            piece = f[i:i + in_piece]
            avail_in = ctypes.c_size_t(len(piece))
            while True:
                avail_out = ctypes.c_size_t(out_piece)
                # (the pointer is passed by value, src/ffi/broccoli.rs:109-128: the caller moves on by what was consumed)
                rest = piece[len(piece) - avail_in.value:]
                r = L.BroccoliConcatStreaming(ctypes.byref(st), ctypes.byref(avail_in), rest, ctypes.byref(avail_out), buf)
                out += buf.raw[:out_piece - avail_out.value]
                assert r in (1, 2), r
                if r == 1:  # BroccoliNeedsMoreInput
                    assert avail_in.value == 0
                    break
    while True:
        avail_out = ctypes.c_size_t(out_piece)
        r = L.BroccoliConcatFinished(ctypes.byref(st), ctypes.byref(avail_out), buf)
        out += buf.raw[:out_piece - avail_out.value]
        assert r in (0, 2), r
        if r == 0:
            break
    L.BroccoliDestroyInstance(st)
    return bytes(out)


def test_streaming_concat_equals_whole_chunk_stitcher():
    import test_cabi
    lib = test_cabi._load("emu")
    L = _bind()
    a, b, c = synth.alice()[:70000], synth.markov_text(50000, 3), b"tail"
    files = [orc.stream_compress(x, [(Q, 5), (W, lg), (CAT, 1), (APP, 1)])[0] for x, lg in ((a, 20), (b, 18), (c, 18))]
    want = lib.concat_chunks(files)
    for in_piece, out_piece in ((1 << 20, 1 << 20), (7, 11), (4096, 5)):
        got = _concat(L, files, in_piece, out_piece)
        assert got == want
    assert orc.decompress(want, len(a) + len(b) + len(c)) == a + b + c
    # a window fixed up front (BroccoliCreateInstanceWithWindowSize): files with smaller windows still fit
    got = _concat(L, files, 1000, 1000, window=22)
    assert orc.decompress(got, len(a) + len(b) + len(c)) == a + b + c
    # a file that was not encoded catable is refused with a result code >= 124
    plain = orc.compress(a, 5, 22)
    st = L.BroccoliCreateInstance()
    L.BroccoliNewBrotliFile(ctypes.byref(st))
    buf = ctypes.create_string_buffer(1 << 20)
    for f in (files[0], plain):
        L.BroccoliNewBrotliFile(ctypes.byref(st))
        avail_in, avail_out = ctypes.c_size_t(len(f)), ctypes.c_size_t(1 << 20)
        r = L.BroccoliConcatStreaming(ctypes.byref(st), ctypes.byref(avail_in), f, ctypes.byref(avail_out), buf)
    avail_out = ctypes.c_size_t(1 << 20)
    r = L.BroccoliConcatFinished(ctypes.byref(st), ctypes.byref(avail_out), buf)
    assert r >= 124, r
    L.BroccoliDestroyInstance(st)


def test_streaming_concat_in_constant_memory_slices():
    """files larger than the 64 KiB slice the front end takes in at a time (broccoli_api.cpp), fed and drained in sizes that
    do not divide anything; a file whose header arrives one byte at a time; an empty file between two real ones"""
    import test_cabi
    lib = test_cabi._load("emu")
    L = _bind()
    a, b = synth.random_bytes(300000, 5), synth.markov_text(1 << 20, 9)
    files = [orc.stream_compress(x, [(Q, 5), (W, 22), (CAT, 1), (APP, 1)])[0] for x in (a, b, a[:1000])]
    assert len(files[0]) > 4 * 65536 and len(files[1]) > 2 * 65536
    want = lib.concat_chunks(files)
    for in_piece, out_piece in ((100003, 3001), (1, 1 << 16), (1 << 22, 7), (65536, 65536), (65537, 65535)):
        assert _concat(L, files, in_piece, out_piece) == want, (in_piece, out_piece)
    assert orc.decompress(want, len(a) + len(b) + 1000) == a + b + a[:1000]
    # BroccoliNewBrotliFile without any bytes (a file that turned out empty) leaves the stream as it is
    with_gap = [files[0], b"", files[1], files[2]]
    assert _concat(L, with_gap, 4096, 4096) == want
