"""The BroCatli streaming C API (c/brotli/broccoli.h, src/ffi/broccoli.rs): catable / appendable streams fed in small
pieces and drained through a small output buffer must give the bytes of the ORACLE's BroCatli (oracle/orc_multi.c,
concat/mod.rs:274-608 restated, driven like src/bin/test_broccoli.rs:28-132), the bytes of the product's own whole-chunk
stitcher, and must decode to the concatenation of the inputs.  The cases of the reference's test_broccoli.rs (append then
cat, empty files, one and two byte files on either side, mixed window sizes with and without a window fixed up front,
append-only twice = refused) run against BOTH builds of the product's host code: the product library
(rust-brotli_amd/libbrotli_mi355x.so -- concatenation is host-only code, no device call is made) and the emulation build."""
import ctypes
import os

import emu
import orc
import synth

CAT, APP, Q, W = 167, 168, 1, 2


class State(ctypes.Structure):
    _fields_ = [("unused", ctypes.c_void_p), ("data", ctypes.c_ubyte * 248)]


PRODUCT_SO = os.path.join(os.path.dirname(emu.ROOT + "/"), "rust-brotli_amd", "libbrotli_mi355x.so")


def _bind(kind="emu"):
    emu.build()
    L = ctypes.CDLL(PRODUCT_SO if kind == "product" else os.path.join(emu.EMU_DIR, "libbrotli_emu.so"))
    L.BroccoliCreateInstance.restype = State
    L.BroccoliCreateInstanceWithWindowSize.restype = State
    L.BroccoliCreateInstanceWithWindowSize.argtypes = [ctypes.c_uint8]
    L.BroccoliDestroyInstance.argtypes = [State]
    L.BroccoliNewBrotliFile.argtypes = [ctypes.POINTER(State)]
    L.BroccoliConcatStreaming.argtypes = [ctypes.POINTER(State), ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p,
                                          ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
    L.BroccoliConcatFinished.argtypes = [ctypes.POINTER(State), ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
    return L


def _concat(L, files, in_piece, out_piece, window=None, want_error=False):
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
                if want_error and r >= 124:
                    L.BroccoliDestroyInstance(st)
                    return r
                assert r in (1, 2), r
                if r == 1:  # BroccoliNeedsMoreInput
                    assert avail_in.value == 0
                    break
    while True:
        avail_out = ctypes.c_size_t(out_piece)
        r = L.BroccoliConcatFinished(ctypes.byref(st), ctypes.byref(avail_out), buf)
        out += buf.raw[:out_piece - avail_out.value]
        if want_error and r >= 124:
            L.BroccoliDestroyInstance(st)
            return r
        assert r in (0, 2), r
        if r == 0:
            break
    L.BroccoliDestroyInstance(st)
    return bytes(out)


def _enc(data, lgwin=22, first=False, quality=5, extra=()):
    """the encodings of test_broccoli.rs: the first file appendable with the static dictionary, the others catable"""
    params = [(Q, quality), (W, lgwin)] + ([(APP, 1)] if first else [(CAT, 1), (APP, 1)]) + list(extra)
    return orc.stream_compress(data, params)[0]


def _reference_cases():
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ukko = open(os.path.join(golden, "small", "ukkonooa"), "rb").read()
    fox = open(os.path.join(golden, "small", "quickfox"), "rb").read()
    alice = synth.alice()
    rtu = open(os.path.join(golden, "random_then_unicode"), "rb").read()
    yield "append then empty", [ukko, b""], None
    yield "append then cat", [ukko, fox], None
    yield "one byte", [ukko, bytes([8])], None
    yield "one byte before", [bytes([8]), ukko], None
    yield "two byte", [ukko, bytes([8, 9])], None
    yield "two byte before", [bytes([8, 9]), ukko], None
    yield "empty then cat", [b"", fox], None
    yield "three empties", [b"", b"", b""], None
    yield "many", [alice, rtu[:100000], ukko, fox, b"", bytes([8]), alice[:3000]], None


def _check_reference_cases(kind):
    L = _bind(kind)
    for name, plain, _ in _reference_cases():
        # test_concat (test_broccoli.rs:306-410): every file at its own window size and quality; window fixed up front or not
        for variant, lgwins, window in (("same window", [22] * len(plain), None), ("mixed windows", [(22, 18, 20, 16, 24, 17, 10)[i % 7] for i in range(len(plain))], None),
                                        ("mixed, window 24 up front", [(22, 18, 20, 16, 24, 17, 10)[i % 7] for i in range(len(plain))], 24),
                                        ("descending", [24 - i for i in range(len(plain))], None)):
            files = [_enc(x, lgwins[i], first=(i == 0), quality=(5, 7, 9, 6)[i % 4]) for i, x in enumerate(plain)]
            r, want = orc.concat(files, window=window, bs=2)
            for bs in (4096, 1):
                r2, want2 = orc.concat(files, window=window, bs=bs)
                assert (r2, want2 if r2 == 0 else b"") == (r, want if r == 0 else b""), (name, variant, bs)
            if r != 0:
                # the reference refuses (a later file with a larger window than the first / than the one fixed up front)
                for in_piece, out_piece in ((1 << 20, 1 << 20), (1, 3)):
                    got = _concat(L, files, in_piece, out_piece, window=window, want_error=True)
                    assert got == r, (kind, name, variant, got, r)
                continue
            for in_piece, out_piece in ((1 << 20, 1 << 20), (1, 1 << 16), (2, 2), (4096, 5), (7, 11)):
                got = _concat(L, files, in_piece, out_piece, window=window)
                assert got == want, (kind, name, variant, in_piece, out_piece)
            total = b"".join(plain)
            assert orc.decompress(want, len(total)) == total, (name, variant)
    # append-only twice fails (test_broccoli.rs:168-182, #[should_panic]): the second file is not catable
    ukko, fox = [p for n, p, _ in _reference_cases() if n == "append then cat"][0]
    files = [orc.stream_compress(x, [(Q, 5), (W, 22), (APP, 1)])[0] for x in (ukko, fox)]
    r, _ = orc.concat(files, bs=2)
    assert r >= 124
    assert _concat(L, files, 1 << 20, 1 << 20, want_error=True) == r
    assert _concat(L, files, 1, 2, want_error=True) == r
    # bytes without a BroccoliNewBrotliFile in front of them are judged like the start of a first file, never forwarded as they are
    st = L.BroccoliCreateInstance()
    plain_stream = orc.compress(ukko, 5, 22)  # not catable, not appendable
    buf = ctypes.create_string_buffer(1 << 16)
    avail_in, avail_out = ctypes.c_size_t(len(plain_stream)), ctypes.c_size_t(1 << 16)
    L.BroccoliConcatStreaming(ctypes.byref(st), ctypes.byref(avail_in), plain_stream, ctypes.byref(avail_out), buf)
    taken = (1 << 16) - avail_out.value
    avail_out = ctypes.c_size_t((1 << 16) - taken)
    r = L.BroccoliConcatFinished(ctypes.byref(st), ctypes.byref(avail_out), ctypes.cast(ctypes.addressof(buf) + taken, ctypes.c_char_p))
    L.BroccoliDestroyInstance(st)
    # (what the oracle's BroCatli makes of the same bytes as a first file: a lone stream of any kind passes, concat/mod.rs:567-608)
    assert (r, buf.raw[:(1 << 16) - avail_out.value]) == orc.concat([plain_stream], bs=4096)
    # ... and with a catable file behind it: whatever the oracle's BroCatli makes of the junction (it looks for the two set bits
    # of an empty last meta-block at the end of the first file, concat/mod.rs:277-330)
    for first_file in (plain_stream, orc.compress(synth.alice()[:5000], 5, 22), orc.compress(bytes(100), 6, 18)):
        files = [first_file, _enc(fox, 18)]
        r, want = orc.concat(files, bs=2)
        got = _concat(L, files, 1000, 1000, want_error=True)
        assert got == (want if r == 0 else r), (kind, r)


def test_reference_concat_cases_against_the_oracle_product_library():
    _check_reference_cases("product")


def test_reference_concat_cases_against_the_oracle_emulation_build():
    _check_reference_cases("emu")


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
