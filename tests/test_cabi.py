"""The C ABI (include/brotli_mi355x.h) exercised through the Python host mirror of c/py/brotli.py.
CPU run: the same cabi.cpp/encoder.cpp linked against the emulation seam.  GPU run (-m gpu): the product
library.  Expected bytes always come from the oracle."""
import os
import re
import subprocess
import sys

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
Q, W, SH, MAGIC = 1, 2, 5, 169


def _load(kind):
    import importlib
    if kind == "emu":
        import emu
        emu.build()
        # the python package refuses to import without the product .so: load the module file directly
        spec = importlib.util.spec_from_file_location("brotli_mi355x_emu", os.path.join(ROOT, "rust-brotli_amd", "brotli_mi355x", "__init__.py"))
        mod = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(mod)
        except ImportError:
            pass  # product library not built: only Library(path) is used below
        return mod.Library(os.path.join(emu.EMU_DIR, "libbrotli_emu.so"))
    import brotli_mi355x
    return brotli_mi355x.default_library()


def _suite(lib):
    a = synth.alice()
    # one-shot == oracle one-shot
    assert lib.compress(a, 5, 22) == orc.compress(a, 5, 22)
    assert lib.compress(b"", 5, 22) == b"\x06"
    # quality 10 through the one-shot entry is the reference's "9.5": it runs at quality 9 (encode.rs:1468-1481)
    assert lib.compress(a, 10, 22) == orc.compress(a, 10, 22) == orc.compress(a, 9, 22)
    assert lib.BrotliEncoderVersion() == 0x01000f01
    # streaming in 4096-byte writes == CompressorWriter feeding pattern of the oracle
    e = lib.encoder(params=[(Q, 5), (W, 22)])
    for i in range(0, len(a), 4096):
        e.write(a[i:i + 4096])
    got = e.finish()
    assert not e.set_parameter(Q, 6)  # refused once started (encode.rs:289-295)
    e.close()
    assert got == orc.writer_compress(a, 5, 22, chunk=4096)
    # multi: 1 thread is the plain single stream, N threads = compress_multi + BroCatli
    d = open(os.path.join(HERE, "golden", "random_then_unicode"), "rb").read()
    assert bytes(lib.BrotliCompress(d, {Q: 5, MAGIC: 1}, 1)) == orc.stream_compress(d, [(Q, 5), (MAGIC, 1)])[0]
    for nt in (2, 3, 8):
        got = bytes(lib.BrotliCompress(d, {Q: 5, MAGIC: 1}, nt))
        assert got == orc.compress_multi(d, [(Q, 5), (MAGIC, 1)], nt)
        assert orc.decompress(got, len(d)) == d
    got = bytes(lib.BrotliCompress(d, {Q: 5, MAGIC: 1}, 3))
    assert len(got) <= 144325  # src/bin/test_threading.rs:99-102
    # favor_cpu_efficiency (threading/mod.rs:456-542): the shared hasher equals the per-shard one while every shard starts
    # inside the window (asserted by the reference itself, encode.rs:1249-1268), so the stream is compress_multi's; with a
    # shard beyond the window the reference's table is inconsistent and the call is refused
    FAVOR = 171
    assert bytes(lib.BrotliCompress(d, {Q: 5, W: 22, FAVOR: 1}, 3)) == orc.compress_multi(d, [(Q, 5), (W, 22)], 3)
    with pytest.raises(Exception):
        lib.BrotliCompress(d, {Q: 5, W: 17, FAVOR: 1}, 3)
    # tiny inputs with more threads than bytes (src/bin/test_threading.rs:111-150)
    for data in (b"", b"x", b"xy", b"xyz", d[:17]):
        for nt in (2, 5):
            assert bytes(lib.BrotliCompress(data, {Q: 5, MAGIC: 1}, nt)) == orc.compress_multi(data, [(Q, 5), (MAGIC, 1)], nt)
    # a shard long enough for its ring buffer to wrap (lgwin 17: 256 KiB ring, 128 KiB of prefix): the reference applies the
    # "no match across the end of the custom dictionary" rule to ring indices, i.e. again in every revolution
    # (backward_references/mod.rs:42-54, 1712-1724); found by the fuzz sweep
    for seed in (2, 6):
        t = synth.markov_text(700000, seed)
        assert bytes(lib.BrotliCompress(t, {Q: 5, W: 17}, 2)) == orc.compress_multi(t, [(Q, 5), (W, 17)], 2)
    # chunk + concat path used for multi-GPU == in-process multi
    chunks = [lib.compress_chunk(d, len(d), t, 4, {Q: 5, W: 22}) for t in range(4)]
    assert lib.concat_chunks(chunks) == orc.compress_multi(d, [(Q, 5), (W, 22)], 4)
    # work pool API
    pool = lib.BrotliEncoderCreateWorkPool(4)
    assert bytes(lib.BrotliEncoderCompressWorkPool(pool, a, {Q: 5, W: 22}, 4)) == orc.compress_multi(a, [(Q, 5), (W, 22)], 4)
    lib.BrotliEncoderDestroyWorkPool(pool)
    # BROTLI_OPERATION_FLUSH (CompressorWriter::flush): every piece byte-identical to the reference's, and the
    # concatenation a valid stream
    def flushed(data, cuts, params):
        e = lib.encoder(params=list(params))
        pieces, pos = [], 0
        for c in cuts:
            pieces.append(e.flush(data[pos:c]))
            pos = c
        e.write(data[pos:])
        pieces.append(e.finish())
        e.close()
        return pieces
    for params in ([(Q, 5), (W, 22)], [(Q, 7), (W, 20)]):
        for cuts in ([50000, 100000], [0, 70000, 70000], [1], [65536], [65535, 65537], [3, 10, 100], [len(a)]):
            got = flushed(a, cuts, params)
            assert got == orc.stream_with_flushes(a, params, cuts), (params, cuts)
            assert orc.decompress(b"".join(got), len(a)) == a
    mix = synth.mixed(700000)
    cuts = [100000, 300000, 300001, 650000]
    assert flushed(mix, cuts, [(Q, 5), (W, 22)]) == orc.stream_with_flushes(mix, [(Q, 5), (W, 22)], cuts)
    # BROTLI_OPERATION_EMIT_METADATA: pending input flushed, then the payload as a metadata block (decoders skip it)
    def with_ops(data, ops, params):
        e = lib.encoder(params=list(params))
        pieces, pos = [], 0
        for item in ops:
            if isinstance(item, tuple):
                c, meta = item
                if c > pos:
                    e.write(data[pos:c])
                pieces.append(e.emit_metadata(meta))
            else:
                c = item
                pieces.append(e.flush(data[pos:c]))
            pos = c
        e.write(data[pos:])
        pieces.append(e.finish())
        e.close()
        return pieces
    big = bytes(range(256)) * 300
    for ops in ([(50000, b"hello metadata")], [(0, b"x" * 300)], [(70000, b""), 100000], [(65536, big), (65536, b"second")],
                [10000, (20000, b"mm"), 30000, (30000, b"")], [(len(a), b"tail")]):
        got = with_ops(a, ops, [(Q, 5), (W, 22)])
        assert got == orc.stream_with_flushes(a, [(Q, 5), (W, 22)], ops), ops
        assert orc.decompress(b"".join(got), len(a)) == a
    # two metadata blocks whose output is fetched with BrotliEncoderTakeOutput (available_out = 0): the second one used to
    # be refused because the "metadata still draining" state was only cleared by BrotliEncoderCompressStream
    import ctypes
    e = lib.encoder(params=[(Q, 5), (W, 22)])
    got = bytearray()
    for meta in (b"first block", b"second block"):
        buf = ctypes.create_string_buffer(meta, len(meta))
        avail_in, next_in = ctypes.c_size_t(len(meta)), ctypes.c_void_p(ctypes.addressof(buf))
        avail_out, next_out = ctypes.c_size_t(0), ctypes.c_void_p(0)
        assert lib.lib.BrotliEncoderCompressStream(e._s, 3, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                                   ctypes.byref(next_out), None), meta
        while lib.lib.BrotliEncoderHasMoreOutput(e._s):
            n = ctypes.c_size_t(0)
            ptr = lib.lib.BrotliEncoderTakeOutput(e._s, ctypes.byref(n))
            got += ctypes.string_at(ptr, n.value)
    got += e.finish()
    e.close()
    assert bytes(got) == b"".join(orc.stream_with_flushes(b"", [(Q, 5), (W, 22)], [(0, b"first block"), (0, b"second block")]))
    # BrotliEncoderSetCustomDictionary fixes the hasher parameters on the spot (ensure_initialized inside set_custom_dictionary,
    # encode.rs:1234): with no size hint set, an input of more than 1 MiB still gets the 14-bit bucket table; found by
    # the fuzz sweep
    big = synth.markov_text(1200000, 9)
    for params in ([(Q, 5), (W, 22)], [(Q, 5), (W, 22), (SH, 1200000)]):
        e = lib.encoder(params=list(params), dictionary=a[:50000])
        e.write(big)
        got = e.finish()
        e.close()
        assert got == orc.stream_compress(big, params, prefix=a[:50000], continuation=False)[0], params
    # the static-dictionary throttle (matches < lookups >> 7, mod.rs:1957-1960) tips over with the LAST lookup of a chain:
    # the exact counters say "off" at the next segment's entry although no chain has seen it off yet (the resolver used
    # to go round in circles here; found by the fuzz sweep)
    x = open(os.path.join(HERE, "golden", "dictionary_off_at_segment_boundary.bin"), "rb").read()
    e = lib.encoder(params=[(Q, 8), (W, 17)])
    e.write(x)
    got = e.finish()
    e.close()
    assert got == orc.writer_compress(x, 8, 17, chunk=100000)
    # flushing a stream with a custom dictionary (round 3; tests/test_streaming_dictionary.py has the sweep) ...
    e = lib.encoder(params=[(Q, 5)], dictionary=a[:1000])
    pieces = [e.flush(a[1000:2000])]
    e._stream(2, a[2000:9000])
    pieces.append(bytes(e._out))
    e.close()
    assert pieces == orc.stream_with_flushes(a[1000:9000], [(Q, 5)], [1000], dictionary=a[:1000])
    # ... also with nothing to search yet (the hash table then holds the bare dictionary, and the context bytes of the first
    # meta-block still read as 0)
    e = lib.encoder(params=[(Q, 6), (SH, 5 << 20)], dictionary=a[:1000])
    pieces = [e.flush(b"")]
    e._stream(2, a[1000:60000])
    pieces.append(bytes(e._out))
    e.close()
    assert pieces == orc.stream_with_flushes(a[1000:60000], [(Q, 6), (SH, 5 << 20)], [0], dictionary=a[:1000])
    # unsupported parameters fail loudly instead of silently doing something else
    import brotli_mi355x as _m  # noqa: F401
    with pytest.raises(Exception):
        lib.BrotliCompress(a, {Q: 5, W: 17, 171: 1}, 9)  # (favor_cpu_efficiency with shards beyond the window: refused with a message, never routed elsewhere)
    # qualities 0 .. 4 through the one-shot entry (round 4: the fragment compressors and the BasicHasher family are on the device):
    # the oracle's bytes
    for q in (0, 1, 2, 3, 4):
        assert lib.compress(a[:60000], q, 22) == orc.compress(a[:60000], q, 22), q
    # quality 11 through the one-shot entry (round 4: the Zopfli path is on the device): the oracle's bytes
    assert lib.compress(a[:40000], 11, 22) == orc.compress(a[:40000], 11, 22)


def test_cabi_emulation():
    _suite(_load("emu"))


@pytest.mark.gpu
def test_cabi_gpu():
    lib = _load("gpu")
    assert "gfx950" in lib.device_name()
    _suite(lib)


def test_library_exports_every_declared_symbol():
    """the product .so must load and export every function include/brotli_mi355x.h declares"""
    import ctypes
    header = open(os.path.join(ROOT, "include", "brotli_mi355x.h")).read()
    names = set(re.findall(r"\b(Brotli(?:Encoder|Mi355x)[A-Za-z0-9]+)\s*\(", header))
    assert len(names) >= 26
    so = os.path.join(ROOT, "rust-brotli_amd", "libbrotli_mi355x.so")
    if not os.path.exists(so):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    lib = ctypes.CDLL(so)
    for n in sorted(names):
        assert hasattr(lib, n), n
    assert lib.BrotliEncoderVersion() == 0x01000f01
    lib.BrotliEncoderMaxCompressedSize.restype = ctypes.c_size_t
    lib.BrotliEncoderMaxCompressedSize.argtypes = [ctypes.c_size_t]
    assert lib.BrotliEncoderMaxCompressedSize(0) == 17
    lib.BrotliEncoderMaxCompressedSizeMulti.restype = ctypes.c_size_t
    lib.BrotliEncoderMaxCompressedSizeMulti.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    assert lib.BrotliEncoderMaxCompressedSizeMulti(0, 1) == 25  # src/ffi/multicompress/test.rs:258


@pytest.mark.gpu
def test_direct_fills_switch_gpu():
    """round 6: small zero fills are noted and written by one launch in front of the next operation on the stream
    (device_runtime.hip); BROTLI_MI355X_DIRECT_FILLS=1 (read once per process) issues every fill by itself as before -- the same
    streams either way"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import test_cabi, synth, orc\n"
            "lib = test_cabi._load('gpu')\n"
            "for d, q, w in ((synth.alice(), 5, 22), (synth.mixed(400000, 3), 9, 20), (synth.markov_text(300000, 2), 2, 22), (synth.alice()[:70000], 0, 18)):\n"
            "    assert lib.compress(d, q, w) == orc.compress(d, q, w), (q, w)\n"
            "print('ok')\n") % os.path.dirname(os.path.abspath(__file__))
    for env_extra in ({"BROTLI_MI355X_DIRECT_FILLS": "1"}, {}):
        env = dict(os.environ, **env_extra)
        p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0 and b"ok" in p.stdout, p.stderr.decode()[-2000:]


def _small_calls_gate(lib, threads_n, calls):
    """round 6: one-shot calls of up to 1 MiB take one of four seats (cabi.cpp, SmallCallGate): twelve threads of small calls at
    qualities 5 / 2 / 0 beside one thread of larger ones (not gated) all finish, every stream the oracle's"""
    import threading
    small = [(synth.alice(), 5), (synth.markov_text(90000, 5), 2), (synth.mixed(70000, 4), 0), (synth.alice()[:3000], 9)]
    big = synth.markov_text((1 << 20) + 4097, 6)
    want_small = [orc.compress(d, q, 22) for d, q in small]
    want_big = orc.compress(big, 5, 22)
    errors = []

    def work(i):
        try:
            for c in range(calls):
                d, q = small[(i + c) % len(small)]
                if lib.compress(d, q, 22) != want_small[(i + c) % len(small)]:
                    errors.append(("small", i, c))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def work_big():
        try:
            for _ in range(2):
                if lib.compress(big, 5, 22) != want_big:
                    errors.append("big")
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(threads_n)] + [threading.Thread(target=work_big)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a thread is still waiting for a seat"
    assert not errors, errors


def test_small_calls_take_seats_emulation():
    _small_calls_gate(_load("emu"), threads_n=12, calls=3)


@pytest.mark.gpu
def test_small_calls_take_seats_gpu():
    _small_calls_gate(_load("gpu"), threads_n=12, calls=12)


@pytest.mark.gpu
def test_concurrent_calls_from_threads():
    """independent BrotliEncoderCompress calls from several host threads (one HIP stream per thread) must not disturb
    each other"""
    import threading
    lib = _load("gpu")
    inputs = [synth.alice(), synth.markov_text(1 << 20, 7), synth.mixed(600000, 3), synth.random_bytes(300000, 11),
              synth.markov_text(3 << 20, 9), bytes(500000)]
    expected = [orc.compress(d, 5, 22) for d in inputs]
    results = [None] * len(inputs)
    errors = []

    def work(i):
        try:
            for _ in range(3):
                results[i] = lib.compress(inputs[i], 5, 22)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert results == expected


def _multi_shards_in_flight(lib, repeats=4):
    """BrotliEncoderCompressMulti keeps several shards in flight on helper threads (cabi.cpp, ShardWorkers): the stitched
    stream must be the oracle's compress_multi whatever order the shards finish in -- repeated, because a race between
    the helpers would only show now and then"""
    cases = [(synth.mixed(1000003, seed=5), 5, 22, 8), (synth.markov_text(700001, 3), 6, 20, 5), (synth.mixed(300000, seed=9), 9, 20, 8)]
    for data, q, w, shards in cases:
        want = orc.compress_multi(data, [(Q, q), (W, w)], shards)
        for _ in range(repeats):
            assert bytes(lib.BrotliCompress(data, {Q: q, W: w}, shards)) == want


def test_multi_shards_in_flight_emulation():
    _multi_shards_in_flight(_load("emu"), repeats=2)  # (the CPU suite is sized to minutes; the GPU run repeats four times)


@pytest.mark.gpu
def test_multi_shards_in_flight_gpu():
    _multi_shards_in_flight(_load("gpu"))


def _run_reference_client(path, extra_env=None, args=()):
    """tests/clients/multiexample = the reference's own c/multiexample.c (with the reference's own headers) linked against the product
    library: eleven BrotliEncoderCompressWorkPool (or, with NO_WORK_POOL, BrotliEncoderCompressMulti) calls -- quality 10 + Q9_5 with 10
    threads, 11 + Q9_5 with 11, then qualities 3..11 with as many threads -- each decoded again by the client through libbrotlidec.
    Returns the eleven compressed sizes it prints."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clients", "multiexample")
    if not os.path.exists(exe):
        pytest.skip("tests/clients/multiexample is built where /root/reference is present (__graft_entry__.build)")
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "16")
    env.update(extra_env or {})
    p = subprocess.run([exe] + ([path] if path else []) + list(args), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, (p.returncode, p.stderr.decode()[-2000:])
    sizes = [int(m) for m in re.findall(r"reduced to (\d+)", p.stdout.decode())]
    assert len(sizes) == 11, p.stdout.decode()
    return sizes


def _oracle_sizes_of_the_reference_client(data, lgwin=None):
    """what the same eleven calls give on the oracle (c/arg.h: SIZE_HINT = input size first, then the command line, then the override)"""
    sizes = []
    for i in range(1, 12):
        params = [(5, len(data))]
        if lgwin is not None:
            params.append((2, lgwin))
        q = i
        if i < 3:
            q = i + 9
            params.append((150, 1))  # BROTLI_PARAM_Q9_5
        params.append((1, q))
        sizes.append(len(orc.compress_multi(data, params, min(max(q, 1), 16))))
    return sizes


@pytest.mark.gpu
@pytest.mark.parametrize("no_work_pool", [False, True])
def test_reference_client_multiexample_gpu(tmp_path, no_work_pool):
    """the drop-in claim at the link level: the reference's own client, compiled against the reference's own headers, runs on the
    product library, every stream decodes (the client aborts otherwise), and every size equals the oracle's for the same call"""
    env = {"NO_WORK_POOL": "1"} if no_work_pool else {}
    # its built-in text (a few hundred bytes: shards of some forty bytes at eleven threads)
    example = (b"Mary had a little lamb. Its fleece was white as snow.\n"
               b"And every where that Mary went, the lamb was sure to go.\n"
               b"It followed her to school one day which was against the rule.\n"
               b"It made the children laugh and play to see a lamb at sch00l!\n\n\n\n"
               b"0 1 1 2 3 5 8 13 21 34 55 89 144 233 377 610 987 1597 2584 4181 6765\n"
               b"\x11\x99\x2f\xfc\xfe\xef\xff\xd8\xfd\x9c\x43"
               b"Additional testing characters here\x00")
    assert _run_reference_client(None, env) == _oracle_sizes_of_the_reference_client(example)
    # alice29.txt at lgwin 22
    f = tmp_path / "alice29.txt"
    f.write_bytes(synth.alice())
    assert _run_reference_client(str(f), env, ["-w22"]) == _oracle_sizes_of_the_reference_client(synth.alice(), 22)
