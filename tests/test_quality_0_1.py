"""Qualities 0 and 1: compress_fragment (one pass) and compress_fragment_two_pass as device code
(rust-brotli_amd/csrc/fragment_device.h: one wavefront per fragment), behind what BrotliEncoderCompressStream does instead of the
ring-buffer path at these qualities (BrotliEncoderCompressStreamFast, encode.rs:2706-2861; fragment_stream.cpp): every call's input
is cut into fragments of at most 1 << lgwin bytes, each on a fresh hash table; the stream carries the open byte and, at quality 0,
the command prefix code.

Catable streams -- and with them streams that were given a custom dictionary and the shards of BrotliEncoderCompressMulti, which
are catable at these qualities (encode.rs:1237-1241) -- go through the reference's ring-buffer path after all: encode_data hands
every input block of 1 << lgwin bytes to the same fragment compressors (encode.rs:2335-2389), behind the two raw first bytes of
a catable stream and the magic-number block (FragmentRingCompress).

The reference holds no exact size for these qualities; everything is byte identity with the oracle (oracle/orc_fragment.c, the
quality 0 / 1 branch of encode_data in orc_encode.c), which tests/test_oracle_vs_libbrotlienc.py holds against libbrotlienc 1.0.9
for the same qualities (the catable forms exist in the Rust sources only).  One call is refused: a metadata block behind input
on a catable stream -- the reference does not return from it.  CPU: the emulation build; -m gpu: the product library."""
import glob
import os

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
Q, W, SH, LARGE = 1, 2, 5, 6


def _one_shot(lib, small):
    """BrotliEncoderCompress: one FINISH with everything -> fragments of 1 << lgwin bytes"""
    a = synth.alice()
    cases = [("alice", a, 22), ("alice w16 (three fragments)", a, 16), ("alice w10 (149 fragments)", a, 10), ("alice w24", a, 24),
             ("zeros", bytes(300000), 22), ("random (stored raw)", synth.random_bytes(200000), 22), ("3 bytes", b"abc", 22),
             ("15 bytes (below the input margin)", a[:15], 22), ("16 bytes", a[:16], 22), ("empty", b"", 22),
             ("markov 1.5 MiB w20 (two fragments)", synth.markov_text(3 << 19, 5), 20), ("mixed 1 MiB w18", synth.mixed(1 << 20, 3), 18)]
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))):
        cases.append((os.path.basename(f), open(f, "rb").read(), 22))
    if not small:
        cases += [("silesia-like 3 MiB", synth.silesia_like(3 << 20, 4), 22), ("stretches 2 MiB w17", synth.stretches(2 << 20, 9), 17),
                  ("random_then_unicode", open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read(), 22),
                  ("text 5 MiB w22 (a block past the 1 MiB merge limit)", synth.markov_text(5 << 20, 6), 22)]
    bad = []
    for name, d, w in cases:
        for q in (0, 1):
            got = lib.compress(d, q, w)
            if got != orc.compress(d, q, w) or orc.decompress(got, len(d)) != d:
                bad.append((name, q, w))
    assert not bad, bad


def _streams(lib, small):
    """the stream operations: every PROCESS / FLUSH / FINISH call is compressed on the spot, fragment boundaries are call boundaries;
    FLUSH seals the open byte with an empty metadata block; EMIT_METADATA writes its header behind the open byte"""
    d = synth.mixed(300000 if small else 900000, 7)
    n = len(d)
    for q, w, cuts in ((0, 22, [n // 3, 2 * n // 3 + 1]), (1, 16, [0, 70000, 70001, n // 2]), (0, 18, [5]), (1, 22, [n // 5, n // 2]), (0, 10, [1000, 1001, 3000])):
        params = [(Q, q), (W, w)]
        e = lib.encoder(params=params)
        pieces, last = [], 0
        for c in cuts:
            pieces.append(e.flush(d[last:c]))
            last = c
        e.write(d[last:])
        pieces.append(e.finish())
        e.close()
        assert pieces == orc.stream_with_flushes(d, params, cuts), (q, w, cuts)
    # metadata blocks between pieces, extra parameters (none of them changes the fragment path), the writer pattern
    for q in (0, 1):
        params = [(Q, q), (W, 20), (168, 1), (169, 1), (172, 1), (SH, 12345)]
        ops = [(0, b"first"), n // 4, (n // 2, b"M" * 300), n // 2 + 10]
        e = lib.encoder(params=params)
        pieces, pos = [], 0
        for item in ops:
            if isinstance(item, tuple):
                if item[0] > pos:
                    e.write(d[pos:item[0]])
                pieces.append(e.emit_metadata(item[1]))
                pos = item[0]
            else:
                pieces.append(e.flush(d[pos:item]))
                pos = item
        e.write(d[pos:])
        pieces.append(e.finish())
        e.close()
        assert pieces == orc.stream_with_flushes(d, params, ops), q
        for chunk in (1000, 65536, 100000):
            e = lib.encoder(params=[(Q, q), (W, 22)])
            for i in range(0, n, chunk):
                e.write(d[i:i + chunk])
            got = e.finish()
            e.close()
            assert got == orc.writer_compress(d, q, 22, chunk=chunk), (q, chunk)
            assert orc.decompress(got, n) == d


def _catable(lib, small):
    """the ring-buffer path of qualities 0 / 1: catable streams (two raw first bytes, magic-number block, blocks of 1 << lgwin bytes),
    streams with a custom dictionary (ignored at these qualities but for turning catable on), shards, flushes"""
    a = synth.alice()
    d = synth.mixed(300000 if small else 700000, 3)
    for q in (0, 1):
        for data, params in ((a, [(Q, q), (W, 22), (167, 1)]), (a, [(Q, q), (W, 16), (167, 1), (169, 1)]), (d, [(Q, q), (W, 18), (167, 1)]),
                             (b"x", [(Q, q), (167, 1)]), (b"", [(Q, q), (167, 1)]), (b"ab", [(Q, q), (167, 1)]),
                             (a, [(Q, q), (W, 18), (167, 1), (173, 1)]), (a, [(Q, q), (LARGE, 1), (W, 26), (167, 1), (169, 1)])):
            e = lib.encoder(params=params)
            e.write(data)
            got = e.finish()
            e.close()
            assert got == orc.stream_compress(data, params)[0], (q, params)
        cuts = [1, 100000, 100001, len(d) * 2 // 3]
        params = [(Q, q), (W, 17), (167, 1)]
        e = lib.encoder(params=params)
        pieces, last = [], 0
        for c in cuts:
            pieces.append(e.flush(d[last:c]))
            last = c
        e.write(d[last:])
        pieces.append(e.finish())
        e.close()
        assert pieces == orc.stream_with_flushes(d, params, cuts), q
        assert orc.decompress(b"".join(pieces), len(d)) == d
        for nt in (2, 4, 7):
            got = bytes(lib.BrotliCompress(d, {Q: q, W: 20}, nt))
            assert got == orc.compress_multi(d, [(Q, q), (W, 20)], nt), (q, nt)
            assert orc.decompress(got, len(d)) == d
        e = lib.encoder(params=[(Q, q), (W, 22)], dictionary=a[:5000])
        e.write(a[5000:])
        got = e.finish()
        e.close()
        assert got == orc.stream_compress(a[5000:], [(Q, q), (W, 22)], prefix=a[:5000], continuation=False)[0], q


def _refusals(lib):
    a = synth.alice()
    for q in (0, 1):
        e = lib.encoder(params=[(Q, q), (W, 22), (167, 1)])
        try:
            e.write(a)
            with pytest.raises(Exception):
                e.emit_metadata(b"never returns in the reference")
            # the refusal leaves the stream as it was: it can still be finished
            assert e.finish() == orc.stream_compress(a, [(Q, q), (W, 22), (167, 1)])[0]
        finally:
            e.close()
        # ... and the reference DOES return while everything received so far goes out as the raw first bytes of the catable
        # stream (they are what moves last_flush_pos_, encode.rs:2283-2333): a metadata block behind 0, 1 or 2 bytes of input
        for upto in (0, 1, 2):
            params = [(Q, q), (W, 18), (167, 1)]
            d = a[:50000]
            e = lib.encoder(params=params)
            try:
                pieces = []
                if upto:
                    e.write(d[:upto])
                pieces.append(e.emit_metadata(b"meta" * 5))
                e.write(d[upto:])
                pieces.append(e.finish())
            finally:
                e.close()
            assert pieces == orc.stream_with_flushes(d, params, [(upto, b"meta" * 5)]), (q, upto)


def test_catable_streams_dictionaries_and_shards_emu():
    import test_cabi
    _catable(test_cabi._load("emu"), small=False)


@pytest.mark.gpu
def test_catable_streams_dictionaries_and_shards_gpu():
    import test_cabi
    _catable(test_cabi._load("gpu"), small=True)


def _several_fragments(lib):
    """12 MiB at lgwin 22: three fragments of one call, each on a fresh table, the command code of quality 0 handed on"""
    d = synth.markov_text(12 << 20, 33)
    for q in (0, 1):
        got = lib.compress(d, q, 22)
        assert got == orc.compress(d, q, 22), q
        assert orc.decompress(got, len(d)) == d


def test_calls_larger_than_a_batch_emu():
    """one call's fragments go through the device in batches (256 MiB of input; here turned down to 200 KB): same stream"""
    import subprocess
    import sys
    child = ("import sys; sys.path.insert(0, %r); import orc, synth, test_cabi; lib = test_cabi._load('emu'); d = synth.mixed(1500000, 5)\n"
             "for q, w in ((0, 16), (1, 16), (0, 10), (1, 22)):\n    assert lib.compress(d, q, w) == orc.compress(d, q, w), (q, w)\nprint('ok')" % HERE)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, BROTLI_MI355X_FRAGMENT_BATCH="200000"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


_SIDE_BY_SIDE_CHILD = """
import random, sys
sys.path.insert(0, %r)
import orc, synth, test_cabi
lib = test_cabi._load(%r)
rng = random.Random(%d)
text, rnd = synth.markov_text(1 << 20, 3), synth.random_bytes(1 << 20)
for it in range(%d):
    # text, random bytes and zeros in pieces of all sizes: stored and compressed meta-blocks alternate inside and across fragments,
    # so that fragments start at every bit phase and end byte aligned or not
    parts, total = [], rng.choice([3000, 20000, 100000, 400000])
    while sum(map(len, parts)) < total:
        k, ln = rng.choice([0, 0, 1, 1, 2]), rng.choice([1, 7, 100, 900, 1024, 1500, 5000, 40000, 140000])
        off = rng.randrange(0, (1 << 20) - ln)
        parts.append(text[off:off + ln] if k == 0 else (rnd[off:off + ln] if k == 1 else bytes(ln)))
    d = b"".join(parts)[:total + rng.randrange(0, 50)]
    w = rng.choice([10, 10, 11, 12, 14, 16, 17, 18])
    for q in (0, 1):
        assert lib.compress(d, q, w) == orc.compress(d, q, w), (it, q, w, len(d))
print("ok")
"""


def _side_by_side(which, seed, cases, force_again):
    """Round 5: the fragments of a call are compressed side by side, each from bit 0 of a slot of its own, quality 0 in two passes
    (the command code a fragment leaves behind does not depend on the one it came in with -- checked by BROTLI_MI355X_SELFTEST),
    and joined afterwards; BROTLI_MI355X_TEST_FRAGMENT_AGAIN sends every fragment off phase 0 through the one-by-one path that
    the join falls back to where the raw fall-back hangs on the padding of an alignment."""
    import subprocess
    import sys
    env = dict(os.environ, BROTLI_MI355X_SELFTEST="1")
    if force_again:
        env["BROTLI_MI355X_TEST_FRAGMENT_AGAIN"] = "1"
    r = subprocess.run([sys.executable, "-c", _SIDE_BY_SIDE_CHILD % (HERE, which, seed, cases)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_fragments_side_by_side_emu():
    _side_by_side("emu", 1, 40, False)
    _side_by_side("emu", 2, 25, True)


@pytest.mark.gpu
def test_fragments_side_by_side_gpu():
    _side_by_side("gpu", 11, 40, False)
    _side_by_side("gpu", 12, 20, True)


def test_several_fragments_emu():
    import test_cabi
    _several_fragments(test_cabi._load("emu"))


@pytest.mark.gpu
def test_several_fragments_gpu():
    import test_cabi
    _several_fragments(test_cabi._load("gpu"))


def test_one_shot_emu():
    import test_cabi
    _one_shot(test_cabi._load("emu"), small=False)


def test_streams_emu():
    import test_cabi
    _streams(test_cabi._load("emu"), small=False)


def test_what_is_refused_emu():
    import test_cabi
    _refusals(test_cabi._load("emu"))


@pytest.mark.gpu
def test_one_shot_gpu():
    import test_cabi
    _one_shot(test_cabi._load("gpu"), small=True)


@pytest.mark.gpu
def test_streams_gpu():
    import test_cabi
    _streams(test_cabi._load("gpu"), small=True)
