"""Qualities 10 and 11 proper: the H10 binary-tree hasher (hash_to_binary_tree.rs), FindAllMatchesH10 with
BrotliFindAllStaticDictionaryMatches, the literal cost model (literal_cost.rs) and the Zopfli shortest-path parse
(backward_references/hq.rs: one pass at quality 10, two at 11) as device code (rust-brotli_amd/csrc/zopfli_device.h), behind the
quality >= 10 meta-block builder.

The reference holds two exact sizes for the path: alice29 at quality 10 -> 47 488 bytes and at quality 11 -> 46 493 bytes
(src/bin/integration_tests.rs:401-449, 4096-byte reads through roundtrip_helper).  Everything else is byte identity with the
oracle (oracle/orc_zopfli.c, pinned on the same two sizes).  CPU: the emulation build; -m gpu: the product library."""
import glob
import os

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
Q, W, SH, LARGE = 1, 2, 5, 6


def _kat(lib):
    a = synth.alice()
    for quality, size in ((10, 47488), (11, 46493)):
        params = [(Q, quality), (W, 22)]
        e = lib.encoder(params=params)
        for i in range(0, len(a), 4096):
            e.write(a[i:i + 4096])
        got = e.finish()
        e.close()
        assert len(got) == size, (quality, len(got))
        assert got == orc.reader_compress(a, params, chunk=4096)
        assert orc.decompress(got, len(a)) == a


def _cases(small):
    a = synth.alice()
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))):
        d = open(f, "rb").read()
        if small and len(d) > 20000:
            continue
        for q in (10, 11):
            yield "%s q%d" % (os.path.basename(f), q), d, [(Q, q), (W, 22), (SH, len(d))], b""
    if small:
        a = a[:60000]  # (the device slice runs at a few hundredths of a MB/s: the GPU suite keeps its inputs short)
    h = len(a) // 2
    for q in (10, 11):
        yield "alice w16 q%d" % q, a, [(Q, q), (W, 16)], b""
        if not small:
            yield "alice w18 hint q%d" % q, a, [(Q, q), (W, 18), (SH, len(a))], b""
            yield "alice catable q%d" % q, a, [(Q, q), (W, 22), (167, 1)], b""
        yield "alice appendable + magic q%d" % q, a, [(Q, q), (W, 22), (168, 1), (169, 1)], b""
        yield "alice large window q%d" % q, a, [(Q, q), (LARGE, 1), (W, 26)], b""
        yield "alice second half behind the first as dictionary q%d" % q, a[h:], [(Q, q), (W, 22), (167, 1), (168, 1)], a[:h]
        # (on the device the degenerate inputs are kept short: zero fill is ONE hash key -- a tree of depth 64 at every eighth
        # position, walked by one lane, where the block falls back to the sequential way)
        yield "random (stored raw) q%d" % q, synth.random_bytes(80000 if small else 300000), [(Q, q), (W, 22)], b""
        yield "zeros (copies past the quick step) q%d" % q, bytes(40000 if small else 700000), [(Q, q), (W, 22)], b""
        if not small:
            yield "mixed 1 MiB q%d" % q, synth.mixed(1 << 20), [(Q, q), (W, 22)], b""
            yield "stretches 1 MiB q%d" % q, synth.stretches(1 << 20, 9), [(Q, q), (W, 20)], b""
            yield "markov 1.5 MiB (six blocks) q%d" % q, synth.markov_text(3 << 19), [(Q, q), (W, 22), (SH, 3 << 19)], b""
            yield "random_then_unicode q%d" % q, open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read(), [(Q, q), (W, 22)], b""


def _identity(L, small):
    from cmp_stream import check_bytes
    bad = [name for name, data, params, prefix in _cases(small) if not check_bytes(L, name, data, params, prefix=prefix)]
    assert not bad, bad


def _multi(lib):
    """BrotliEncoderCompressMulti at quality 10: every shard behind the first is primed with its prefix as a custom dictionary
    (HasherPrependCustomDictionary stores it into the H10 trees, encode.rs:1163-1194)"""
    d = synth.markov_text(700000, 21)
    for q, nt in ((10, 3), (11, 2)):
        got = bytes(lib.BrotliCompress(d, {Q: q, W: 20}, nt))
        assert got == orc.compress_multi(d, [(Q, q), (W, 20)], nt), (q, nt)
        assert orc.decompress(got, len(d)) == d


def test_reference_kats_47488_46493_emu():
    import test_cabi
    _kat(test_cabi._load("emu"))


def test_identity_with_the_oracle_emu():
    import emu
    _identity(emu.lib(), small=False)


def test_multi_shard_emu():
    import test_cabi
    _multi(test_cabi._load("emu"))


def _flushes(lib, small=False):
    """FLUSH in the middle of a stream: the H10 trees travel from piece to piece (ZopfliCarry), moved to the next piece's text
    positions; a flush in front of any input; a custom dictionary in front of a flushed stream"""
    d = synth.mixed(600000, 7)
    a = synth.alice()
    cases = ((10, 20, [200000, 400001], d, None), (11, 18, [0, 70000, 70001, 300000], d[:380000], None),
             (10, 22, [50000], a, None), (11, 22, [1000], a[20000:90000], a[:20000]))
    if small:  # (the GPU suite: short inputs, see _cases)
        cases = ((10, 18, [30000, 60001], d[:90000], None), (11, 17, [0, 20000, 20001], a[:50000], None), (11, 22, [1000], a[20000:50000], a[:20000]))
    for q, w, cuts, data, dic in cases:
        params = [(Q, q), (W, w)]
        e = lib.encoder(params=params, dictionary=dic)
        pieces, last = [], 0
        for c in cuts:
            pieces.append(e.flush(data[last:c]))
            last = c
        e._stream(2, data[last:])
        pieces.append(bytes(e._out))
        e.close()
        want = orc.stream_with_flushes(data, params, cuts, write_size=1 << 30, dictionary=dic)
        assert [len(x) for x in pieces] == [len(x) for x in want], (q, w, cuts)
        assert pieces == want, (q, w, cuts)


_STREAMED = r"""
import sys
sys.path.insert(0, %(tests)r)
import orc, synth, test_cabi
lib = test_cabi._load(%(kind)r)
Q, W, SH = 1, 2, 5
for name, d, q, w, chunk in (("markov 2.5 MiB, quality 10, lgwin 18", synth.markov_text(5 << 19, 11), 10, 18, 65536),
                             ("mixed 1.5 MiB, quality 11, lgwin 20", synth.mixed(3 << 19, 12), 11, 20, 100003)):
    params = [(Q, q), (W, w)]
    e = lib.encoder(params=params)
    early = 0
    for i in range(0, len(d), chunk):
        e.write(d[i:i + chunk])
        early = max(early, len(e._out))
    got = e.finish()
    e.close()
    assert got == orc.reader_compress(d, params, chunk=chunk), name
    assert early > len(got) // 3, (name, "PROCESS handed nothing out", early)
    print("OK", name, len(got), early)
"""


def _streamed(kind):
    """bounded-memory streaming at qualities 10 / 11: the batch is turned down so that a few MiB go through several pieces -- every
    piece starts again at the first block of the meta-block the one in front left open, on the trees as they were there"""
    import subprocess
    import sys
    env = dict(os.environ, BROTLI_MI355X_STREAM_BATCH=str(1 << 19))
    r = subprocess.run([sys.executable, "-c", _STREAMED % dict(tests=HERE, kind=kind)], env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_flushes_emu():
    import test_cabi
    _flushes(test_cabi._load("emu"))


def test_streamed_in_pieces_emu():
    _streamed("emu")


@pytest.mark.gpu
def test_reference_kats_47488_46493_gpu():
    import test_cabi
    _kat(test_cabi._load("gpu"))


@pytest.mark.gpu
def test_identity_with_the_oracle_gpu():
    import gpulib
    _identity(gpulib.lib(), small=True)


@pytest.mark.gpu
def test_multi_shard_gpu():
    import test_cabi
    _multi(test_cabi._load("gpu"))


@pytest.mark.gpu
def test_flushes_gpu():
    import test_cabi
    _flushes(test_cabi._load("gpu"), small=True)
