"""Qualities 2, 3 and 4: the BasicHasher family H2 / H3 / H4 / H54 (backward_references/mod.rs:237-596) under the
greedy / lazy parse as device code (rust-brotli_amd/csrc/quick_device.h: one chain per stream on the reference's own hash
table), and the meta-block writers of qualities 2 and 3 (store_meta_block_fast / _trivial, brotli_bit_stream.rs:2345-2742;
metablock_fast.h) on the kernels of the greedy path.  Quality 4 goes through the greedy builder like quality 5.

The reference holds no exact size for these qualities; everything is byte identity with the oracle (oracle/orc_lz77.c,
orc_metablock.c), which tests/test_oracle_vs_libbrotlienc.py holds against libbrotlienc 1.0.9 for the same qualities.
CPU: the emulation build; -m gpu: the product library."""
import glob
import os

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
Q, W, SH, LARGE = 1, 2, 5, 6


def _cases(small):
    a = synth.alice()
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))):
        d = open(f, "rb").read()
        for q in (2, 3, 4):  # (the 13 fixtures of the reference's testdata at every quality, lgwin 18 and 22 -- on the device too)
            for w in (18, 22):
                yield "%s q%d w%d" % (os.path.basename(f), q, w), d, [(Q, q), (W, w), (SH, len(d))], b""
    h = len(a) // 2
    text = synth.markov_text(3 << 19 if not small else 300000, 5)
    for q in (2, 3, 4):
        yield "alice q%d" % q, a, [(Q, q), (W, 22)], b""
        yield "alice w16 (second ring lap: range stores file dead entries) q%d" % q, a, [(Q, q), (W, 16)], b""
        yield "alice w10 q%d" % q, a[:40000], [(Q, q), (W, 10)], b""
        yield "alice catable q%d" % q, a, [(Q, q), (W, 22), (167, 1)], b""
        yield "alice appendable + magic + byte align q%d" % q, a, [(Q, q), (W, 18), (168, 1), (169, 1), (172, 1)], b""
        yield "alice large window q%d" % q, a, [(Q, q), (LARGE, 1), (W, 26)], b""
        yield "alice second half behind the first as dictionary q%d" % q, a[h:], [(Q, q), (W, 22), (167, 1), (168, 1)], a[:h]
        yield "alice no dictionary q%d" % q, a[:50000], [(Q, q), (W, 22), (170, 1)], b""
        yield "random (stored raw, literal sprees) q%d" % q, synth.random_bytes(100000 if small else 400000), [(Q, q), (W, 22)], b""
        yield "zeros q%d" % q, bytes(100000 if small else 700000), [(Q, q), (W, 22)], b""
        # (quality 4 with a size hint of 1 MiB or more: H54 -- seven bytes hashed, 2^20 buckets, no static dictionary)
        yield "markov, size hint (H54 at quality 4) q%d" % q, text, [(Q, q), (W, 22), (SH, 3 << 19)], b""
        yield "markov w17 q%d" % q, text, [(Q, q), (W, 17)], b""
        if not small:
            yield "mixed 1 MiB q%d" % q, synth.mixed(1 << 20), [(Q, q), (W, 22)], b""
            yield "stretches 1 MiB w20 q%d" % q, synth.stretches(1 << 20, 9), [(Q, q), (W, 20)], b""
            yield "silesia-like 2 MiB w18 q%d" % q, synth.silesia_like(2 << 20, 4), [(Q, q), (W, 18)], b""
            yield "random_then_unicode q%d" % q, open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read(), [(Q, q), (W, 22)], b""


def _identity(L, small):
    from cmp_stream import check_bytes
    bad = [name for name, data, params, prefix in _cases(small) if not check_bytes(L, name, data, params, prefix=prefix)]
    assert not bad, bad


def _one_shot_and_multi(lib, small):
    a = synth.alice()
    for q in (2, 3, 4):
        assert lib.compress(a, q, 22) == orc.compress(a, q, 22), q  # BrotliEncoderCompress
    d = synth.markov_text(500000 if small else 1500000, 21)
    for q, w, nt in ((2, 20, 3), (3, 18, 5), (4, 22, 2)):
        got = bytes(lib.BrotliCompress(d, {Q: q, W: w}, nt))  # BrotliEncoderCompressMulti: shards primed with their prefix
        assert got == orc.compress_multi(d, [(Q, q), (W, w)], nt), (q, nt)
        assert orc.decompress(got, len(d)) == d


def _flushes(lib, small):
    """FLUSH / EMIT_METADATA in the middle of a stream: the hash table travels from piece to piece (QuickCarry), moved to the next
    piece's text positions; a flush in front of any input; a custom dictionary in front of a flushed stream"""
    d = synth.mixed(300000 if small else 900000, 7)
    a = synth.alice()
    n = len(d)
    cases = ((2, 20, [n // 3, 2 * n // 3 + 1], d, None), (3, 16, [0, 70000, 70001, n // 2], d, None), (4, 22, [50000], a, None),
             (4, 17, [n // 5, n // 2], d, None), (2, 22, [1000], a[20000:90000], a[:20000]), (3, 18, [30000], a[20000:], a[:20000]))
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
small = %(small)r
for name, d, q, w, chunk in (("markov, quality 2, lgwin 18", synth.markov_text((3 << 19) if small else (5 << 19), 11), 2, 18, 65536),
                             ("mixed, quality 3, lgwin 16", synth.mixed((2 << 19) if small else (3 << 19), 12), 3, 16, 100003),
                             # (a meta-block holds up to 1 << min(lgwin + 1, 24) bytes: the window is chosen so that several close inside the input)
                             ("silesia-like, quality 4", synth.silesia_like((2 << 19) if small else (4 << 19), 13), 4, 17 if small else 20, 4096)):
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


def _streamed(kind, small):
    """bounded-memory streaming: the batch is turned down so that a few MiB go through several pieces -- every piece starts again
    at the first block of the meta-block the one in front left open, on the hash table as it was there"""
    import subprocess
    import sys
    env = dict(os.environ, BROTLI_MI355X_STREAM_BATCH=str(1 << 18))
    r = subprocess.run([sys.executable, "-c", _STREAMED % dict(tests=HERE, kind=kind, small=small)], env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _past_the_ring_and_shards(lib, small):
    """the default window on one stream past its first ring-buffer lap (8 MiB at lgwin 22: from there on the quads a copy files are
    ring-buffer indices no search can reach, mod.rs:299-321), and the parallelism this path has: sixteen shards side by side"""
    d = synth.markov_text(9 << 20, 31)
    for q, w in ((3, 22),) if small else ((2, 22), (3, 22), (4, 21)):
        got = lib.compress(d, q, w)
        assert got == orc.compress(d, q, w), (q, w)
    big = synth.markov_text((16 if small else 32) << 20, 32)
    got = bytes(lib.BrotliCompress(big, {Q: 2, W: 22}, 16))
    assert got == orc.compress_multi(big, [(Q, 2), (W, 22)], 16)
    assert orc.decompress(got, len(big)) == big


def test_past_the_ring_and_shards_emu():
    import test_cabi
    _past_the_ring_and_shards(test_cabi._load("emu"), small=False)


@pytest.mark.gpu
def test_past_the_ring_and_shards_gpu():
    import test_cabi
    _past_the_ring_and_shards(test_cabi._load("gpu"), small=True)


def _paths(kind, small):
    """Round 6: the speculative path (quick_spec.h) beside its fall-backs and test switches, every one against the oracle -- the serial
    walk on the reference's own table (chosen up front, and taken over when the rounds do not settle: one launch allowed), every
    round on the pass over everything instead of the repairs, and the selftest that compares the repaired candidates with a
    rebuild from the final flags; windows inside and past the first ring-buffer lap, a dictionary in front, shards"""
    import test_cabi
    lib = test_cabi._load(kind)
    a = synth.alice()
    inputs = [("alice", a, 22), ("alice w16", a, 16), ("markov", synth.markov_text(300000 if small else 900000, 3), 22),
              ("mixed w18", synth.mixed(200000 if small else 600000, 9), 18), ("random", synth.random_bytes(150000), 22),
              ("zeros + text", bytes(70000) + a[:60000] + bytes(40000), 20), ("short", a[:1000], 22), ("63 bytes", a[:63], 22)]
    switches = ({"BROTLI_MI355X_QUICK_ROUNDS": "1"}, {"BROTLI_MI355X_QUICK_NO_INCREMENTAL": "1"}, {"BROTLI_MI355X_SELFTEST": "1"},
                {"BROTLI_MI355X_QUICK_SERIAL": "1"}, {"BROTLI_MI355X_QUICK_CAP_DIV": "100000"}, {"BROTLI_MI355X_SEGMENT_BYTES": "256"})
    want = {}
    for env in switches:
        for k, v in env.items():
            os.environ[k] = v
        try:
            for name, d, w in inputs:
                for q in (2, 3, 4):
                    if (name, q) not in want:
                        want[(name, q)] = orc.compress(d, q, w)
                    assert lib.compress(d, q, w) == want[(name, q)], (env, name, q)
            d = synth.markov_text(400000, 8)
            got = bytes(lib.BrotliCompress(d, {Q: 3, W: 20}, 4))
            assert got == orc.compress_multi(d, [(Q, 3), (W, 20)], 4), env
        finally:
            for k in env:
                del os.environ[k]


def _block_chains(kind, small):
    """Round 6: inputs whose chains do not fall into step inside their blocks (most segments still to parse again behind the second
    launch) are re-cut into one chain per block -- on candidates first, on tables of their own (QsTables::own) where that stops making
    progress; here also from the first launch on, and with a window of three blocks behind the frontier"""
    import test_cabi
    lib = test_cabi._load(kind)
    n = (640 << 10) if small else (4 << 20)
    inputs = [("stretches", synth.stretches(n, 9)), ("repeated excerpts", synth.repeated_excerpts(n, 35)), ("mixed", synth.mixed(n, 7))]
    switches = ({}, {"BROTLI_MI355X_QUICK_OWN_TABLES_FIRST": "1"}, {"BROTLI_MI355X_QUICK_OWN_TABLES_FIRST": "1", "BROTLI_MI355X_QUICK_OWN_WINDOW": "3"})
    want = {}
    for env in switches:
        for k, v in env.items():
            os.environ[k] = v
        try:
            for name, d in inputs:
                for q in (2, 3, 4):
                    if (name, q) not in want:
                        want[(name, q)] = orc.compress(d, q, 22)
                    assert lib.compress(d, q, 22) == want[(name, q)], (env, name, q)
        finally:
            for k in env:
                del os.environ[k]


def test_one_chain_per_block_emu():
    _block_chains("emu", small=True)


@pytest.mark.gpu
def test_one_chain_per_block_gpu():
    _block_chains("gpu", small=False)


def test_speculative_path_and_its_fallbacks_emu():
    _paths("emu", small=True)


@pytest.mark.gpu
def test_speculative_path_and_its_fallbacks_gpu():
    _paths("gpu", small=False)


def test_identity_with_the_oracle_emu():
    import emu
    _identity(emu.lib(), small=False)


def test_one_shot_and_multi_shard_emu():
    import test_cabi
    _one_shot_and_multi(test_cabi._load("emu"), small=False)


def test_flushes_emu():
    import test_cabi
    _flushes(test_cabi._load("emu"), small=False)


def test_streamed_in_pieces_emu():
    _streamed("emu", small=False)


@pytest.mark.gpu
def test_identity_with_the_oracle_gpu():
    import gpulib
    _identity(gpulib.lib(), small=False)  # (round 6: the whole set, as on the emulation build)


@pytest.mark.gpu
def test_one_shot_and_multi_shard_gpu():
    import test_cabi
    _one_shot_and_multi(test_cabi._load("gpu"), small=False)


@pytest.mark.gpu
def test_flushes_gpu():
    import test_cabi
    _flushes(test_cabi._load("gpu"), small=False)


@pytest.mark.gpu
def test_streamed_in_pieces_gpu():
    _streamed("gpu", small=False)
