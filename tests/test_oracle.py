"""The oracle (CPU restatement of the reference) against the reference's own pins:
  * encoder_compress(q9, lgwin16, alice29) == 51737 bytes        (src/enc/encode.rs:3073-3099)
  * compress_multi size bounds on random_then_unicode             (src/bin/test_threading.rs:91-110)
  * "quality 9.5" exact sizes on random_then_unicode               (src/bin/integration_tests.rs:397-428)
  * alice29 at quality 10 / 11 == 47488 / 46493 bytes               (src/bin/integration_tests.rs:401-449)
  * round trips through an independent decoder (libbrotlidec)
plus the frozen sha256 of its own outputs (tests/golden/oracle_hashes.json)."""
import glob
import hashlib
import json
import os

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
Q, W, MAGIC = 1, 2, 169


def test_reference_kat_51737():
    a = synth.alice()
    assert hashlib.sha256(a).hexdigest().startswith("7467306e")
    c, st = orc.compress(a, 9, 16, with_stats=True)
    assert len(c) == 51737
    assert orc.decompress(c, len(a)) == a


def test_compress_multi_bounds():
    d = open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read()
    assert len(d) == 272666
    for nt, q, bound in ((3, 5, 144325), (5, 9, 139126)):
        c = orc.compress_multi(d, [(Q, q), (MAGIC, 1)], nt)
        assert len(c) <= bound
        assert orc.decompress(c, len(d)) == d


def test_reference_kat_quality_9_5():
    """src/bin/integration_tests.rs:397-428: roundtrip_helper(RANDOM_THEN_UNICODE, 10, 28, q9_5) == 130036 bytes and
    (11, 22, q9_5) == 129715 (std build, f32).  q9_5 keeps the greedy LZ77 stage (H9 at quality 10; H6 with 512-deep
    rings and 16 cache candidates at quality 11, encode.rs:834-893) and runs the quality >= 10 meta-block builder behind it:
    distance-parameter search, BrotliSplitBlock, BrotliClusterHistograms, BrotliPopulationCost (orc_hq_metablock.c).
    The helper feeds 4096-byte reads with size_hint 2 MiB through BrotliCompressCustomIo (orc.reader_compress)."""
    d = open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read()
    Q9_5, HINT = 150, 5
    for q, w, size in ((10, 28, 130036), (11, 22, 129715)):
        c = orc.reader_compress(d, [(Q, q), (Q9_5, 1), (W, w), (HINT, 2048 * 1024)], chunk=4096)
        assert len(c) == size
        assert orc.decompress(c, len(d)) == d


def test_reference_kat_quality_10_and_11():
    """src/bin/integration_tests.rs:430-449 with the constants of :408-413 (std build, f32): alice29 through
    roundtrip_helper(.., 10, 22, false) == 47488 bytes and (.., 11, 22, false) == 46493.  That is the H10 binary-tree
    hasher, FindAllMatchesH10 with BrotliFindAllStaticDictionaryMatches, the Zopfli shortest path (one pass at quality
    10, two with the command-derived cost model at 11), the literal cost model, and the quality >= 10 meta-block builder
    (orc_zopfli.c, orc_static_dict.c, orc_hq_metablock.c).  size_hint is 2 MiB for inputs above 100 000 bytes."""
    a = synth.alice()
    HINT = 5
    for q, size in ((10, 47488), (11, 46493)):
        c = orc.reader_compress(a, [(Q, q), (W, 22), (HINT, 2048 * 1024)], chunk=4096)
        assert len(c) == size
        assert orc.decompress(c, len(a)) == a


def test_round_trips_quality_10_11():
    files = sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))) + [os.path.join(GOLDEN, "random_then_unicode")]
    for f in files:
        d = open(f, "rb").read()
        for q, w in ((10, 16), (11, 22)):
            c = orc.reader_compress(d, [(Q, q), (W, w)], chunk=65536)
            assert orc.decompress(c, len(d)) == d, (f, q, w)
    # one-shot and custom-dictionary / multi-shard paths at the high qualities
    d = open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read()
    assert orc.decompress(orc.compress(d, 11, 22), len(d)) == d
    assert orc.compress(d, 10, 22) == orc.compress(d, 9, 22)  # encoder_compress runs quality 10 as 9 (encode.rs:1468)
    for q in (10, 11):
        c = orc.compress_multi(d, [(Q, q), (W, 22)], 3)
        assert orc.decompress(c, len(d)) == d


@pytest.mark.parametrize("q,w", [(5, 22), (6, 16), (7, 22), (9, 22), (9, 16), (5, 10), (5, 18)])
def test_round_trips(q, w):
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))) + [os.path.join(GOLDEN, "alice29.txt")]:
        d = open(f, "rb").read()
        assert orc.decompress(orc.compress(d, q, w), len(d)) == d, f
        assert orc.decompress(orc.writer_compress(d, q, w, chunk=4096), len(d)) == d, f


def test_empty_and_tiny_multi():
    for data in (b"", b"x", b"xy", b"xyz"):
        for nt in (1, 2, 5):
            c = orc.compress_multi(data, [(Q, 5), (MAGIC, 1)], nt)
            assert orc.decompress(c, len(data)) == data


def test_frozen_hashes():
    """sha256 of oracle outputs frozen when the oracle reproduced the reference KAT (regression guard)"""
    frozen = json.load(open(os.path.join(GOLDEN, "oracle_hashes.json")))
    a = synth.alice()
    got = {}
    for q in (5, 6, 7, 8, 9):
        got["alice29 q%d w22" % q] = hashlib.sha256(orc.compress(a, q, 22)).hexdigest()
    got["alice29 q9 w16"] = hashlib.sha256(orc.compress(a, 9, 16)).hexdigest()
    got["alice29 writer4096 q5 w22"] = hashlib.sha256(orc.writer_compress(a, 5, 22, chunk=4096)).hexdigest()
    m = synth.markov_text(1 << 20)
    got["markov1M q5 w22"] = hashlib.sha256(orc.compress(m, 5, 22)).hexdigest()
    assert got == frozen


@pytest.mark.parametrize("q", list(range(12)))
def test_every_quality_round_trips_on_mixed_content(q):
    """the oracle covers every quality of the reference encoder: 0 / 1 (fragment compressors), 2-4 (BasicHasher), 5-9
    (H5 / H6 / H9, greedy meta-blocks), 10 / 11 (H10 + Zopfli, block splitter + clustering)"""
    data = synth.mixed(768 << 10, 11) + synth.silesia_like(256 << 10, 12, 8 << 10, 64 << 10)
    for w in (16, 22):
        out = orc.stream_compress(data, [(Q, q), (W, w)])[0]
        assert orc.decompress(out, len(data)) == data, (q, w)


def test_h5_store_range_masks_positions():
    """AdvHasher::StoreRangeOptBatch (mod.rs:1163-1232, the H5 family: StoreLookahead 4) files the positions a copy covers
    four at a time -- and writes the MASKED position into the bucket.  Once the input is longer than the ring buffer
    (2 << max(lgwin, lgblock) bytes) such an entry looks further away than max_backward (backward = cur_ix - entry,
    :1763-1775) and ends the walk through its bucket: the reference loses most of its candidates there.  C stores absolute
    positions (test switch).  Below the ring size, and for H6 (lookahead 8) and H9, nothing changes."""
    import ctypes
    cell = ctypes.c_int.in_dll(orc.lib(), "orc_test_c109_adv_store_range")
    text = synth.markov_text(3 << 20)
    sizes = {}
    for c_style in (0, 1):
        cell.value = c_style
        try:
            sizes[c_style] = {w: len(orc.compress(text, 5, w)) for w in (18, 22)}
            small = orc.compress(text[:500000], 5, 18)  # shorter than the 512 KiB ring
            sizes[c_style]["small"] = len(small)
        finally:
            cell.value = 0
    assert sizes[0][22] == sizes[1][22]           # 8 MiB ring: never wraps on 3 MiB
    assert sizes[0]["small"] == sizes[1]["small"]  # no revolution yet
    assert sizes[0][18] > 1.2 * sizes[1][18]       # 512 KiB ring: 1 092 273 vs 871 791 bytes
    out = orc.compress(text, 5, 18)
    assert orc.decompress(out, len(text)) == text


def test_frozen_hashes_low_and_high_qualities():
    """streams frozen when the restatement reproduced the reference's exact sizes (qualities 10 / 11 / "9.5") and was
    byte-identical to libbrotlienc modulo the documented differences (qualities 2..4): a regression guard"""
    frozen = json.load(open(os.path.join(GOLDEN, "oracle_hashes_q2_4_q10_11.json")))
    a = synth.alice()
    d = open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read()
    HINT, Q95 = 5, 150
    got = {}
    for q in (2, 3, 4):
        got["alice29 q%d w22" % q] = hashlib.sha256(orc.compress(a, q, 22)).hexdigest()
    got["markov1M q4 w22 (H54)"] = hashlib.sha256(orc.compress(synth.markov_text(1 << 20), 4, 22)).hexdigest()
    for q, size in ((10, 47488), (11, 46493)):
        c = orc.reader_compress(a, [(Q, q), (W, 22), (HINT, 2048 * 1024)])
        got["alice29 reader4096 q%d w22 hint2M (%d)" % (q, size)] = hashlib.sha256(c).hexdigest()
    for q, w, size in ((10, 28, 130036), (11, 22, 129715)):
        c = orc.reader_compress(d, [(Q, q), (Q95, 1), (W, w), (HINT, 2048 * 1024)])
        got["random_then_unicode reader4096 q%d q9_5 w%d hint2M (%d)" % (q, w, size)] = hashlib.sha256(c).hexdigest()
    assert got == frozen


def test_log2f_restatement_matches_libm():
    """the device uses a restated glibc log2f; it must equal libm on the values FastLog2 can see"""
    import ctypes
    import numpy as np
    libm = ctypes.CDLL("libm.so.6")
    libm.log2f.restype = ctypes.c_float
    libm.log2f.argtypes = [ctypes.c_float]
    L = orc.lib()
    rng = np.random.RandomState(1)
    vals = list(range(256, 70000)) + [int(x) for x in rng.randint(256, 1 << 24, size=60000)] + [1 << 24, (1 << 24) - 1]
    for v in vals:
        assert L.orc_log2f_restated(float(v)) == libm.log2f(float(v)), v
    L.orc_log2f_check_all.restype = ctypes.c_uint32
    assert L.orc_log2f_check_all() == 0  # every integer in [256, 2^24]
