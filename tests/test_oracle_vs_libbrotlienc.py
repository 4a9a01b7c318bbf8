"""The CPU oracle against an INDEPENDENT implementation: Google's C encoder (system libbrotlienc.so.1, 1.0.9).

rust-brotli is a port of that encoder and cannot be built here, so no stream of the reference itself pins the q5..q8
paths byte for byte.  This file closes most of that gap: the oracle's streams are compared with
libbrotlienc's, and every difference between the two is traced to one of four places where the Rust source
(/root/reference) and the C source 1.0.9 really differ.  Each of them exists in the oracle as a TEST SWITCH that is off
by default (= rust-brotli's behaviour, the one the product reproduces):

  orc_test_c109_rle_store_rule  after a copy with distance < len/4 C skips the front of StoreRange ("avoid hash
                                poisoning with RLE data"); rust-brotli stores from position + 2
                                (src/enc/backward_references/mod.rs:2516-2521)
  orc_test_c109_spree_tail      in a literal spree rust-brotli abandons the rest of the block once a sparse jump would come
                                within kMargin of its end (mod.rs:2529-2533); C clamps the jump and keeps storing
  orc_test_c109_entropy         rust-brotli looks up log2 of a histogram count with `p as u16` (truncating counts
                                >= 65 536) and sums in f32 (src/enc/bit_cost.rs:13-33); C sums in double with the true log
  orc_test_c109_hasher_choice   H6 from size_hint >= 1 MiB in C, > 4 MiB in rust-brotli; H5 bucket bits 14 below quality 7
                                at every size in C, only up to 1 MiB in rust-brotli (src/enc/encode.rs:863-893)

Qualities 2..4 (BasicHasher H2 / H3 / H4 / H54, the quality 2 / 3 meta-block writers) add two more, because rust-brotli
ports an older C version of that hasher than 1.0.9:

  orc_test_c109_basic_layout       C 1.0.9 keeps the BUCKET_SWEEP slots of a key 8 entries apart, wrapped inside the table,
                                   and takes the last-distance / single-slot candidate only if it beats the score so far;
                                   rust-brotli uses slots key .. key + BUCKET_SWEEP - 1 and takes them unconditionally
                                   (src/enc/backward_references/mod.rs:322-327, 376-440)
  orc_test_c109_basic_store_range  rust-brotli's StoreRangeOptBasic files four positions at a time under the sweep slot of
                                   the first one and stores masked positions (mod.rs:254-285); C stores one by one

Quality 0 (the one-pass fragment compressor) differs only in ShouldMergeBlock, which rust-brotli evaluates in f32
(orc_test_c109_merge_block_double; no decision on the inputs below flips because of it).  Quality 1 (the two-pass
fragment compressor) has one:

  orc_test_c109_two_pass_min_match with a 2^15-entry table C still matches 4 bytes (B <= 15), rust-brotli switches to 6
                                   (`$table_bits < 15`, src/enc/compress_fragment_two_pass.rs:723)

A fifth one shows only on inputs longer than the ring buffer with an H5 hasher (here: lgwin 18 and more than 512 KiB):

  orc_test_c109_adv_store_range rust-brotli's StoreRangeOptBatch writes MASKED positions into the H5 bucket rings
                                (mod.rs:1163-1232); past the first revolution of the ring buffer they end the bucket walk
                                of FindLongestMatch.  C stores absolute positions.

With these switched to the C behaviour the oracle is BYTE-IDENTICAL to libbrotlienc on every input below, at qualities
5..8 and several window sizes: hashing (H5 with 14 / 15 bucket bits, H6), bucket rings, the static dictionary, lazy
matching, the distance cache, command coding, context modelling, greedy block splitting, histogram optimisation, Huffman
trees, context maps and bit emission of the restatement all agree with an implementation written by other people.
Quality 9 is not comparable (rust-brotli's H9 does not exist in C); it is pinned by the reference's own 51 737 KAT."""
import contextlib
import ctypes
import os

import pytest

import brotli_parse
import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
SWITCHES = ("orc_test_c109_rle_store_rule", "orc_test_c109_spree_tail", "orc_test_c109_entropy", "orc_test_c109_hasher_choice",
            "orc_test_c109_adv_store_range")
LOW_QUALITY_SWITCHES = SWITCHES + ("orc_test_c109_basic_layout", "orc_test_c109_basic_store_range",
                                   "orc_test_c109_two_pass_min_match", "orc_test_c109_merge_block_double")


@pytest.fixture(scope="module")
def genc():
    try:
        lib = ctypes.CDLL("libbrotlienc.so.1")
    except OSError:
        pytest.skip("no system libbrotlienc.so.1")
    lib.BrotliEncoderCompress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p,
                                          ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]

    def compress(data, quality, lgwin):
        cap = len(data) + len(data) // 2 + 1024
        out = ctypes.create_string_buffer(cap)
        n = ctypes.c_size_t(cap)
        assert lib.BrotliEncoderCompress(quality, lgwin, 0, len(data), data, ctypes.byref(n), out) == 1
        return out.raw[:n.value]
    return compress


@contextlib.contextmanager
def c109_behaviour(*names):
    cells = [ctypes.c_int.in_dll(orc.lib(), n) for n in names]
    try:
        for c in cells:
            c.value = 1
        yield
    finally:
        for c in cells:
            c.value = 0


def fixture_files():
    out = [("alice29.txt", synth.alice()), ("random_then_unicode", open(os.path.join(HERE, "golden", "random_then_unicode"), "rb").read())]
    small = os.path.join(HERE, "golden", "small")
    for name in sorted(os.listdir(small)):
        out.append((name, open(os.path.join(small, name), "rb").read()))
    return out


def test_identical_without_any_switch_where_the_sources_agree(genc):
    """random_then_unicode has no run-length copies, no literal spree that reaches a block end and no symbol more
    frequent than 65 535: rust-brotli's and C's encoders must agree on it as they stand."""
    data = open(os.path.join(HERE, "golden", "random_then_unicode"), "rb").read()
    for quality in (5, 6, 7, 8):
        assert orc.compress(data, quality, 22) == genc(data, quality, 22), quality


def test_alice29_parts_ways_at_the_rle_store_rule(genc):
    """Without switches alice29 differs by one byte at q5 (52 808 vs 52 809).  The parser locates the first command on
    which the two streams disagree; the match the oracle uses there starts at a position that C never put into its hash
    table: it lies in the part of an earlier copy with distance < len / 4 that C's StoreRange skips."""
    data = synth.alice()
    a, b = orc.compress(data, 5, 22), genc(data, 5, 22)
    assert (len(a), len(b)) == (52808, 52809)
    pa, pb = brotli_parse.parse(a), brotli_parse.parse(b)
    assert pa["output"] == data and pb["output"] == data
    where = brotli_parse.first_difference(pa, pb)
    assert where[1] == "command", where
    index, mine, theirs = where[2]
    insert_len, copy_len, _, distance, pos = mine
    assert theirs[4] == pos  # the same parse up to here
    source = pos + insert_len - distance  # where the oracle's match comes from
    cmds = pa["metablocks"][where[0]]["cmds"][:index]
    covering = [c for c in cmds if c[1] and c[4] + c[0] + 2 <= source < c[4] + c[0] + c[1]]
    assert covering, "the source position is not inside an earlier copy"
    ins, length, _, dist, cpos = covering[-1]
    start = cpos + ins
    assert dist < (length >> 2), (dist, length)
    assert source < start + length - 4 * dist, "C would have stored this position as well"
    with c109_behaviour("orc_test_c109_rle_store_rule"):
        assert orc.compress(data, 5, 22) == b


@pytest.mark.parametrize("quality", [5, 6, 7, 8])
def test_identical_to_libbrotlienc_modulo_the_four_source_differences(genc, quality):
    inputs = fixture_files() + [
        ("markov 1 MiB", synth.markov_text(1 << 20)),          # size_hint == 1 MiB: H5 here, H6 in C
        ("markov 2 MiB", synth.markov_text(2 << 20)),
        ("markov 6 MiB", synth.markov_text(6 << 20)),          # H6 in both; command symbols more frequent than 65 535
        ("mixed 3 MiB", synth.mixed(3 << 20)),                 # text, binary records, zero fill, hex, random
        ("stretches 5 MiB", synth.stretches(5 << 20)),         # literal sprees across block ends
        ("silesia-like 9 MiB", synth.silesia_like(9 << 20, 3, 64 << 10, 2 << 20)),  # two meta-blocks, 13-context literal model
    ]
    with c109_behaviour(*SWITCHES):
        for name, data in inputs:
            for lgwin in ((22, 18) if len(data) > (1 << 20) else (22, 24, 18, 17)):
                mine, theirs = orc.compress(data, quality, lgwin), genc(data, quality, lgwin)
                assert mine == theirs, (name, quality, lgwin, len(mine), len(theirs))
    # and the switches are off again: the oracle is back to rust-brotli's behaviour
    assert len(orc.compress(synth.alice(), 5, 22)) == 52808


@pytest.mark.parametrize("quality", [0, 1, 2, 3, 4])
def test_low_qualities_identical_to_libbrotlienc_modulo_the_source_differences(genc, quality):
    """Quality 0: compress_fragment (one pass, the command code carried from block to block, merged blocks).
    Quality 1: compress_fragment_two_pass (CreateCommands, the 128-symbol command code, BuildAndStoreHuffmanTreeFast) behind
    the ring-buffer-less stream path of encode.rs:2706-2861.
    H2 (quality 2, store_meta_block_fast with BrotliBuildAndStoreHuffmanTreeFast and the static command / distance
    codes), H3 (quality 3, store_meta_block_trivial), H4 and -- from size_hint 1 MiB -- H54 (quality 4, greedy block
    splitter without context modelling); lgblock 14 and the delayed-symbol flush rule below quality 4."""
    inputs = fixture_files() + [
        ("markov 1 MiB", synth.markov_text(1 << 20)),
        ("markov 2 MiB", synth.markov_text(2 << 20)),
        ("mixed 3 MiB", synth.mixed(3 << 20)),
        ("stretches 5 MiB", synth.stretches(5 << 20)),
    ]
    with c109_behaviour(*LOW_QUALITY_SWITCHES):
        for name, data in inputs:
            for lgwin in ((22, 18) if len(data) > (1 << 20) else (22, 24, 18, 16)):
                mine, theirs = orc.compress(data, quality, lgwin), genc(data, quality, lgwin)
                assert mine == theirs, (name, quality, lgwin, len(mine), len(theirs))
    # without the switches: rust-brotli's own behaviour, which must at least decode
    for name, data in inputs[:4]:
        assert orc.decompress(orc.compress(data, quality, 22), len(data)) == data


def _genc_stream(data, params, cuts, write_size=0):
    """libbrotlienc's stream API driven like orc.stream_with_flushes: FLUSH after the bytes up to each offset in `cuts` (or
    EMIT_METADATA for an (offset, bytes) entry), PROCESS in pieces of write_size in between, FINISH at the end"""
    lib = ctypes.CDLL("libbrotlienc.so.1")
    lib.BrotliEncoderCreateInstance.restype = ctypes.c_void_p
    lib.BrotliEncoderCreateInstance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.BrotliEncoderSetParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    lib.BrotliEncoderDestroyInstance.argtypes = [ctypes.c_void_p]
    lib.BrotliEncoderHasMoreOutput.argtypes = [ctypes.c_void_p]
    lib.BrotliEncoderIsFinished.argtypes = [ctypes.c_void_p]
    lib.BrotliEncoderCompressStream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p),
                                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    s = lib.BrotliEncoderCreateInstance(None, None, None)
    for k, v in params:
        assert lib.BrotliEncoderSetParameter(s, k, v)
    cap = len(data) + len(data) // 4 + 65536
    out = ctypes.create_string_buffer(cap)
    inbuf = ctypes.create_string_buffer(data, len(data) if len(data) else 1)
    base = ctypes.addressof(inbuf)
    avail_out = ctypes.c_size_t(cap)
    next_out = ctypes.c_void_p(ctypes.addressof(out))
    total = ctypes.c_size_t(0)

    def call(op, address, count):
        avail_in = ctypes.c_size_t(count)
        next_in = ctypes.c_void_p(address)
        while True:
            assert lib.BrotliEncoderCompressStream(s, op, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                                   ctypes.byref(next_out), ctypes.byref(total))
            if avail_in.value == 0 and not lib.BrotliEncoderHasMoreOutput(s):
                break

    pieces, pos, done_out = [], 0, 0
    for item in list(cuts) + [len(data)]:
        cut, meta = (item if isinstance(item, tuple) else (item, None))
        final = meta is None and cut == len(data) and len(pieces) == len(cuts)
        if write_size:
            while cut - pos > write_size:
                call(0, base + pos, write_size)
                pos += write_size
        if meta is None:
            call(2 if final else 1, base + pos, cut - pos)
        else:
            if cut > pos:
                call(0, base + pos, cut - pos)
            mbuf = ctypes.create_string_buffer(bytes(meta), max(1, len(meta)))
            call(3, ctypes.addressof(mbuf), len(meta))
        pos = cut
        produced = cap - avail_out.value
        pieces.append(out.raw[done_out:produced])
        done_out = produced
    assert lib.BrotliEncoderIsFinished(s)
    lib.BrotliEncoderDestroyInstance(s)
    return pieces


@pytest.mark.parametrize("quality", [0, 1, 2, 3, 4, 5, 6, 8])
def test_stream_operations_identical_to_libbrotlienc(genc, quality):
    """The stream state machine -- PROCESS in pieces, FLUSH (the byte-alignment block), EMIT_METADATA, FINISH, the size hint taken
    from the first call, at qualities 0 / 1 the fragments cut at call boundaries -- against libbrotlienc's, driven with the same
    calls (the source differences switched to the C behaviour as above)."""
    Q, W = 1, 2
    text = synth.markov_text(900000, 41)
    mixed = synth.mixed(700000, 42)
    cases = [
        (text, 22, [300000, 600001], 0),
        (text, 18, [1, 65536, 65537], 4096),
        (mixed, 20, [0, 250000], 100003),
        # (a ONE-byte metadata block is written with MSKIPBYTES = 0 by both encoders -- write_metadata_header, encode.rs:2545-2575, like
        # C 1.0.9 -- and no decoder takes the stream; the lengths here are 40, 0 and 2)
        (mixed, 17, [(100000, b"metadata" * 5), 400000, (400000, b""), (650000, b"xy")], 65536),
        (synth.alice(), 22, [], 1000),
        # (lgwin <= 16 at qualities 5 .. 8: C has the H40 / H41 / H42 hashers, rust-brotli falls back to H6 -- not comparable)
        (synth.alice(), 16 if quality < 5 else 17, [(0, b"before anything"), 70000], 0),
    ]
    with c109_behaviour(*LOW_QUALITY_SWITCHES):
        for data, lgwin, cuts, write_size in cases:
            params = [(Q, quality), (W, lgwin)]
            mine = orc.stream_with_flushes(data, params, cuts, write_size=write_size)
            theirs = _genc_stream(data, params, cuts, write_size=write_size)
            assert [len(x) for x in mine] == [len(x) for x in theirs], (quality, lgwin, cuts, write_size)
            assert mine == theirs, (quality, lgwin, cuts, write_size)
            assert orc.decompress(b"".join(mine), len(data)) == data
        # a small sweep of random call sequences on top
        rng = synth.XorShift(1000 + quality)
        pools = (text, mixed, synth.alice())
        for _ in range(30):
            pool = pools[rng.next() % 3]
            n = 1 + rng.next() % min(len(pool) - 1, 300000)
            o = rng.next() % (len(pool) - n)
            data = pool[o:o + n]
            lgwin = ((16, 18, 20, 22) if quality < 5 else (17, 18, 20, 22))[rng.next() % 4]
            cuts = []
            for c in sorted(rng.next() % (n + 1) for _ in range(rng.next() % 4)):
                cuts.append((c, b"m" * (2 + rng.next() % 50)) if rng.next() % 3 == 0 else c)
            write_size = (0, 1000, 4096, 65536, 100003)[rng.next() % 5]
            params = [(Q, quality), (W, lgwin)]
            mine = orc.stream_with_flushes(data, params, cuts, write_size=write_size)
            theirs = _genc_stream(data, params, cuts, write_size=write_size)
            assert mine == theirs, (quality, lgwin, n, o, cuts, write_size)
