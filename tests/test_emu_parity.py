"""CPU tests of the PRODUCT's host driver and chain / meta-block code, compiled for the host by
tests/emu (one lane per wave, serial loops instead of kernels), against the oracle.  They prove the
speculative-parse resolver, meta-block planning, stitching and C-ABI state machine without a GPU; the
`-m gpu` tests repeat the same comparisons through the HIP kernels."""
import glob
import ctypes
import os

import pytest

import emu
import orc
import synth
from cmp_lz77 import check
from cmp_stream import check_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
Q, W, SH = 1, 2, 5


@pytest.fixture(scope="module")
def L():
    return emu.lib()


@pytest.mark.parametrize("q", [5, 6, 7, 8])
def test_lz77_commands_alice(L, q):
    assert check("alice", synth.alice(), q, 22, lib=L)


@pytest.mark.parametrize("seg", [512, 1024, 65536])
def test_lz77_segment_sizes(L, seg):
    assert check("alice", synth.alice(), 5, 22, seg=seg, lib=L)


def test_lz77_h6_and_spree(L):
    assert check("markov5M", synth.markov_text(5 << 20), 5, 22, lib=L)  # size_hint > 4 MiB: H6
    assert check("random200k", synth.random_bytes(200000), 5, 22, lib=L)
    assert check("zeros200k", bytes(200000), 5, 22, lib=L)


def test_lz77_silesia_like(L):
    # literal sprees inside mixed content (the resolver predicts their phase), binary records (distance caches that
    # pass through whole segments), random data with its rare cache matches (checked, not re-parsed)
    assert check("silesia2M", synth.silesia_like(2 << 20, min_segment=16 << 10, max_segment=256 << 10), 5, 22, lib=L)
    assert check("binary1M", synth.silesia_like(1 << 20, only=60), 5, 22, lib=L)
    assert check("random2M", synth.random_bytes(2 << 20), 5, 22, lib=L)
    assert check("enwik1M", synth.enwik_like(1 << 20), 5, 22, lib=L)


def test_stream_bytes_fixtures(L):
    files = sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))) + [os.path.join(GOLDEN, "alice29.txt")]
    for f in files:
        d = open(f, "rb").read()
        b = os.path.basename(f)
        assert check_bytes(L, b, d, [(Q, 5), (W, 22), (SH, len(d))], verbose=False)
        assert check_bytes(L, b, d, [(Q, 7), (W, 22)], verbose=False)
        assert check_bytes(L, b, d, [(Q, 5), (W, 18), (SH, len(d))], verbose=False)
        assert check_bytes(L, b, d, [(Q, 5), (W, 22), (168, 1), (169, 1)], verbose=False)  # appendable + magic number
        assert check_bytes(L, b, d, [(Q, 5), (W, 22), (167, 1)], verbose=False)  # catable
        assert check_bytes(L, b, d, [(Q, 6), (W, 22), (168, 1), (172, 1)], verbose=False)  # appendable + byte_align


def test_stream_bytes_synthetic(L):
    d = synth.markov_text(3 << 20)
    assert check_bytes(L, "markov3M", d, [(Q, 5), (W, 22), (SH, len(d))], verbose=False)
    assert check_bytes(L, "mixed2M", synth.mixed(2 << 20), [(Q, 5), (W, 22)], verbose=False)
    assert check_bytes(L, "random300k", synth.random_bytes(300000), [(Q, 5), (W, 22)], verbose=False)


def test_shard_with_prefix(L):
    a = synth.alice()
    h = len(a) // 3
    assert check_bytes(L, "alice shard", a[h:2 * h], [(Q, 5), (W, 22), (167, 1), (168, 1)], prefix=a[:h], verbose=False)


def test_stored_flags_equal_reference_table(L):
    import cmp_flags
    for name, d in cmp_flags.tricky_inputs():
        assert cmp_flags.stored_flags_match(L, d), name
        assert check(name, d, 5, 22, lib=L)
        assert check_bytes(L, name, d, [(Q, 5), (W, 22), (SH, len(d))], verbose=False)


def test_quality9_h9(L):
    """quality 9 = the H9 hasher (256-deep rings, 16 cache candidates, its own scoring): the reference's own size KAT
    (alice29 at q9 / lgwin 16 -> 51 737 bytes, src/enc/encode.rs:3091) through the product's host + chain code"""
    a = synth.alice()
    out, _ = emu.encode_stream(L, a, [(Q, 9), (W, 16), (SH, len(a))])
    assert len(out) == 51737
    assert out == orc.compress(a, 9, 16)
    assert check("alice q9", a, 9, 22, lib=L)
    d = synth.markov_text(2 << 20)
    assert check("markov2M q9 w18", d, 9, 18, lib=L)  # ring buffer of 512 KiB: positions wrap four times
    import cmp_flags
    assert cmp_flags.stored_flags_match(L, bytes(2 << 20), 9, 18)
    assert check_bytes(L, "mixed1M q9", synth.mixed(1 << 20), [(Q, 9), (W, 22)], verbose=False)


def test_distance_cache_check(L):
    import check_cache_cases
    check_cache_cases.run(L)


def test_input_on_which_the_reference_fails(L):
    """A shard whose parse cuts a match to ONE byte at the end of the custom dictionary (fix_unbroken_len,
    backward_references/mod.rs:42-54, accepted through the last-distance candidate at score 2070 > 2020): the reference
    then indexes kCopyBase with GetCopyLengthCode(1) = 65535 (command.rs:91-93) and panics, its FFI reports failure.  The
    oracle flags that state and the product refuses the input with a message instead of emitting something (found by
    the fuzz sweep as a GPU memory fault)."""
    x = open(os.path.join(GOLDEN, "copy_of_length_one.bin"), "rb").read()
    with pytest.raises(orc.ReferencePanics):
        orc.stream_compress(x[64:], [(Q, 6), (W, 20)], prefix=x[:64])
    with pytest.raises(RuntimeError, match="reference encoder fails"):
        emu.encode_stream(L, x[64:], [(Q, 6), (W, 20)], prefix=x[:64])
    # the same bytes without the dictionary boundary are fine
    assert check_bytes(L, "no boundary", x, [(Q, 6), (W, 20)], verbose=False)


def test_lazy_probe_across_a_segment_boundary(L):
    """The lazy evaluation of a chain probes up to four positions ahead, i.e. into the next segment, whose chain records
    the "searched" flags of those positions -- possibly a round later.  Validation has to treat the first positions of
    a segment as searched by the chain in front regardless (found by the fuzz sweep with 256-byte segments: a stale
    candidate row went unnoticed and the fixed point was not the sequential parse)."""
    d = open(os.path.join(GOLDEN, "lazy_probe_across_segments.bin"), "rb").read()
    assert check("lazy probe", d, 5, 22, seg=256, lib=L)
    assert check("lazy probe", d, 5, 22, seg=512, lib=L)


def test_dictionary_still_alive_where_a_chain_ran_blind(L):
    """Found by the device sweep (fuzz_gpu.py 200 6, case 69): chains entered with exact "dictionary off" counters skip
    the dictionary probes and the virtual bookkeeping (mode 4).  When a later pass of the resolver moves the point where
    the throttle trips further down the stream, such a parse says nothing about what a live dictionary would have
    found and has to be redone; taking it for a "no dictionary match possible" parse gave a stream 423 bytes short."""
    pool = synth.silesia_like(6 << 20, 79, min_segment=8 << 10, max_segment=256 << 10)
    data = pool[1146783:1146783 + 2328561]
    out, _ = emu.encode_stream(L, data, [(Q, 9), (W, 17), (SH, len(data))], segment_bytes=4096)
    assert out == orc.compress(data, 9, 17)


def test_quality9_flag_changes_are_answered_by_single_searches(L):
    """qualities 6-9 (rank structures): a flag change touches the candidate lists of up to 256 later positions of its key; the
    searches of those positions are repeated (search log, lz77_recheck_searches) instead of parsing their segments again --
    same bytes, and a few dozen re-parsed chains instead of a quarter of all segments"""
    data = synth.enwik_like(4 << 20)
    out, _ = emu.encode_stream(L, data, [(Q, 9), (W, 22), (5, len(data))])
    assert out == orc.compress(data, 9, 22)
    _, st = emu.lz77_trace(L, data, quality=9, lgwin=22, segment_bytes=1024)
    nseg = len(data) // 1024
    # warm-up dry runs (384 of every 1024 bytes) + round 0 come to 1.375 chains per segment; the rest are re-parses
    # (about 60 with the re-check, about 1050 under the coarse rule)
    assert st["segments_parsed"] - 1.375 * nseg < 0.05 * nseg, st


def test_incompressible_input_converges_in_a_few_rounds(L):
    """Round counts are a property of the algorithm, not of the device, so the emulation build measures them.  Two rules of
    the host resolver keep incompressible input from being parsed over and over: the entry guessed behind a tail without
    copies carries the literal-spree phase by arithmetic (Lz77Stage::Warmup; before: every segment of random input parsed
    twice, 12 rounds at quality 5), and a segment without copies whose entry differs in the distance cache alone always
    gets its own chain (RunRounds; before: the change crept two segments per round at quality 9, 74 rounds on 4 MiB)."""
    data = synth.random_bytes(4 << 20)
    nseg = len(data) // 2048
    for q, max_rounds, max_parses in ((5, 8, 1.6), (9, 12, 4.5)):
        mbs, st = emu.lz77_trace(L, data, quality=q, lgwin=22, segment_bytes=2048)
        assert st["rounds"] <= max_rounds, (q, st)
        assert st["segments_parsed"] <= max_parses * nseg, (q, st)
    out, _ = emu.encode_stream(L, data[:1 << 20], [(Q, 9), (W, 22), (SH, 1 << 20)])
    assert out == orc.compress(data[:1 << 20], 9, 22)
    lit = synth.stretches(3 << 20)
    _, st = emu.lz77_trace(L, lit, quality=9, lgwin=22, segment_bytes=2048)
    assert st["rounds"] <= 8, st  # 29 before


def test_h5_past_the_first_ring_revolution(L):
    """The reference's StoreRangeOptBatch writes MASKED positions into the H5 rings (mod.rs:1163-1232), which end
    FindLongestMatch's bucket walk once the stream has passed one ring-buffer size (mod.rs:1763-1775) -- where the C encoder
    stores absolute positions (the oracle keeps that as a switch for its comparison with libbrotlienc).  Such inputs are
    parsed by a live chain on the reference's own bucket rings (lz77_live.h): one-shot, shards, qualities 5..8, with the
    verification pass that repeats every logged search against the rings the final flags imply."""
    import test_cabi
    os.environ["BROTLI_MI355X_LIVE_VERIFY"] = "1"
    try:
        data = synth.markov_text(1 << 20)
        out, st = emu.encode_stream(L, data, [(Q, 5), (W, 17), (SH, len(data))])
        assert out == orc.compress(data, 5, 17)
        assert st["lz77_rounds"] == 1
        cell = ctypes.c_int.in_dll(orc.lib(), "orc_test_c109_adv_store_range")
        cell.value = 1
        try:
            assert out != orc.compress(data, 5, 17)  # (the C behaviour is another stream: 871 791 bytes instead of 1 092 273 on 3 MiB)
        finally:
            cell.value = 0
        for data, w in ((synth.mixed(1 << 20), 17), (synth.markov_text(2 << 20)[:1500000], 18), (synth.stretches(1 << 20), 17), (bytes(1 << 20), 17)):
            out, _ = emu.encode_stream(L, data, [(Q, 5), (W, w), (SH, len(data))])
            assert out == orc.compress(data, 5, w)
        # incompressible input: a meta-block stored uncompressed hands on the distance cache of its start (encode.rs:1994) --
        # the chain takes should_compress's decision itself (every-13th-byte histogram, f32 entropy)
        data = synth.random_bytes(2 << 20)
        out, st = emu.encode_stream(L, data, [(Q, 5), (W, 17), (SH, len(data))])
        assert out == orc.compress(data, 5, 17)
        assert st["lz77_rounds"] == 1
        # qualities 6..8 (rings of 32..128 entries)
        data = synth.mixed(1 << 20)
        for q in (6, 7, 8):
            out, _ = emu.encode_stream(L, data, [(Q, q), (W, 17), (SH, len(data))])
            assert out == orc.compress(data, q, 17)
        lib = test_cabi._load("emu")
        t = synth.markov_text(900000, 2)
        assert bytes(lib.BrotliCompress(t, {Q: 5, W: 17}, 3)) == orc.compress_multi(t, [(Q, 5), (W, 17)], 3)
    finally:
        del os.environ["BROTLI_MI355X_LIVE_VERIFY"]


def test_live_chain_on_inputs_without_masked_entries(L):
    """BROTLI_MI355X_LIVE=1 sends any H5 / H6 input through the live chain: the same streams as the speculative parse"""
    os.environ["BROTLI_MI355X_LIVE"] = "1"
    os.environ["BROTLI_MI355X_LIVE_VERIFY"] = "1"
    try:
        for data, q, w in ((synth.alice(), 5, 22), (synth.markov_text(2 << 20), 5, 22), (synth.mixed(1 << 20), 7, 22), (synth.markov_text(5 << 20)[:4500000], 5, 22)):
            out, _ = emu.encode_stream(L, data, [(Q, q), (W, w), (SH, len(data))])
            assert out == orc.compress(data, q, w)
    finally:
        del os.environ["BROTLI_MI355X_LIVE"]
        del os.environ["BROTLI_MI355X_LIVE_VERIFY"]


def _quad_straddles_ring_end(L):
    # A compress_multi shard at lgwin 17 (H5, ring buffer of 256 KiB) whose 131 056-byte prefix + 158 925 bytes pass the end of
    # the first ring-buffer revolution inside a copy: StoreRangeOptBatch files a quad as (start & mask) + 0..3
    # (mod.rs:1163-1232), so the ONE quad that straddles that point keeps true positions for all four -- the entry 262 144
    # stays a candidate where "position >= ring size => masked" ended the bucket walk (found by the API sweep, round 3).
    import os
    blob = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quad_straddles_ring_end.bin"), "rb").read()
    pre, data = blob[:131056], blob[131056:]
    for q in (5, 6, 7, 8):
        assert check("quad straddles the ring end q%d" % q, data, q, 17, size_hint=0, catable=True, prefix=pre, lib=L)


def test_quad_that_straddles_the_end_of_the_first_ring_revolution(L):
    _quad_straddles_ring_end(L)


def test_group_state_machine_opt_in():
    """lz77_groups.h (four chains per wavefront, an opt-in experiment on the device): its state machine -- one search per step, the
    transitions written with selects, the dictionary stage with selects -- on the emulation build, one lane per group, for every launch
    that qualifies (BROTLI_MI355X_GROUPS_MIN=0; the switch is read once per process, hence the child process)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import emu, synth; from cmp_lz77 import check; L = emu.lib(); "
            "ok = check('alice', synth.alice(), 5, 22, lib=L) and check('markov5M', synth.markov_text(5 << 20), 5, 22, lib=L) and "
            "check('silesia1M', synth.silesia_like(1 << 20, min_segment=16 << 10, max_segment=256 << 10), 5, 22, lib=L) "
            "and check('random300k', synth.random_bytes(300000), 5, 22, lib=L) and check('zeros200k', bytes(200000), 5, 22, lib=L); sys.exit(0 if ok else 1)" % HERE)
    env = dict(os.environ, BROTLI_MI355X_GROUPS_MIN="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-2000:]


def test_catable_stream_recheck_at_block_starts(L):
    """Round 6, API sweep seed 62 case 213: on a catable stream (two raw first bytes: the first block starts at 2, the others at
    multiples of 64 KiB) the single-search re-check of qualities 6-9 counted its blocks from the first block's START, so the first two
    positions of every block were repeated against the end of the block in front -- a search of one or two bytes that never finds
    anything -- and a candidate that appeared there in a later round went unnoticed (one command of 22 227: insert 15 / copy 4 where the
    reference has insert 14 / copy 5).  Segments of 512 bytes, as the library cuts an input of this size."""
    d = open(os.path.join(GOLDEN, "catable_recheck_at_block_start.bin"), "rb").read()
    assert check("catable recheck", d, 6, 24, catable=True, seg=512, lib=L)
    for q in (6, 7, 8, 9):
        assert check_bytes(L, "catable recheck q%d" % q, d, [(Q, q), (W, 24), (SH, len(d)), (167, 1)], seg=512)
