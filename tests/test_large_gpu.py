"""Oracle identity on the GPU at the sizes BASELINE.json states.

The CPU oracle needs minutes for these inputs, so it is not run here: tools/freeze_large_hashes.py ran it once over the
same seeded inputs (tests/large_cases.py) and froze sha256(stream) in tests/golden/large_hashes.json.  The speculative
parse behaves differently at these sizes (4 KiB segments from 128 MiB on, more rounds, several meta-blocks per stream,
shards of 128 MiB), so identity at small sizes does not carry over -- these tests are the parity gate for configs[1..4]."""
import hashlib
import json
import os

import pytest

import large_cases
import orc
import synth

pytestmark = pytest.mark.gpu
Q, W, SH = 1, 2, 5
HERE = os.path.dirname(os.path.abspath(__file__))
FROZEN = json.load(open(os.path.join(HERE, "golden", "large_hashes.json")))


@pytest.fixture(scope="module")
def L():
    import gpulib
    return gpulib.lib()


def _input(name):
    data = large_cases.make_input(name, FROZEN)
    want = FROZEN[name]
    assert len(data) == want["input_bytes"]
    assert hashlib.sha256(data).hexdigest() == want["input_sha256"], "the input generator drifted"
    return data


def _check(name, out, data):
    want = FROZEN[name]
    if hashlib.sha256(out).hexdigest() != want["stream_sha256"]:
        # say more than "hash differs": does it at least decode, and how far off is the size?
        ok = hashlib.sha256(orc.decompress(out, len(data))).digest() == hashlib.sha256(data).digest()
        raise AssertionError("%s: stream differs from the oracle's (%d bytes, oracle %d; decodes to the input: %s)" %
                             (name, len(out), want["stream_bytes"], ok))
    assert len(out) == want["stream_bytes"]


def _one_shot(L, name):
    import emu
    case = large_cases.CASES[name]
    data = _input(name)
    out, st = emu.encode_stream(L, data, [(Q, case["quality"]), (W, case["lgwin"]), (SH, min(len(data), 1 << 30))])
    _check(name, out, data)
    return data, out, st


def test_config2_text_64MiB(L):
    _one_shot(L, "c2_text_64MiB_q5")


def test_text_256MiB(L):
    data, out, st = _one_shot(L, "text_256MiB_q5")
    # the parse is a fixed point, not a race: the same bytes again
    import emu
    out2, _ = emu.encode_stream(L, data, [(Q, 5), (W, 22), (SH, len(data))])
    assert out2 == out


def test_config3_enwik_256MiB_quality9(L):
    _one_shot(L, "c3_enwik_256MiB_q9")


def test_config5_xorshift_1GiB_is_stored(L):
    data, out, st = _one_shot(L, "c5_xorshift_1GiB_q5")
    # every meta-block stored raw (should_compress, encode.rs:1325-1354)
    assert st["uncompressed_metablocks"] == st["metablocks"]
    assert 0 < len(out) - len(data) < 16 * st["metablocks"] + 64


def test_zero_fill_1GiB(L):
    data, out, st = _one_shot(L, "zero_1GiB_q5")
    assert len(out) < 4096
    assert orc.decompress(out, len(data)) == data


def _multi(name):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-brotli_amd"))
    import brotli_mi355x
    case = large_cases.CASES[name]
    data = _input(name)
    params = {1: case["quality"], 2: case["lgwin"]}
    if case.get("hint"):
        params[5] = case["hint"]
    got = bytes(brotli_mi355x.default_library().BrotliCompress(data, params, case["shards"]))
    _check(name, got, data)


def test_config4_silesia_like_1GiB_multi_8_shards(L):
    """BASELINE config 4 at a quarter of its size on one GPU: BrotliEncoderCompressMulti with 8 shards of 128 MiB over a
    Silesia-like mix and the input size as BROTLI_PARAM_SIZE_HINT (H6 shards), byte-identical to the oracle's compress_multi
    (src/enc/threading/mod.rs:333-453)"""
    _multi("c4_silesia_1GiB_multi8_hinted")


def test_config4_without_a_size_hint_h5_shards_past_the_ring(L):
    """the same call without a size hint: H5 shards, masked ring entries from 8 MiB on (mod.rs:1163-1232) -- every shard is
    parsed by one live chain (lz77_live.h); 8 shards of 16 MiB"""
    _multi("c4_silesia_128MiB_multi8_h5")


def test_silesia_like_64MiB_multi_8_shards_against_live_oracle(L):
    """the same path with the oracle run in the test (small enough), so that a stale fixture cannot hide a drift"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-brotli_amd"))
    import brotli_mi355x
    data = synth.silesia_like(64 << 20, min_segment=256 << 10, max_segment=8 << 20)
    got = bytes(brotli_mi355x.default_library().BrotliCompress(data, {1: 5, 2: 22}, 8))
    assert got == orc.compress_multi(data, [(1, 5), (2, 22)], 8)
    assert orc.decompress(got, len(data)) == data


def test_stream_4GiB_bounded_memory(L):
    """A single 4 GiB stream (4 MiB writes with PROCESS, then FINISH; size hint 1 GiB) through BrotliEncoderCompressStream:
    encoded batch by batch with only a window of the stream kept (the encoder's input buffer never exceeds window + batch),
    through the reference's hasher reset at the 3 GiB position wrap (encode.rs:1623-1631, 1705-1710), output handed out
    during PROCESS -- and the same bytes as the oracle's stream encoder fed the same way (frozen: it needs three minutes)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-brotli_amd"))
    import brotli_mi355x
    name = "stream_4GiB_q5_w22_hinted"
    data = _input(name)
    lib = brotli_mi355x.default_library()
    e = lib.encoder(params=[(Q, 5), (W, 22), (SH, large_cases.CASES[name]["hint"])])
    h = hashlib.sha256()
    total = 0
    chunk = large_cases.CASES[name]["writer_chunk"]
    early = 0
    for i in range(0, len(data), chunk):
        e.write(data[i:i + chunk])
        if e._out:
            h.update(e._out)
            total += len(e._out)
            early = total
            e._out = bytearray()
    tail = e.finish()
    e.close()
    h.update(tail)
    total += len(tail)
    assert early > total // 2, "PROCESS did not hand out output mid-stream"
    assert total == FROZEN[name]["stream_bytes"]
    assert h.hexdigest() == FROZEN[name]["stream_sha256"]


def test_one_shot_above_2GiB(L):
    """BrotliEncoderCompress on 2.25 GiB (round 2 refused anything from 2 GiB): the call runs through the stream state machine
    in 64 MiB batches (cabi.cpp CompressOneShotStreamed, bounded device memory) and yields the reference's one-shot stream
    (size hint = (u32) input size, as encode.rs:1474 sets it)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-brotli_amd"))
    import brotli_mi355x
    name = "oneshot_2304MiB_q5_w22"
    if name not in FROZEN:
        pytest.skip("no frozen oracle hash (tools/freeze_large_hashes.py %s)" % name)
    data = _input(name)
    out = brotli_mi355x.default_library().compress(data, 5, 22)
    _check(name, out, data)
