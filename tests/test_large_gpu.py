"""Full-size checks on the GPU (BASELINE.json sizes) through properties that do not need the oracle to run for
minutes: the stream must decode (system libbrotlidec) to exactly the input, incompressible input must come out as
stored meta-blocks, and compressing the same input twice must give the same bytes (the parse is a fixed point, not a
race)."""
import ctypes
import hashlib

import pytest

import orc
import synth

pytestmark = pytest.mark.gpu
Q, W, SH = 1, 2, 5


@pytest.fixture(scope="module")
def L():
    import gpulib
    return gpulib.lib()


def _encode(L, data):
    import emu
    return emu.encode_stream(L, data, [(Q, 5), (W, 22), (SH, min(len(data), 1 << 30))])


def test_text_256MiB_round_trip(L):
    block = synth.markov_text(64 << 20)
    data = block * 4  # repeats lie beyond the 4 MiB window: every copy has to be found inside its own 64 MiB
    out, st = _encode(L, data)
    assert len(out) < len(data) // 3
    assert hashlib.sha256(orc.decompress(out, len(data))).digest() == hashlib.sha256(data).digest()
    out2, _ = _encode(L, data)
    assert out2 == out


def test_random_1GiB_is_stored(L):
    data = synth.random_bytes(64 << 20) * 16
    out, st = _encode(L, data)
    # every meta-block stored raw (should_compress, encode.rs:1325-1354): 8 MiB blocks, a few header bytes each
    overhead = len(out) - len(data)
    print("overhead", overhead, "metablocks", st["metablocks"], st["uncompressed_metablocks"])
    assert 0 < overhead < 16 * st["metablocks"] + 64
    assert st["uncompressed_metablocks"] == st["metablocks"]
    assert hashlib.sha256(orc.decompress(out, len(data))).digest() == hashlib.sha256(data).digest()


def test_zero_fill_512MiB(L):
    data = bytes(512 << 20)
    out, st = _encode(L, data)
    assert len(out) < 4096
    assert orc.decompress(out, len(data)) == data


def test_quality9_256MiB_round_trip(L):
    """BASELINE.json configs[2]: 256 MiB enwik-style corpus at quality 9 (H9)"""
    import emu
    block = synth.markov_text(32 << 20)
    parts = []
    for i in range(8):  # page-like records with varying headers so that the eight copies are not byte-identical
        parts.append(b"<page><title>%d</title><id>%d</id><text>" % (i * 7919, i) + block[i:] + b"</text></page>\n")
    data = b"".join(parts)[:256 << 20]
    out, st = emu.encode_stream(L, data, [(Q, 9), (W, 22), (SH, len(data))])
    assert len(out) < len(data) // 3
    assert hashlib.sha256(orc.decompress(out, len(data))).digest() == hashlib.sha256(data).digest()


def test_silesia_like_multi_8_shards(L):
    """BASELINE config 4 at 1/64 of its size on one GPU: BrotliEncoderCompressMulti with 8 shards over a Silesia-like mix
    (SURVEY 8d C4), byte-identical to the oracle's compress_multi (src/enc/threading/mod.rs:333-453) and round-tripping."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-brotli_amd"))
    import brotli_mi355x
    data = synth.silesia_like(64 << 20, min_segment=256 << 10, max_segment=8 << 20)
    params = {1: 5, 2: 22}
    got = bytes(brotli_mi355x.default_library().BrotliCompress(data, params, 8))
    assert got == orc.compress_multi(data, [(1, 5), (2, 22)], 8)
    assert orc.decompress(got, len(data)) == data
