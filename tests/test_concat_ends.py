"""BrotliMi355xConcatChunkEnds (stitching from the first / last bytes of every chunk, bodies copied by the caller) against
BrotliMi355xConcatChunks on the same chunks; chunks come from the host emulation build, the expected stream from the
oracle's compress_multi (src/enc/threading/mod.rs:565-660, src/concat/mod.rs:274-608)."""
import ctypes
import importlib.util
import os

import pytest
import torch

import emu
import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
Q, W = 1, 2


def _load():
    emu.build()
    pkg = os.path.join(ROOT, "rust-brotli_amd", "brotli_mi355x")
    spec = importlib.util.spec_from_file_location("brotli_mi355x_multi", os.path.join(pkg, "multi.py"))
    multi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(multi)
    spec = importlib.util.spec_from_file_location("brotli_mi355x_emu", os.path.join(pkg, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError:
        pass
    so = os.path.join(emu.EMU_DIR, "libbrotli_emu.so")
    return multi, mod.Library(so), multi.ShardEncoder(ctypes.CDLL(so))


def _stitch_from_ends(library, chunks):
    n = len(chunks)
    heads = torch.zeros((n, 8), dtype=torch.uint8)
    tails = torch.zeros((n, 8), dtype=torch.uint8)
    for i, c in enumerate(chunks):
        k = min(8, len(c))
        heads[i, :k] = torch.frombuffer(bytearray(c[:k]), dtype=torch.uint8)
        tails[i, :k] = torch.frombuffer(bytearray(c[len(c) - k:]), dtype=torch.uint8)
    out = torch.full((sum(len(c) for c in chunks) + 64,), 0xAA, dtype=torch.uint8)
    total, bodies = library.concat_chunk_ends(heads, tails, [len(c) for c in chunks], out.data_ptr(), out.numel())
    for c, (dst, src, count) in zip(chunks, bodies):
        if count:
            out[dst:dst + count] = torch.frombuffer(bytearray(c[src:src + count]), dtype=torch.uint8)
    return bytes(out[:total].numpy())


@pytest.mark.parametrize("lgwin,size,world", [(22, 500000, 3), (18, 300001, 4), (22, 40, 2)])
def test_stitch_from_chunk_ends(lgwin, size, world):
    multi, library, enc = _load()
    data = synth.mixed(size, seed=5 + world)
    params = [(Q, 5), (W, lgwin)]
    chunks = []
    for r in range(world):
        lo, start, end = multi.shard_window(len(data), r, world, lgwin)
        chunks.append(enc.encode(multi.shard_params(params, r), data[lo:start], data[start:end], end - start, False))
    want = library.concat_chunks(chunks)
    assert want == orc.compress_multi(data, params, world)
    assert _stitch_from_ends(library, chunks) == want


def test_stitch_from_ends_of_tiny_chunks():
    _, library, _ = _load()
    # catable / appendable empty-ish streams of a few bytes each (shorter than the 8-byte ends)
    chunks = [orc.compress_multi(b"ab", [(Q, 5), (W, 22)], 1)]
    assert _stitch_from_ends(library, chunks) == library.concat_chunks(chunks)
