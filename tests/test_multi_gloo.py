"""The N > 1 path (one shard per process, gather, BroCatli stitch on rank 0) with world_size 2 on the gloo backend.
The shard encoder is the host emulation build (no GPU here); expected bytes come from the oracle's compress_multi
(= BrotliEncoderCompressMulti with 2 threads, src/enc/threading/mod.rs:333-411)."""
import os
import socket
import sys

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
Q, W = 1, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, lgwin, out_path, nshards=0, extra_params=()):
    import ctypes
    import importlib.util
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    import emu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    library = mod.Library(so)
    enc = multi.ShardEncoder(ctypes.CDLL(so))
    params = [(Q, 5), (W, lgwin)] + list(extra_params)
    if nshards:
        # more shards than ranks, dealt round-robin (bench.py's config 4 path): some ranks idle in the last round
        def shard_input(s):
            lo, start, end = multi.shard_window(len(data), s, nshards, lgwin)
            return data[lo:start], data[start:end], end - start, False
        stream = multi.compress_multi_over_ranks(dist, library, enc, params, len(data), nshards, rank, world, "cpu", shard_input)
    else:
        lo, start, end = multi.shard_window(len(data), rank, world, lgwin)
        stream = multi.compress_sharded(dist, library, enc, params, lgwin, data[lo:start], data[start:end], end - start,
                                        False, rank, world, "cpu")
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(stream)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lgwin,size", [(22, 600000), (18, 700001)])
def test_two_ranks_gloo(tmp_path, lgwin, size):
    import torch.multiprocessing as mp
    import emu
    emu.build()
    data = synth.mixed(size, seed=77 + lgwin)
    out = str(tmp_path / "stream.br")
    mp.spawn(_worker, args=(2, _free_port(), data, lgwin, out), nprocs=2, join=True)
    got = open(out, "rb").read()
    assert got == orc.compress_multi(data, [(Q, 5), (W, lgwin)], 2)
    assert orc.decompress(got, len(data)) == data


def test_four_ranks_five_uneven_shards(tmp_path):
    """world_size 4, five shards of uneven size (the input length is not a multiple of 5) dealt round-robin: the second
    round has one busy rank and three that contribute an empty payload to the gather"""
    import torch.multiprocessing as mp
    import emu
    emu.build()
    data = synth.mixed(1000003, seed=5)
    out = str(tmp_path / "stream.br")
    mp.spawn(_worker, args=(4, _free_port(), data, 22, out, 5), nprocs=4, join=True)
    got = open(out, "rb").read()
    assert got == orc.compress_multi(data, [(Q, 5), (W, 22)], 5)
    assert orc.decompress(got, len(data)) == data


def test_empty_shards_and_magic_number(tmp_path):
    """an input shorter than the number of shards (empty shards), and BROTLI_PARAM_MAGIC_NUMBER: only the first shard
    carries the magic block (compress_part clears it for the others, threading/mod.rs:354-357)"""
    import torch.multiprocessing as mp
    import emu
    emu.build()
    for data, nshards, extra in ((b"abc", 4, ()), (synth.markov_text(300000, 9), 3, ((169, 1),))):
        out = str(tmp_path / "stream.br")
        mp.spawn(_worker, args=(2, _free_port(), data, 22, out, nshards, extra), nprocs=2, join=True)
        got = open(out, "rb").read()
        assert got == orc.compress_multi(data, [(Q, 5), (W, 22)] + list(extra), nshards)


def test_two_ranks_quality_9_5(tmp_path):
    """the shards of a quality-9.5 job (quality 10 + BROTLI_PARAM_Q9_5: every shard's meta-blocks built by the quality >= 10
    builder) across two ranks, stitched on rank 0"""
    import torch.multiprocessing as mp
    import emu
    emu.build()
    data = synth.mixed(700000, seed=31)
    out = str(tmp_path / "stream.br")
    extra = ((Q, 10), (150, 1))  # (later settings of a parameter win)
    mp.spawn(_worker, args=(2, _free_port(), data, 22, out, 3, extra), nprocs=2, join=True)
    got = open(out, "rb").read()
    assert got == orc.compress_multi(data, [(Q, 10), (150, 1), (W, 22)], 3)
    assert orc.decompress(got, len(data)) == data
