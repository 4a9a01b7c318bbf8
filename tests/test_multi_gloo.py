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


def _worker(rank, world, port, data, lgwin, out_path):
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
    lo, start, end = multi.shard_window(len(data), rank, world, lgwin)
    stream = multi.compress_sharded(dist, library, enc, [(Q, 5), (W, lgwin)], lgwin, data[lo:start], data[start:end], end - start,
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
