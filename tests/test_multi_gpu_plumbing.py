"""The multi-GPU job's plumbing (device-resident shard output, RCCL collectives, pinned staging, zero-copy stitch) on ONE
GPU: a process group of size 1 on the nccl backend runs the same code path as bench.py --gpus N."""
import os
import socket
import sys

import pytest

import orc
import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
Q, W = 1, 2


def test_device_shard_job_world1():
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
    import brotli_mi355x as bm
    from brotli_mi355x import multi
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        lib = bm.default_library()
        enc = multi.ShardEncoder(lib.lib)
        data = synth.markov_text(3 << 20, 5)
        dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        job = multi.DeviceShardJob(dist, lib, enc, 0, 1, len(data))
        params = [(Q, 5), (W, 22)]
        host = lib.concat_chunks([enc.encode(multi.shard_params(params, 0), b"", dev.data_ptr(), len(data), True)])
        # two-stage pipeline: a step hands out the stream of the step before it; buffers alternate from step to step
        assert job.step(params, b"", dev.data_ptr(), len(data)) is None
        for _ in range(3):
            assert bytes(job.step(params, b"", dev.data_ptr(), len(data))) == host
        got = bytes(job.finish())
        assert job.finish() is None
        assert got == host
        assert got == lib.concat_chunks([orc.stream_compress(data, multi.shard_params(params, 0))[0]])
        assert orc.decompress(got, len(data)) == data
    finally:
        dist.destroy_process_group()


def test_bench_shard_job_path_prints_its_line():
    """bench.py as the driver launches it for N > 1 -- process group on the nccl backend, device-resident shard, gather,
    stitch, then BASELINE config 4 (cut to 1 GiB, 8 shards dealt to the ranks) -- with a single rank: the code path of the
    multi-GPU run is executed on every GPU round, and must print its one JSON line with config 4 verified."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BROTLI_MI355X_BENCH_SHARD_JOB="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config4", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["unit"] == "MB/s" and line["value"] > 0
    assert "compress_multi shard per GPU" in line["config"]["workload"]
    assert "error" not in line["config4"], line["config4"]
    assert line["config4"]["identical_to_cpu_oracle"] is True, line["config4"]
