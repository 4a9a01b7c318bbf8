#!/usr/bin/env python3
"""bench.py -- compress throughput of the MI355X brotli encoder hot path (quality 5, lgwin 22).

A "step" is one full compression of one batch of synthetic input: 64 MiB of English-like text
(word-bigram Markov chain over alice29.txt tokens, SURVEY 8d / BASELINE.json configs[1]) resident in HBM
when the timed region starts; the compressed stream lands in host memory.  With N > 1 (one process per
GPU, launched by torch.distributed.run) the job is the reference's compress_multi split: rank r compresses
chunk r of an N x 64 MiB stream (its chunk plus the preceding <= 4 MiB as LZ77 prefix, exactly what a
compress_part worker sees, src/enc/threading/mod.rs:337-411); the compressed chunks are gathered to rank 0
over RCCL and stitched there (BroCatli).  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (see the driver contract): metric/value/unit + `roofline` for the dominant
kernel (k_parse_segments, HBM bound, timed with HIP events on its stream inside the library) +
`cpu_baseline` (the oracle = CPU restatement of the reference path, single thread, same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))

WORKLOAD_BYTES = 64 << 20
QUALITY, LGWIN = 5, 22
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s peak


def workload(nbytes, seed):
    import synth
    cache = "/tmp/brotli_mi355x_markov_%d_%x.bin" % (nbytes, seed)
    if os.path.exists(cache) and os.path.getsize(cache) == nbytes:
        return open(cache, "rb").read()
    data = synth.markov_text(nbytes, seed)
    try:
        with open(cache + ".tmp%d" % os.getpid(), "wb") as f:
            f.write(data)
        os.replace(cache + ".tmp%d" % os.getpid(), cache)
    except OSError:
        pass
    return data


def cpu_baseline(data):
    """the oracle (port of the reference path), one thread, on the same workload"""
    import subprocess
    import orc
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_fast.so"])
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "liborc_fast.so"))
    L.orc_max_compressed_size.restype = ctypes.c_size_t
    L.orc_max_compressed_size.argtypes = [ctypes.c_size_t]
    L.orc_encoder_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p,
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_void_p]
    cap = L.orc_max_compressed_size(len(data)) + 64
    out = ctypes.create_string_buffer(cap)
    best = None
    reps = 0
    t_all = time.time()
    while reps < 2 or (time.time() - t_all < 10.0 and reps < 6):
        n = ctypes.c_size_t(cap)
        t = time.time()
        ok = L.orc_encoder_compress(QUALITY, LGWIN, 0, len(data), data, ctypes.byref(n), out, None)
        dt = time.time() - t
        assert ok
        best = dt if best is None else min(best, dt)
        reps += 1
    return {"value": round(len(data) / best / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "the whole %d MiB workload, best of %d runs, oracle built -O3 -march=native" % (len(data) >> 20, reps)}, out.raw[:n.value]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=WORKLOAD_BYTES >> 20, help="per-GPU batch size in MiB")
    ap.add_argument("--segment-bytes", type=int, default=0, help="bytes per parse chain (0 = the library's choice for the input size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    # (test aid: BROTLI_MI355X_BENCH_SHARD_JOB=1 runs the multi-GPU code path -- process group, device-resident shard,
    # gather, stitch -- with a single rank)
    shard_job = world > 1 or os.environ.get("BROTLI_MI355X_BENCH_SHARD_JOB") == "1"
    if shard_job:
        import torch.distributed as dist_mod
        dist = dist_mod
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group(backend="nccl", rank=0, world_size=1)
        else:
            dist.init_process_group(backend="nccl")  # RCCL on ROCm

    import brotli_mi355x as bm
    lib = bm.default_library()
    per_gpu = args.mib << 20
    total = per_gpu * world
    # every rank needs its chunk and the preceding window of the global stream: generate the part it needs
    seed = 0x5EED000000000002
    if not shard_job:
        data = workload(per_gpu, seed)
        prefix = b""
        chunk = data
    else:
        # the N x 64 MiB stream is the concatenation of N independently seeded texts, so that a rank only has to
        # generate its own shard and the one in front of it (for the LZ77 prefix)
        from brotli_mi355x import multi as _m
        lo, start, end = _m.shard_window(total, rank, world, LGWIN)
        assert start == rank * per_gpu and end == start + per_gpu
        chunk = workload(per_gpu, seed + rank)
        prefix = workload(per_gpu, seed + rank - 1)[lo - (start - per_gpu):] if rank else b""
        data = chunk if rank == 0 else None
    dev = torch.frombuffer(bytearray(chunk), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()

    # shard encoder: explicit prefix + device resident chunk through the library's flat entry point
    from brotli_mi355x import multi
    enc = multi.ShardEncoder(lib.lib, args.segment_bytes)
    params = [(bm.BROTLI_PARAM_QUALITY, QUALITY), (bm.BROTLI_PARAM_LGWIN, LGWIN)]
    if not shard_job:
        params.append((bm.BROTLI_PARAM_SIZE_HINT, per_gpu))  # BrotliEncoderCompress sets SIZE_HINT = input size
    st = enc.stats
    job = multi.DeviceShardJob(dist, lib, enc, rank, world, per_gpu) if shard_job else None

    def one_step():
        if not shard_job:
            comp = enc.encode(params, b"", dev.data_ptr(), len(chunk), True)
        else:
            comp = job.step(params, prefix, dev.data_ptr(), len(chunk))  # compressed shards gathered GPU to GPU, stitched on rank 0
        return comp, list(st)

    for _ in range(args.warmup):
        one_step()
    if job:
        job.finish()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    agg = {"parse_ms": 0.0, "launches": 0, "segments": 0.0, "searches": 0.0, "commands": 0.0, "rounds": 0.0}
    comp = None
    for _ in range(args.steps):
        comp, s = one_step()
        agg["parse_ms"] += s[26]
        agg["launches"] += int(s[27])
        agg["segments"] += s[28]
        agg["searches"] = s[1]
        agg["commands"] = s[2]
        agg["rounds"] += s[0]
        nseg = s[29]
        seg_bytes = s[30]
        phases = s[10:20]
        lz_ms, mb_ms, lib_ms = s[7], s[8], s[9]
    if job:
        comp = job.finish()  # the pipeline hands out step i while step i + 1 is encoded: drain it inside the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.time() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total * args.steps / elapsed / 1e6
    # algorithmic bytes of one launch of the dominant kernel (DESIGN.md "roofline"): for the bytes a launch
    # covers: input read once (1 B/B) + stored-flag written (1 B/B) + per searched position the candidate row
    # (ring depth 16 x 4 B) and its key/rank (6 B) + 16 B per command written.
    S, K = agg["searches"], agg["commands"]
    alg_full_pass = per_gpu * 2.0 + S * (16 * 4 + 6) + K * 16
    frac_per_launch = (agg["segments"] / max(1, agg["launches"])) / max(1.0, nseg)
    alg_bytes_per_launch = alg_full_pass * frac_per_launch
    avg_launch_ms = agg["parse_ms"] / max(1, agg["launches"])
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_parse.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    line = {
        "metric": "compress MB/s at q5 lgwin22", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d MiB synthetic English-like text per GPU (word-bigram Markov over alice29 tokens), quality=5, lgwin=22, "
                               "%s" % (args.mib, "one-shot BrotliEncoderCompress semantics (H6 hasher)" if not shard_job else
                                       "compress_multi chunk per GPU + RCCL gather + BroCatli stitch"),
                   "input_bytes_total": total, "compressed_bytes": len(comp), "ratio": round(total / max(1, len(comp)), 4),
                   "segment_bytes": int(seg_bytes), "lz77_rounds_per_step": agg["rounds"] / args.steps,
                   "stage_ms_last_step": {"lz77": round(lz_ms, 2), "metablock": round(mb_ms, 2), "phases": [round(x, 2) for x in phases], "library_total": round(lib_ms, 2)}},
        "roofline": {"bound": "hbm", "kernel": "k_parse_segments", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "avg_launch_ms": round(avg_launch_ms, 3), "launches_per_step": agg["launches"] / args.steps,
                     "alg_bytes_per_launch": int(alg_bytes_per_launch)},
    }
    if not args.no_cpu_baseline and not shard_job:  # (the CPU leg is reported at N = 1 only)
        sample = data
        base, ref_bytes = cpu_baseline(sample)
        line["cpu_baseline"] = base
        line["config"]["identical_to_cpu_oracle"] = (ref_bytes == comp)
    print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
