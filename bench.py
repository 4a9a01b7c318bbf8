#!/usr/bin/env python3
"""bench.py -- compress throughput of the MI355X brotli encoder hot path (quality 5, lgwin 22).

Headline (the one JSON line the driver reads; BASELINE.json configs[1]): a "step" is one full compression of 64 MiB of
English-like text (word-bigram Markov chain over alice29.txt tokens) per GPU, resident in HBM when the timed
region starts; N = 1: one-shot BrotliEncoderCompress semantics (H6), the compressed stream stays in HBM (SURVEY 8d's
device-resident time; `output_to_pinned_host` and `e2e` give the PCIe-inclusive rates).
N > 1 (one process per GPU under torch.distributed.run): ONE stream of N x 64 MiB compressed with the reference's
compress_multi split (src/enc/threading/mod.rs:333-411): rank r encodes shard r (its 64 MiB plus the preceding <= 4 MiB
as LZ77 prefix), the shards are gathered to rank 0 over RCCL and stitched there (BroCatli).  Weak scaling: per-GPU work
is fixed.  The stitched stream is checked against the CPU oracle's compress_multi (frozen hash for N > 1).

The ONE stdout line is compact (< 4 KB, tests/test_bench_line.py: round 5's 24.6 KB line never reached the driver's record) and carries
  roofline      k_parse_segments (dominant kernel) against the 8 TB/s HBM peak, from what its launches really did:
                the kernel counts the positions walked, searches and commands of every chain (re-parses and dry runs
                included); bytes = 2 B per position walked (text read, flag written) + 64 B candidate row per search +
                16 B per command; time = HIP events around every launch.  useful_only_frac: one pass over the input / total
                kernel time; whole_step_frac: SURVEY 8d's full formula / ms_per_step.  traffic: FETCH_SIZE + WRITE_SIZE per launch
                from the newest profiles/r*_pmc_parse.json taken on the same kernel sources (null otherwise)
  cpu_baseline  the CPU oracle (port of the reference path) pinned to one core, same workload; plus Google's
                libbrotlienc 1.0.9 at the same settings as an independent column
  output_to_pinned_host_ms / e2e_pinned_ms / e2e_c_abi_pageable_ms   the same step with the PCIe copies inside the timed region
  other         digest of the side workloads, {name: [MB/s, x one CPU core of this host, identical to the oracle]}
The side workloads (N = 1) run in a CHILD process under a wall-clock budget (--extras-budget, default 60 s) after the headline is
measured; their full records -- BASELINE configs[2], [4], zero fill, the 1 GiB cut of configs[3], qualities 0-4 / 9.5 / 10 / 11, alice29
(SURVEY 8d C1), the CompressorWriter pattern without a size hint -- go to bench_extras.json and stderr, never into the stdout line.
  config4       (N > 1, or --config4) BASELINE configs[3], cut to 1 GiB: Silesia-like mix, BrotliEncoderCompressMulti with
                8 shards dealt to the N GPUs -- strong scaling, same stream for every N, verified against the frozen hash.
                (At the full 4 GiB -- 512 MiB per shard -- the REFERENCE fails on every seed that was tried)
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time


# 16 hardware queues -- eight shard workers side by side -- instead of the runtime's four: the HIP runtime reads the variable when
# it starts (torch starts it here), and the library never touches the environment itself
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# every CPU figure of this file comes from the optimised build of the oracle (oracle/liborc_fast.so: -O3 -march=native), run on THIS
# host, pinned to one core (tests/orc.py reads the variable when it loads the library)
os.environ.setdefault("ORC_FAST", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))

WORKLOAD_BYTES = 64 << 20
QUALITY, LGWIN = 5, 22
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s peak
TEXT_SEED = 0x5EED000000000002


def source_fingerprint():
    """sha256 over the library's sources (rust-brotli_amd/csrc/*, sorted): what a PMC summary under profiles/ was measured on.
    (There is no .git on the GPU box, so the commit hash itself cannot be checked there.)"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "rust-brotli_amd", "csrc")
    for name in sorted(os.listdir(src)):
        h.update(name.encode())
        h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:12]


def kernel_fingerprint(unit="lz77_kernels.hip"):
    """sha256 over one translation unit of the library: `unit` and every file of rust-brotli_amd/csrc it includes, directly or not.
    What a per-kernel measurement (PMC traffic, kernel shares) hangs on when other files of the library change: the code of the
    kernels in that unit is compiled from exactly these files."""
    import re
    src = os.path.join(ROOT, "rust-brotli_amd", "csrc")
    seen, todo = set(), [unit]
    while todo:
        name = todo.pop()
        if name in seen or not os.path.exists(os.path.join(src, name)):
            continue
        seen.add(name)
        todo += re.findall(r'^\s*#\s*include\s+"([^"/]+)"', open(os.path.join(src, name)).read(), re.M)
    h = hashlib.sha256()
    for name in sorted(seen):
        h.update(name.encode())
        h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:12]


def measured_on_these_sources(j):
    """(usable, words) for a summary under profiles/: measured on this very source tree, or at least on the same sources of the
    LZ77 kernels' translation unit (the rest of the library has changed since)"""
    here, unit = source_fingerprint(), kernel_fingerprint()
    if j.get("source_fingerprint") == here:
        return True, "same library sources as this run: fingerprint %s" % here
    if j.get("kernel_fingerprint") == unit:
        return True, ("same sources of the kernels' translation unit as this run (lz77_kernels.hip and the headers it includes: fingerprint %s); "
                      "other files of the library have changed since (whole-library fingerprint then %s, now %s)" % (unit, j.get("source_fingerprint"), here))
    return False, "measured on other library sources (fingerprint %s / kernels %s; this run: %s / %s)" % (
        j.get("source_fingerprint", "none recorded"), j.get("kernel_fingerprint", "none recorded"), here, unit)


def frozen_hashes():
    p = os.path.join(ROOT, "tests", "golden", "large_hashes.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def text_workload(nbytes):
    import synth
    return synth.markov_text(nbytes, TEXT_SEED)


def pin_to_one_core():
    """the CPU legs run on one core: pin the process to it (taskset) so that the scheduler does not move it"""
    try:
        old = os.sched_getaffinity(0)
        core = sorted(old)[-1]
        os.sched_setaffinity(0, {core})
        return old, core
    except (AttributeError, OSError):
        return None, -1


def cpu_baseline(data):
    """the oracle (port of the reference path), one pinned thread, the whole workload; libbrotlienc beside it"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_fast.so"])
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "liborc_fast.so"))
    L.orc_max_compressed_size.restype = ctypes.c_size_t
    L.orc_max_compressed_size.argtypes = [ctypes.c_size_t]
    L.orc_encoder_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p,
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_void_p]
    cap = L.orc_max_compressed_size(len(data)) + 64
    out = ctypes.create_string_buffer(cap)
    old, core = pin_to_one_core()
    best = None
    reps = 0
    t_all = time.time()
    while reps < 2 or (time.time() - t_all < 12.0 and reps < 6):
        n = ctypes.c_size_t(cap)
        t = time.time()
        ok = L.orc_encoder_compress(QUALITY, LGWIN, 0, len(data), data, ctypes.byref(n), out, None)
        dt = time.time() - t
        assert ok
        best = dt if best is None else min(best, dt)
        reps += 1
    ref = out.raw[:n.value]
    base = {"value": round(len(data) / best / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "the whole %d MiB workload, best of %d runs, oracle built -O3 -march=native, pinned to core %d" % (len(data) >> 20, reps, core)}
    try:
        G = ctypes.CDLL("libbrotlienc.so.1")
        G.BrotliEncoderCompress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p,
                                            ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
        sample = data[:16 << 20]
        gout = ctypes.create_string_buffer(len(sample) + (len(sample) >> 2) + 1024)
        gbest = None
        for _ in range(2):
            n2 = ctypes.c_size_t(len(gout))
            t = time.time()
            assert G.BrotliEncoderCompress(QUALITY, LGWIN, 0, len(sample), sample, ctypes.byref(n2), gout) == 1
            dt = time.time() - t
            gbest = dt if gbest is None else min(gbest, dt)
        base["libbrotlienc_1_0_9"] = {"value": round(len(sample) / gbest / 1e6, 2), "unit": "MB/s", "cores": 1,
                                      "sample": "first 16 MiB of the workload, best of 2, BrotliEncoderCompress(5, 22)"}
    except OSError:
        pass
    if old is not None:
        os.sched_setaffinity(0, old)
    return base, ref


class pinned_to_one_core:
    """with pinned_to_one_core(): ... -- the CPU legs run on one core and stay there"""
    def __enter__(self):
        self.old, self.core = pin_to_one_core()
        return self

    def __exit__(self, *a):
        if self.old is not None:
            os.sched_setaffinity(0, self.old)


def cpu_oracle_timed(fn, nbytes, sample):
    """fn() = the oracle call (liborc_fast.so through tests/orc.py, ORC_FAST), timed once on this host, pinned to one core"""
    with pinned_to_one_core() as pin:
        t0 = time.time()
        out = fn()
        sec = time.time() - t0
    return out, {"value": round(nbytes / sec / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port", "ms": round(sec * 1e3, 1),
                 "sample": "%s; oracle built -O3 -march=native, run on this host pinned to core %d" % (sample, pin.core)}


def cpu_oracle_sample(data, quality, lgwin, sample_bytes):
    """the large workloads: a bounded sample (the first sample_bytes through the oracle's one-shot entry at the same quality and
    window) instead of minutes of CPU time per bench run"""
    import orc
    part = data[:sample_bytes]
    _, col = cpu_oracle_timed(lambda: orc.compress(part, quality, lgwin), len(part),
                              "the first %d MiB of the workload through the oracle's one-shot entry (quality %d, lgwin %d)" % (len(part) >> 20, quality, lgwin))
    return col


def stream_roofline(kernel, input_bytes, output_bytes, sec, note):
    """roofline block of a workload whose device time is one sequential kernel per stream: the compulsory traffic (input read once,
    stream written once) over the wall time of the call -- what bounds such a path is the latency of one wavefront, which is what the
    fraction of the 8 TB/s says"""
    achieved = (input_bytes + output_bytes) / sec / 1e9 if sec > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 8),
            "traffic": None, "accounting": "compulsory bytes (input + compressed stream) / wall time of the call", "what_bounds_it": note}


def timed_steps(fn, steps, warmup, torch):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    out = None
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / steps, out


def parse_roofline(walked, searches, cmds, parse_ms, launches, row_bytes):
    """roofline block of the parse kernel from what its launches did (same accounting as the headline): bytes = 2 B per
    position walked + one candidate row per search (64 B at ring depth 16; 4 B x ring depth otherwise) + 16 B per command"""
    bytes_done = 2.0 * walked + float(row_bytes) * searches + 16.0 * cmds
    sec = parse_ms * 1e-3
    achieved = bytes_done / sec / 1e9 if sec > 0 else 0.0
    return {"bound": "hbm", "kernel": "k_parse_segments", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "avg_launch_ms": round(parse_ms / max(1, launches), 3), "launches": int(launches),
            "alg_bytes_per_launch": int(bytes_done / max(1, launches)), "row_bytes_per_search": row_bytes,
            "traffic": None, "traffic_source": "profiles/r03_<workload>.json holds the FETCH_SIZE / WRITE_SIZE passes of this workload"}


def dominant_kernel_from_profile(name):
    """what the newest rocprofv3 summary of this workload under profiles/ (tools/profile_workloads.sh) says about where its GPU time
    goes -- quoted only if it was measured on the same library sources as this run (fingerprint), like the PMC figure of the headline"""
    import glob as _glob
    for path in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % name)), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        rel = os.path.relpath(path, ROOT)
        usable, words = measured_on_these_sources(j)
        if not usable:
            return {"kernel": None, "source": "null: %s was %s" % (rel, words)}
        kernels = j.get("kernels", {})
        if not kernels:
            return None
        top = max(kernels.items(), key=lambda kv: kv[1].get("total_ms", 0.0))
        traffic = None
        c = j.get("hbm_counters_kib_raw", {}).get(top[0])
        if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            calls = max(1, c["FETCH_SIZE"].get("launches", 1))
            traffic = int((c["FETCH_SIZE"]["kib_total_both_calls"] + c["WRITE_SIZE"]["kib_total_both_calls"]) * 1024 / calls)
        return {"kernel": top[0], "share_of_gpu_time_pct": top[1].get("pct"), "avg_launch_ms": top[1].get("avg_ms"), "launches_both_calls": top[1].get("calls"),
                "traffic_bytes_per_launch": traffic, "source": "%s (rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes; %s)" % (rel, words)}
    return None


class Budget:
    """wall-clock budget of the extras: a task is skipped (and listed as such) when what is left is less than what it is expected
    to take -- the headline never waits for more than --extras-budget seconds of side workloads"""
    def __init__(self, seconds):
        self.end = time.time() + seconds
        self.skipped = []

    def left(self):
        return self.end - time.time()

    def allows(self, name, estimate_s):
        if self.left() >= estimate_s:
            return True
        self.skipped.append(name)
        return False


def large_workload(name, torch, bm, lib, enc, frozen, work_fn):
    """one of BASELINE configs[2], [4], zero fill, configs[3] cut to 1 GiB, at its stated size on this one GPU"""
    import large_cases
    case = large_cases.CASES[name]
    t0 = time.time()
    data = large_cases.make_input(name, frozen)
    gen_s = time.time() - t0
    entry = {"workload": name, "input_bytes": len(data), "quality": case["quality"], "lgwin": case["lgwin"]}
    if case.get("shards"):
        # host buffers in, host buffer out (BrotliEncoderCompressMulti C ABI): the PCIe copies are inside the time
        params = {bm.BROTLI_PARAM_QUALITY: case["quality"], bm.BROTLI_PARAM_LGWIN: case["lgwin"]}
        if case.get("hint"):
            params[bm.BROTLI_PARAM_SIZE_HINT] = case["hint"]
        # (two untimed calls for the hinted shards: the device memory pools of the eight worker threads settle with the second one -- the
        # shards of a call differ in what they need and land on other threads every time)
        sec, out = timed_steps(lambda: bytes(lib.BrotliCompress(data, params, case["shards"])), 1, 0 if not case.get("hint") else 2, torch)
        entry["residency"] = "host to host through BrotliEncoderCompressMulti (%d shards on one GPU)" % case["shards"]
        entry["hasher"] = ("H6 shards: the caller states the input size (BROTLI_PARAM_SIZE_HINT, as c/brotli.c does)" if case.get("hint") else
                           "H5 shards (no size hint): masked ring entries past the 8 MiB ring buffer (DESIGN.md section 3.5)")
    else:
        dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        params = [(bm.BROTLI_PARAM_QUALITY, case["quality"]), (bm.BROTLI_PARAM_LGWIN, case["lgwin"]),
                  (bm.BROTLI_PARAM_SIZE_HINT, min(len(data), 1 << 30))]
        cap = len(data) + len(data) // 4 + 4096
        pinned = torch.empty(cap, dtype=torch.uint8).pin_memory()  # (as in the headline step: page-locked output buffer)
        saved = (enc._out, enc._out_owner)
        enc.use_output_buffer(pinned.data_ptr(), cap, pinned)
        sec, out = timed_steps(lambda: enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=False), 2, 1, torch)
        out = bytes(out)
        enc._out, enc._out_owner = saved
        entry["residency"] = "input resident in HBM, output to page-locked host memory"
        entry["lz77_rounds"] = enc.stats[0]
        if work_fn is not None:  # (the last of the timed calls)
            work = (ctypes.c_double * 4)()
            work_fn(work)
            row_bytes = 64 if case["quality"] == 5 else 4 * (1 << {6: 5, 7: 6, 8: 7, 9: 8}[case["quality"]])
            entry["roofline"] = parse_roofline(work[0], work[1], work[2], enc.stats[26], enc.stats[27], row_bytes)
        del dev, pinned
    entry.update({"value": round(len(data) / sec / 1e6, 1), "unit": "MB/s", "ms_per_step": round(sec * 1e3, 2), "compressed_bytes": len(out),
                  "identical_to_cpu_oracle": hashlib.sha256(out).hexdigest() == frozen[name]["stream_sha256"],
                  "input_generated_in_s": round(gen_s, 1)})
    dk = dominant_kernel_from_profile(name)
    if dk:
        entry["dominant_kernel"] = dk
    # the CPU column: the oracle on THIS host, one pinned core, on a bounded sample of the same input through the one-shot entry
    # (an ESTIMATE of the ratio: the device figure covers the whole workload and, for the shard cases, another call shape)
    try:
        entry["cpu_oracle"] = cpu_oracle_sample(data, case["quality"], case["lgwin"], (8 << 20) if case["quality"] >= 9 else (32 << 20))
        entry["vs_cpu_oracle_sampled"] = round((len(data) / sec / 1e6) / entry["cpu_oracle"]["value"], 2)
    except Exception as e:
        entry["cpu_oracle"] = {"error": repr(e)}
    if name == "c4_silesia_128MiB_multi8_h5":
        entry["roofline"] = stream_roofline("k_parse_live", len(data), len(out), sec, "the reference's own rings per shard (DESIGN.md section 3.5)")
    return entry


def other_workloads(torch, bm, lib, enc, frozen, work_fn, budget):
    """the side workloads in priority order, each under the wall-clock budget (estimates in seconds from the last measured run)"""
    res = []

    def run(name, estimate_s, fn):
        if not budget.allows(name, estimate_s):
            return
        t0 = time.time()
        try:
            got = fn()
        except Exception as e:
            got = [{"workload": name, "error": repr(e)}]
        for e in (got if isinstance(got, list) else [got]):
            e.setdefault("bench_wall_s", round(time.time() - t0, 1))
            res.append(e)
            sys.stderr.write("[bench extras] " + json.dumps(e) + "\n")

    run("c1_alice29", 2, lambda: small_input_workloads(lib))
    run("q2_4_text", 8, lambda: quality_2_4_workloads(bm, lib))
    for name, est in (("c3_enwik_256MiB_q9", 6), ("c5_xorshift_1GiB_q5", 5), ("zero_1GiB_q5", 4)):
        if name in frozen:
            run(name, est, lambda name=name: large_workload(name, torch, bm, lib, enc, frozen, work_fn))
    run("q0_1_text", 10, lambda: quality_0_1_workloads(lib))
    run("q9_5_text_8MiB", 5, lambda: quality_9_5_workloads(torch, bm, enc))
    if "c4_silesia_1GiB_multi8_hinted" in frozen:
        run("c4_silesia_1GiB_multi8_hinted", 10, lambda: large_workload("c4_silesia_1GiB_multi8_hinted", torch, bm, lib, enc, frozen, work_fn))
    run("stream_nohint", 12, lambda: stream_nohint_workload(bm, lib))
    run("q10_11_alice29", 8, lambda: quality_10_11_workloads(torch, bm, enc, ("alice29",)))
    if "c4_silesia_128MiB_multi8_h5" in frozen:
        run("c4_silesia_128MiB_multi8_h5", 20, lambda: large_workload("c4_silesia_128MiB_multi8_h5", torch, bm, lib, enc, frozen, work_fn))
    run("q10_11_text_1MiB", 40, lambda: quality_10_11_workloads(torch, bm, enc, ("text_1MiB",)))
    return res


def _host_call_entry(lib, name, d, quality, lgwin, path, kernel, bound, reps=2):
    """BrotliEncoderCompress with host buffers in and out, best of `reps`, compared with the oracle run here on one pinned core"""
    import orc
    entry = {"workload": name, "input_bytes": len(d), "quality": quality, "lgwin": lgwin,
             "residency": "host buffers in and out (BrotliEncoderCompress)", "path": path}
    try:
        lib.compress(d[:65536], quality, lgwin)
        best = None
        for _ in range(reps):
            t0 = time.time()
            out = lib.compress(d, quality, lgwin)
            sec = time.time() - t0
            best = sec if best is None else min(best, sec)
        want, col = cpu_oracle_timed(lambda: orc.compress(d, quality, lgwin), len(d), "the same input, one run")
        entry.update({"value": round(len(d) / best / 1e6, 3), "unit": "MB/s", "ms_per_step": round(best * 1e3, 1), "compressed_bytes": len(out),
                      "identical_to_cpu_oracle": out == want, "cpu_oracle": col, "vs_cpu_oracle": round(len(d) / best / 1e6 / col["value"], 3),
                      "roofline": stream_roofline(kernel, len(d), len(out), best, bound)})
    except Exception as e:
        entry["error"] = repr(e)
    return entry


def quality_2_4_workloads(bm, lib):
    """SURVEY row f3, qualities 2..4 (BasicHasher family) through BrotliEncoderCompress with host buffers in and out: 2 MiB of text
    at each quality, 64 MiB at quality 4, and 16 MiB through BrotliEncoderCompressMulti as 16 shards at quality 2.
    Each compared with the oracle run here (liborc_fast.so, one pinned core), whose rate stands beside it."""
    import orc
    import synth
    res = []
    data = synth.markov_text(2 << 20)
    quick = "BasicHasher qualities on the speculative path: segments side by side on candidates derived from per-position flags (DESIGN.md section 3.10)"
    bound = "rounds of a fixed-point iteration over 512 B .. 2 KiB segments, one wavefront each (DESIGN.md section 3.10)"
    for quality in (2, 3, 4):
        res.append(_host_call_entry(lib, "q%d_text_2MiB" % quality, data, quality, 22, quick, "k_qs_parse", bound))
    q4_rate = res[-1].get("value", 0.0) or 0.0
    if q4_rate >= 30.0:  # (64 MiB at the 2 MiB rate must fit the budget of the side workloads: below 30 MB/s it does not)
        try:
            big = synth.markov_text(64 << 20, 5)
            res.append(_host_call_entry(lib, "q4_text_64MiB", big, 4, 22, quick, "k_qs_parse", bound, reps=2))
            del big
        except Exception as e:
            res.append({"workload": "q4_text_64MiB", "error": repr(e)})
    else:
        res.append({"workload": "q4_text_64MiB", "skipped": "one stream of quality 4 runs at %.1f MB/s on 2 MiB: 64 MiB would take %.0f s" % (q4_rate, 67.1 / max(q4_rate, 0.01))})
    try:
        big = synth.markov_text(16 << 20, 77)
        params = {bm.BROTLI_PARAM_QUALITY: 2, bm.BROTLI_PARAM_LGWIN: 22}
        lib.BrotliCompress(big[:1 << 20], params, 16)  # (warm-up: the helper threads of the library and their device memory pools)
        t0 = time.time()
        out = bytes(lib.BrotliCompress(big, params, 16))
        sec = time.time() - t0
        want, col = cpu_oracle_timed(lambda: orc.compress_multi(big, [(bm.BROTLI_PARAM_QUALITY, 2), (bm.BROTLI_PARAM_LGWIN, 22)], 16), len(big),
                                     "the same call, shards one after the other")
        res.append({"workload": "q2_text_16MiB_multi16", "input_bytes": len(big), "quality": 2, "lgwin": 22, "shards": 16,
                    "residency": "host buffers in and out (BrotliEncoderCompressMulti)", "value": round(len(big) / sec / 1e6, 3), "unit": "MB/s",
                    "ms_per_step": round(sec * 1e3, 1), "compressed_bytes": len(out), "identical_to_cpu_oracle": out == want, "cpu_oracle": col,
                    "vs_cpu_oracle": round(len(big) / sec / 1e6 / col["value"], 3),
                    "roofline": stream_roofline("k_qs_parse", len(big), len(out), sec, bound + "; 16 shards side by side")})
    except Exception as e:
        res.append({"workload": "q2_text_16MiB_multi16", "error": repr(e)})
    return res


def quality_0_1_workloads(lib):
    """SURVEY row f3, qualities 0 / 1: the fragments of a call (1 << lgwin bytes each, every one on a hash table of its own) run side
    by side: 2 MiB at lgwin 22 is ONE fragment, 64 MiB at lgwin 18 are 256, at lgwin 22 sixteen."""
    import synth
    res = []
    frag = "fragments of the call side by side (DESIGN.md section 3.10)"
    bound = "the parallelism of a call is its number of fragments"
    data = synth.markov_text(2 << 20)
    for quality in (0, 1):
        res.append(_host_call_entry(lib, "q%d_text_2MiB" % quality, data, quality, 22, frag + ": ONE fragment here", "k_fragment", bound))
    try:
        big = synth.markov_text(64 << 20, 5)
        for quality in (0, 1):
            res.append(_host_call_entry(lib, "q%d_text_64MiB_w18" % quality, big, quality, 18, frag + ": 256 fragments", "k_fragment", bound, reps=1))
        for quality in (0, 1):
            res.append(_host_call_entry(lib, "q%d_text_64MiB_w22" % quality, big, quality, 22, frag + ": 16 fragments", "k_fragment", bound, reps=1))
    except Exception as e:
        res.append({"workload": "q0_q1_text_64MiB", "error": repr(e)})
    return res


def stream_nohint_workload(bm, lib):
    """the CompressorWriter pattern (src/enc/writer.rs:269-313): BrotliEncoderCompressStream fed 4 KiB writes, NO size hint (size_hint
    = the first write => H5, whose StoreRangeOptBatch files masked ring entries past the 8 MiB ring buffer, mod.rs:1163-1232), quality 5,
    lgwin 22, text: 7 MiB (inside the first lap of the ring: the speculative path) and 16 MiB (past it: one live chain); the same input
    through the oracle's writer fed the same way (its rate is the CPU column, its bytes the expected stream)"""
    import orc
    import synth
    # (16 MiB: past the 8 MiB ring buffer, where the masked entries begin; 64 MiB at the rate of one live chain would not fit the budget)
    res = []
    for name, mib in (("stream_7MiB_q5_nohint", 7), ("stream_16MiB_q5_nohint", 16)):
        data = synth.markov_text(mib << 20, 5)
        entry = {"workload": name, "input_bytes": len(data), "quality": QUALITY, "lgwin": LGWIN,
                 "residency": "host buffers, BrotliEncoderCompressStream(PROCESS) in 4 KiB writes, then FINISH"}
        try:
            L = lib.lib
            step = 4096
            src = ctypes.create_string_buffer(data, len(data))
            cap = len(data) + (len(data) >> 2) + 4096
            dst = ctypes.create_string_buffer(cap)
            best = None
            for _ in range(2 if mib < 8 else 1):
                t0 = time.time()
                st = L.BrotliEncoderCreateInstance(None, None, None)
                L.BrotliEncoderSetParameter(st, int(bm.BROTLI_PARAM_QUALITY), QUALITY)
                L.BrotliEncoderSetParameter(st, int(bm.BROTLI_PARAM_LGWIN), LGWIN)
                base_in, base_out = ctypes.addressof(src), ctypes.addressof(dst)
                avail_out = ctypes.c_size_t(cap)
                next_out = ctypes.c_void_p(base_out)
                avail_in = ctypes.c_size_t(0)
                next_in = ctypes.c_void_p(base_in)
                total = ctypes.c_size_t(0)
                call = L.BrotliEncoderCompressStream
                refs = (ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out), ctypes.byref(next_out), ctypes.byref(total))
                for i in range(0, len(data), step):
                    avail_in.value = min(step, len(data) - i)
                    next_in.value = base_in + i
                    while True:
                        if not call(st, 0, *refs):
                            raise RuntimeError("BrotliEncoderCompressStream failed: " + lib.last_error())
                        if avail_in.value == 0:
                            break
                avail_in.value = 0
                while True:
                    if not call(st, 2, *refs):
                        raise RuntimeError("BrotliEncoderCompressStream(FINISH) failed: " + lib.last_error())
                    if L.BrotliEncoderIsFinished(st):
                        break
                n_out = cap - avail_out.value
                L.BrotliEncoderDestroyInstance(st)
                sec = time.time() - t0
                best = sec if best is None else min(best, sec)
            sec = best
            out = dst.raw[:n_out]
            want, col = cpu_oracle_timed(lambda: orc.writer_compress(data, QUALITY, LGWIN, chunk=step), len(data),
                                         "the same input through the oracle's CompressorWriter pattern, 4 KiB writes")
            entry.update({"value": round(len(data) / sec / 1e6, 2), "unit": "MB/s", "ms_per_step": round(sec * 1e3, 1), "compressed_bytes": len(out),
                          "identical_to_cpu_oracle": out == want, "cpu_oracle": col, "vs_cpu_oracle": round(len(data) / sec / 1e6 / col["value"], 3)})
        except Exception as e:
            entry["error"] = repr(e)
        res.append(entry)
    return res


def small_input_workloads(lib):
    """SURVEY section 8(d) input C1: testdata/alice29.txt (152 089 bytes) through BrotliEncoderCompress(5, 22), host buffers in and
    out -- what ONE small call costs on the device (a chain of about 150 launches and copies and four host round trips, whatever
    the size), and 64 calls from 16 host threads at once (every thread on its own HIP stream)."""
    import threading
    import orc
    import synth
    data = synth.alice()
    res = []
    try:
        want, col = cpu_oracle_timed(lambda: [orc.compress(data, QUALITY, LGWIN) for _ in range(5)][-1], 5 * len(data), "alice29.txt, five calls in a row")
        for _ in range(3):
            got = lib.compress(data, QUALITY, LGWIN)
        n = 30
        t0 = time.time()
        for _ in range(n):
            lib.compress(data, QUALITY, LGWIN)
        sec = (time.time() - t0) / n
        res.append({"workload": "c1_alice29_q5_one_call", "input_bytes": len(data), "quality": QUALITY, "lgwin": LGWIN,
                    "residency": "host buffers in and out (BrotliEncoderCompress), one call after the other", "value": round(len(data) / sec / 1e6, 2), "unit": "MB/s",
                    "ms_per_step": round(sec * 1e3, 3), "compressed_bytes": len(got), "identical_to_cpu_oracle": got == want, "cpu_oracle": col,
                    "vs_cpu_oracle": round(len(data) / sec / 1e6 / col["value"], 3),
                    "roofline": stream_roofline("k_parse_segments", len(data), len(got), sec,
                                                "launch chain and host round trips: the call is a fixed number of small launches, not bytes")})
        bad = []
        # sixteen threads of a server that is up: every thread has made two calls before the clock starts (a thread's first call sets
        # up its stream and its device memory pool: 64 calls from sixteen NEW threads measure sixteen set-ups, round 5: 50 MB/s)
        ready, go = threading.Barrier(17), threading.Barrier(17)
        done = threading.Barrier(17)

        def work():
            for _ in range(2):
                lib.compress(data, QUALITY, LGWIN)
            ready.wait()
            go.wait()
            for _ in range(4):
                if lib.compress(data, QUALITY, LGWIN) != want:
                    bad.append(1)
            done.wait()
        threads = [threading.Thread(target=work) for _ in range(16)]
        for t in threads:
            t.start()
        ready.wait()
        t0 = time.time()
        go.wait()
        done.wait()
        sec = time.time() - t0
        for t in threads:
            t.join()
        res.append({"workload": "c1_alice29_q5_64_calls_16_threads", "input_bytes": 64 * len(data), "quality": QUALITY, "lgwin": LGWIN,
                    "residency": "host buffers in and out, 16 host threads x 4 calls (after two untimed calls per thread), each thread on its own stream", "value": round(64 * len(data) / sec / 1e6, 2),
                    "unit": "MB/s", "ms_per_step": round(sec * 1e3, 2), "identical_to_cpu_oracle": not bad, "cpu_oracle": col,
                    "vs_cpu_oracle": round(64 * len(data) / sec / 1e6 / col["value"], 3),
                    "note": "a small call is bound by the host side of the HIP runtime (about 150 launches and copies), which the threads share: four calls are in flight at a time (BROTLI_MI355X_SMALL_CALLS_IN_FLIGHT)"})
    except Exception as e:
        res.append({"workload": "c1_alice29_q5", "error": repr(e)})
    return res


def quality_10_11_workloads(torch, bm, enc, names=("alice29", "text_1MiB")):
    """qualities 10 and 11 proper (H10 trees + the Zopfli shortest-path parse on the device, zopfli_device.h) on the
    input the reference holds its known answers for -- alice29, 47 488 / 46 493 bytes (src/bin/integration_tests.rs:401-449) -- and on
    1 MiB of the text generator; each compared with the oracle run here (one core), whose rate stands beside it"""
    import orc
    import synth
    res = []
    inputs = {"alice29": synth.alice, "text_1MiB": lambda: synth.markov_text(1 << 20)}
    for name, data in [(n, inputs[n]()) for n in names]:
        try:
            dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        except Exception as e:
            return res + [{"workload": "q10_11_" + name, "error": repr(e)}]
        for quality in (10, 11):
            params = [(bm.BROTLI_PARAM_QUALITY, quality), (bm.BROTLI_PARAM_LGWIN, 22), (bm.BROTLI_PARAM_SIZE_HINT, len(data))]
            entry = {"workload": "q%d_%s" % (quality, name), "input_bytes": len(data), "quality": quality, "lgwin": 22,
                     "residency": "input resident in HBM, output to host memory",
                     "path": "first device slice: matches of a block side by side (one lane per group of hash keys), the shortest-path "
                             "programme on one wavefront per stream (DESIGN.md section 3.9)"}
            try:
                sec, out = timed_steps(lambda: enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=True), 1, 0, torch)
                out = bytes(out)
                (want, _), col = cpu_oracle_timed(lambda: orc.stream_compress(data, params), len(data), "the same input, one run")
                entry.update({"value": round(len(data) / sec / 1e6, 3), "unit": "MB/s", "ms_per_step": round(sec * 1e3, 1), "compressed_bytes": len(out),
                              "identical_to_cpu_oracle": out == want, "cpu_oracle": col, "vs_cpu_oracle": round(len(data) / sec / 1e6 / col["value"], 3),
                              "roofline": stream_roofline("k_zopfli_parse", len(data), len(out), sec,
                                                          "the shortest-path programme of a block on one wavefront (about 6-9 us per position)")})
                if name == "alice29":
                    entry["reference_known_answer_bytes"] = {10: 47488, 11: 46493}[quality]
            except Exception as e:
                entry["error"] = repr(e)
            res.append(entry)
        del dev
    return res


def quality_9_5_workloads(torch, bm, enc):
    """quality 10 + BROTLI_PARAM_Q9_5 (the quality >= 10 meta-block builder behind the H9 search), 8 MiB of
    the text generator; compared with the oracle run here (one core), which is also the CPU figure beside it"""
    import orc
    import synth
    res = []
    try:
        data = synth.markov_text(8 << 20)
        dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    except Exception as e:
        return [{"workload": "q9_5_text_8MiB", "error": repr(e)}]
    for lgwin in (22, 18):
        params = [(bm.BROTLI_PARAM_QUALITY, 10), (150, 1), (bm.BROTLI_PARAM_LGWIN, lgwin), (bm.BROTLI_PARAM_SIZE_HINT, len(data))]
        entry = {"workload": "q9_5_text_8MiB_w%d" % lgwin, "input_bytes": len(data), "quality": "10 + Q9_5", "lgwin": lgwin,
                 "residency": "input resident in HBM, output to host memory"}
        try:
            sec, out = timed_steps(lambda: enc.encode(params, b"", dev.data_ptr(), len(data), True, copy=True), 1, 1, torch)
            out = bytes(out)
            (want, _), col = cpu_oracle_timed(lambda: orc.stream_compress(data, params), len(data), "the same 8 MiB, one run")
            entry.update({"value": round(len(data) / sec / 1e6, 2), "unit": "MB/s", "ms_per_step": round(sec * 1e3, 1), "compressed_bytes": len(out),
                          "identical_to_cpu_oracle": out == want, "cpu_oracle": col, "vs_cpu_oracle": round(len(data) / sec / 1e6 / col["value"], 3)})
        except Exception as e:
            entry["error"] = repr(e)
        res.append(entry)
    return res


def config4(torch, dist, bm, lib, enc, rank, world, frozen, steps):
    """BASELINE configs[3] cut to 1 GiB: Silesia-like mix, compress_multi with 8 shards dealt round-robin to the ranks"""
    import large_cases
    import synth
    from brotli_mi355x import multi
    name = "c4_silesia_1GiB_multi8_hinted"
    if name not in frozen:
        return {"workload": name, "skipped": "no frozen oracle hash (tools/freeze_large_hashes.py)"}
    total, nshards = frozen[name]["input_bytes"], 8
    plan = synth.silesia_plan(total, int(frozen[name]["seed"], 16))
    mine = {}
    t0 = time.time()
    for s in range(rank, nshards, world):
        lo, start, end = multi.shard_window(total, s, nshards, LGWIN)
        piece = synth.silesia_range(plan, lo, end)
        mine[s] = (piece[:start - lo], torch.frombuffer(bytearray(piece[start - lo:]), dtype=torch.uint8).cuda(), end - start)
    gen_s = time.time() - t0
    params = [(bm.BROTLI_PARAM_QUALITY, QUALITY), (bm.BROTLI_PARAM_LGWIN, LGWIN), (bm.BROTLI_PARAM_SIZE_HINT, min(total, 1 << 30))]

    def shard_input(s):
        prefix, dev, nbytes = mine[s]
        return prefix, dev.data_ptr(), nbytes, True

    def one():
        return multi.compress_multi_over_ranks(dist, lib, enc, params, total, nshards, rank, world, "cuda", shard_input)

    one()  # warm-up
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    out = None
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.time() - t0
    if out is not None:
        out = bytes(out)  # (a view of the encoder's output buffer until here: the next encode overwrites it)
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank != 0:
        return None
    return {"workload": "1 GiB Silesia-like mix (SURVEY 8d C4 recipe), quality=5, lgwin=22, size hint = input size, BrotliEncoderCompressMulti "
                        "semantics with 8 shards of 128 MiB dealt round-robin to %d GPU(s), gathered over RCCL, stitched on rank 0" % world,
            "scaling": "strong", "n_gpus": world, "steps": steps, "value": round(total * steps / elapsed / 1e6, 1), "unit": "MB/s",
            "ms_per_step": round(elapsed / steps * 1e3, 1), "compressed_bytes": len(out),
            "identical_to_cpu_oracle": hashlib.sha256(out).hexdigest() == frozen[name]["stream_sha256"],
            "input_generated_in_s": round(gen_s, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=WORKLOAD_BYTES >> 20, help="per-GPU batch size in MiB")
    ap.add_argument("--segment-bytes", type=int, default=0, help="bytes per parse chain (0 = the library's choice for the input size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip other_workloads / config4 / e2e")
    ap.add_argument("--config4", action="store_true", help="run the 4 GiB config 4 at N = 1 as well (it always runs for N > 1)")
    ap.add_argument("--extras-budget", type=float, default=60.0, help="wall-clock seconds the side workloads may take (they run in a child process)")
    ap.add_argument("--extras-only", action="store_true", help="run the side workloads only and write bench_extras.json")
    args = ap.parse_args()
    if args.extras_only:
        return extras_only(args)

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
    from brotli_mi355x import multi
    lib = bm.default_library()
    frozen = frozen_hashes()
    per_gpu = args.mib << 20
    total = per_gpu * world
    # one stream of world x per_gpu bytes; every rank needs its shard and the window in front of it
    stream = text_workload(total)
    if not shard_job:
        prefix, chunk = b"", stream
    else:
        lo, start, end = multi.shard_window(total, rank, world, LGWIN)
        prefix, chunk = stream[lo:start], stream[start:end]
    if rank != 0:
        del stream
    host_pinned = torch.frombuffer(bytearray(chunk), dtype=torch.uint8).pin_memory()
    dev = host_pinned.cuda()
    torch.cuda.synchronize()

    enc = multi.ShardEncoder(lib.lib, args.segment_bytes)
    # the compressed stream lands in page-locked host memory (a caller that hands over ordinary memory is measured
    # separately: e2e.c_abi_pageable)
    out_cap = len(chunk) + len(chunk) // 4 + 4096
    out_pinned = torch.empty(out_cap, dtype=torch.uint8).pin_memory()
    enc.use_output_buffer(out_pinned.data_ptr(), out_cap, out_pinned)
    work_fn = lib.lib.brotli_mi355x_last_parse_work
    work_fn.argtypes = [ctypes.POINTER(ctypes.c_double)]
    work_fn.restype = None
    # (BrotliEncoderCompress sets SIZE_HINT = input size; the shard job states the size of the whole stream the same way, like
    # c/brotli.c: without it the shards' hashers are H5, whose masked ring entries serialise the parse -- config4_h5 in
    # other_workloads measures that)
    params = [(bm.BROTLI_PARAM_QUALITY, QUALITY), (bm.BROTLI_PARAM_LGWIN, LGWIN), (bm.BROTLI_PARAM_SIZE_HINT, min(total, 1 << 30))]
    st = enc.stats
    job = multi.DeviceShardJob(dist, lib, enc, rank, world, per_gpu) if shard_job else None

    # N = 1: device-resident step as SURVEY 8d defines it -- input already in HBM, compressed stream left in HBM (the PCIe-
    # inclusive rates are the e2e block: stream into page-locked host memory, host -> HBM copy of the input inside the timed
    # region, plain C ABI with pageable buffers)
    out_dev = torch.empty(out_cap, dtype=torch.uint8, device="cuda")
    last_size = [0]

    def one_step():
        if not shard_job:
            last_size[0] = enc.encode_to_device(params, b"", dev.data_ptr(), len(chunk), out_dev.data_ptr(), out_cap)
            comp = None
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
    agg = {"parse_ms": 0.0, "launches": 0, "walked": 0.0, "searches_all": 0.0, "commands_all": 0.0, "rounds": 0.0}
    comp = None
    work = (ctypes.c_double * 4)()
    for _ in range(args.steps):
        comp, s = one_step()
        work_fn(work)
        agg["parse_ms"] += s[26]
        agg["launches"] += int(s[27])
        agg["walked"] += work[0]
        agg["searches_all"] += work[1]
        agg["commands_all"] += work[2]
        agg["rounds"] += s[0]
        S, K, Lit = s[1], s[2], s[3]
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

    c4 = None
    if not args.no_extras and (world > 1 or args.config4):
        try:
            c4 = config4(torch, dist, bm, lib, enc, rank, world, frozen, 1)
        except Exception as e:  # the headline must not die with an extra (every rank takes the same path: no collective is left hanging)
            c4 = {"error": repr(e)} if rank == 0 else None
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    comp = bytes(comp) if shard_job else bytes(out_dev[:last_size[0]].cpu().numpy())
    ms_per_step = elapsed / args.steps * 1e3
    value = total * args.steps / elapsed / 1e6
    # ---- roofline of the dominant kernel, from what its launches did (all chains: dry runs, round 0, re-parses)
    launches = max(1, agg["launches"])
    parse_s = agg["parse_ms"] * 1e-3
    bytes_done = 2.0 * agg["walked"] + 64.0 * agg["searches_all"] + 16.0 * agg["commands_all"]
    achieved = bytes_done / parse_s / 1e9 if parse_s > 0 else 0.0
    one_pass = (2.0 * per_gpu + 64.0 * S + 16.0 * K) * args.steps  # the useful share: one sequential pass over the input
    whole_step = 9.0 * per_gpu + 64.0 * S + 48.0 * K + 2.0 * Lit + len(comp) / world  # SURVEY 8d, bytes per step and GPU
    # PMC counters cannot be collected inside this run: the figure is quoted from the newest profiles/r*_pmc_parse.json -- but only
    # if that summary was measured on THIS source tree (fingerprint of rust-brotli_amd/csrc/*); otherwise traffic is null
    traffic, traffic_src = None, None
    import glob as _glob
    for pmc in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_parse.json")), reverse=True):
        try:
            j = json.load(open(pmc))
        except Exception:
            continue
        rel = os.path.relpath(pmc, ROOT)
        usable, words = measured_on_these_sources(j)
        if usable:
            traffic = j.get("hbm_bytes_per_launch")
            traffic_src = "%s (kernel fingerprint %s = this run)" % (rel, kernel_fingerprint())
        else:
            traffic_src = "null: %s was measured on other kernel sources (tools/gpu_round.sh profile re-takes it)" % rel
        break
    # THE line the driver reads: compact (< 4 KB, tests/test_bench_line.py); everything wordy goes to bench_extras.json and stderr
    line = {
        "metric": "compress MB/s at q5 lgwin22", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d MiB synthetic English-like text per GPU, quality=5, lgwin=22, %s" % (
                       args.mib, "one-shot BrotliEncoderCompress (H6), input and stream resident in HBM" if not shard_job else
                       "one stream of %d MiB, compress_multi shard per GPU + RCCL gather + BroCatli stitch" % (total >> 20)),
                   "input_bytes_total": total, "compressed_bytes": len(comp), "ratio": round(total / max(1, len(comp)), 4),
                   "lz77_rounds_per_step": agg["rounds"] / args.steps,
                   "stage_ms_last_step": {"lz77": round(lz_ms, 2), "metablock": round(mb_ms, 2), "library_total": round(lib_ms, 2)}},
        "roofline": {"bound": "hbm", "kernel": "k_parse_segments", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": round(agg["parse_ms"] / launches, 3), "launches_per_step": agg["launches"] / args.steps,
                     "alg_bytes_per_launch": int(bytes_done / launches),
                     "useful_only_frac": round(one_pass / parse_s / 1e9 / HBM_PEAK_GBS, 5) if parse_s > 0 else 0.0,
                     "whole_step_bytes": int(whole_step),
                     "whole_step_frac": round(whole_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "searches_per_step": agg["searches_all"] / args.steps, "searches_final_parse": S, "commands": K},
    }
    full = {"headline": dict(line), "phases_ms_last_step": [round(x, 2) for x in phases], "segment_bytes": int(seg_bytes),
            "accounting": "roofline bytes = 2 x positions walked + 64 x searches + 16 x commands, summed over every chain of every launch (counted "
                          "by the kernel); time = HIP events around the launches; useful_only = one pass over the input (2N + 64S + 16K) / all "
                          "kernel time; whole_step = SURVEY 8d formula 9N + 64S + 48K + 2L + C per step / ms_per_step",
            "positions_walked_per_step": agg["walked"] / args.steps}
    if shard_job and world > 1:
        key = "text_%dx64MiB_multi%d_hinted" % (world, world)
        if key in frozen and args.mib == 64:
            line["config"]["identical_to_cpu_oracle"] = hashlib.sha256(comp).hexdigest() == frozen[key]["stream_sha256"]
    if not args.no_extras and not shard_job:
        # the step with the stream delivered into page-locked host memory (the headline of rounds 1 and 2)
        sec, _ = timed_steps(lambda: enc.encode(params, b"", dev.data_ptr(), len(chunk), True, copy=False), min(args.steps, 5), 1, torch)
        line["output_to_pinned_host_ms"] = round(sec * 1e3, 3)

        # the same with the input coming from (pinned) host memory: the H2D copy is inside the timed region too
        def e2e_step():
            dev.copy_(host_pinned, non_blocking=True)
            return enc.encode(params, b"", dev.data_ptr(), len(chunk), True)
        sec, _ = timed_steps(e2e_step, min(args.steps, 5), 1, torch)
        line["e2e_pinned_ms"] = round(sec * 1e3, 3)
        # what a drop-in caller of the C ABI sees: BrotliEncoderCompress with input and output in ordinary (pageable) memory
        try:
            cabi = lib.lib.BrotliEncoderCompress
            cabi.restype = ctypes.c_int
            cabi.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
            src = ctypes.create_string_buffer(chunk, len(chunk))
            cap = len(chunk) + len(chunk) // 4 + 4096
            dst = ctypes.create_string_buffer(cap)

            def abi_step():
                n_out = ctypes.c_size_t(cap)
                if cabi(QUALITY, LGWIN, 0, len(chunk), src, ctypes.byref(n_out), dst) != 1:
                    raise RuntimeError("BrotliEncoderCompress failed")
                return n_out.value
            sec, n_out = timed_steps(abi_step, min(args.steps, 5), 2, torch)
            line["e2e_c_abi_pageable_ms"] = round(sec * 1e3, 3)
            line["e2e_c_abi_same_bytes"] = bytes(dst.raw[:n_out]) == comp
            del src, dst
        except Exception as e:
            full["e2e_c_abi_error"] = repr(e)
    if not args.no_cpu_baseline and not shard_job:  # (the CPU leg is reported at N = 1 only)
        base, ref_bytes = cpu_baseline(stream)
        line["cpu_baseline"] = base
        line["config"]["identical_to_cpu_oracle"] = (ref_bytes == comp)
    if c4 is not None:
        full["config4"] = c4
        line["config4"] = {k: c4.get(k) for k in ("value", "unit", "ms_per_step", "identical_to_cpu_oracle", "scaling", "error") if k in c4}
    if not args.no_extras and not shard_job:
        # the side workloads run in a CHILD process under a wall-clock budget: whatever happens there (a crash, a hang cut by the
        # timeout), the line below is printed; their full records go to bench_extras.json and stderr, a digest
        # {name: [MB/s, x one CPU core (oracle on this host), identical to the oracle]} into the line
        import subprocess
        del enc, job, dev, out_dev
        torch.cuda.empty_cache()
        try:
            lib.lib.BrotliMi355xTrimPool()
        except Exception:
            pass
        extras_path = os.path.join(ROOT, "bench_extras.json")
        try:
            if os.path.exists(extras_path):
                os.remove(extras_path)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--extras-only", "--extras-budget", str(args.extras_budget)],
                           stdout=sys.stderr, stderr=sys.stderr, timeout=args.extras_budget + 90)
        except Exception as e:
            full["extras_error"] = repr(e)
        try:
            ex = json.load(open(extras_path))
            full["other_workloads"] = ex.get("other_workloads", [])
            full["extras_skipped_for_budget"] = ex.get("skipped", [])
            line["other"] = extras_digest(full["other_workloads"])
            if ex.get("skipped"):
                line["other_skipped_for_budget"] = len(ex["skipped"])
        except Exception as e:
            full["extras_error"] = full.get("extras_error", "") + " / " + repr(e)
    try:
        json.dump(full, open(os.path.join(ROOT, "bench_extras.json"), "w"), indent=1)
    except Exception:
        pass
    text = final_text(line)
    sys.stderr.flush()
    print(text)
    sys.stdout.flush()
    if dist:
        dist.destroy_process_group()


LINE_LIMIT = 4000  # bytes; the driver keeps an 8 KB tail of stdout and must find the whole line in it


def final_text(line):
    """the one stdout line: compact separators, and below LINE_LIMIT whatever the extras did -- optional parts are dropped in a fixed
    order (digest entries from the end, then the digest, then the e2e figures) until it fits; the required keys (metric, value,
    unit, n_gpus, steps, warmup, ms_per_step, dtype, config, roofline, cpu_baseline) are never touched"""
    line = dict(line)
    text = json.dumps(line, separators=(",", ":"))
    while len(text) >= LINE_LIMIT:
        other = line.get("other")
        if isinstance(other, dict) and other:
            other = dict(other)
            other.popitem()
            line["other"] = other
            line["other_truncated"] = True
        elif "other" in line:
            del line["other"]
        else:
            optional = [k for k in ("config4", "e2e_c_abi_pageable_ms", "e2e_c_abi_same_bytes", "e2e_pinned_ms", "output_to_pinned_host_ms") if k in line]
            if not optional:
                break
            del line[optional[0]]
        text = json.dumps(line, separators=(",", ":"))
    return text


def extras_digest(entries):
    """{workload: [MB/s, x one CPU core of this host (sampled ratios are estimates), identical to the CPU oracle]}"""
    d = {}
    for e in entries:
        name = str(e.get("workload", "?"))[:40]
        if "error" in e:
            d[name] = "error"
            continue
        ratio = e.get("vs_cpu_oracle", e.get("vs_cpu_oracle_sampled"))
        d[name] = [e.get("value"), ratio, e.get("identical_to_cpu_oracle")]
    return d


def extras_only(args):
    """child process of the default run (also: python bench.py --extras-only --extras-budget 600 for everything): the side workloads,
    full records to bench_extras.json + stderr"""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    import brotli_mi355x as bm
    from brotli_mi355x import multi
    lib = bm.default_library()
    enc = multi.ShardEncoder(lib.lib, 0)
    work_fn = lib.lib.brotli_mi355x_last_parse_work
    work_fn.argtypes = [ctypes.POINTER(ctypes.c_double)]
    work_fn.restype = None
    budget = Budget(args.extras_budget)
    res = []
    try:
        res = other_workloads(torch, bm, lib, enc, frozen_hashes(), work_fn, budget)
    finally:
        json.dump({"other_workloads": res, "skipped": budget.skipped}, open(os.path.join(ROOT, "bench_extras.json"), "w"))


if __name__ == "__main__":
    main()
