"""The last stdout line of bench.py is what the driver records (BENCH_rNN.json): it must stay one compact JSON object below 4 KB
whatever the side workloads did (round 5's 24.6 KB line was not recovered by the driver and the round went unmeasured)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline")


def _fake_line(n_other):
    import bench
    entries = [{"workload": "workload_number_%02d_with_a_long_name_xxxxxxxxxxxxxxxxxxxxxxxxxxx" % i, "value": 123456.789, "vs_cpu_oracle": 1234.567,
                "identical_to_cpu_oracle": True} for i in range(n_other)]
    entries.append({"workload": "broken", "error": "RuntimeError('x')"})
    return {
        "metric": "compress MB/s at q5 lgwin22", "value": 3957.61, "unit": "MB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 16.957,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "64 MiB synthetic English-like text per GPU, quality=5, lgwin=22, one-shot BrotliEncoderCompress (H6), input and stream resident in HBM",
                   "input_bytes_total": 67108864, "compressed_bytes": 16777216, "ratio": 4.0001, "lz77_rounds_per_step": 2.0,
                   "stage_ms_last_step": {"lz77": 11.11, "metablock": 3.33, "library_total": 16.66}, "identical_to_cpu_oracle": True},
        "roofline": {"bound": "hbm", "kernel": "k_parse_segments", "achieved": 195.61, "peak": 8000.0, "unit": "GB/s", "frac": 0.02445, "traffic": 863800000,
                     "traffic_source": "profiles/r06_pmc_parse.json (kernel fingerprint 0123456789ab = this run)", "avg_launch_ms": 1.271, "launches_per_step": 5.0,
                     "alg_bytes_per_launch": 248700000, "useful_only_frac": 0.0194, "whole_step_bytes": 1648000000, "whole_step_frac": 0.012,
                     "searches_per_step": 15360000.0, "searches_final_parse": 12000000.0, "commands": 5370000.0},
        "output_to_pinned_host_ms": 17.5, "e2e_pinned_ms": 19.5, "e2e_c_abi_pageable_ms": 25.5, "e2e_c_abi_same_bytes": True,
        "cpu_baseline": {"value": 59.8, "unit": "MB/s", "cores": 1, "kind": "port", "sample": "the whole 64 MiB workload, best of 6 runs, oracle built -O3 -march=native, pinned to core 255",
                         "libbrotlienc_1_0_9": {"value": 66.0, "unit": "MB/s", "cores": 1, "sample": "first 16 MiB of the workload, best of 2, BrotliEncoderCompress(5, 22)"}},
        "other": bench.extras_digest(entries),
    }


@pytest.mark.parametrize("n_other", [0, 25, 400])
def test_final_line_is_compact_and_keeps_the_required_keys(n_other):
    import bench
    text = bench.final_text(_fake_line(n_other))
    assert "\n" not in text and len(text) < 4096
    j = json.loads(text)
    for k in REQUIRED:
        assert k in j, k
    assert j["roofline"]["frac"] == 0.02445 and j["cpu_baseline"]["kind"] == "port"
    if n_other == 25:
        assert len(j["other"]) == 26 and j["other"]["broken"] == "error"  # (the digest of this file's workloads fits whole)
    if n_other == 400:
        assert j.get("other_truncated") is True


def test_budget_skips_what_does_not_fit():
    import bench
    b = bench.Budget(5.0)
    assert b.allows("short", 1.0)
    assert not b.allows("long", 30.0)
    assert b.skipped == ["long"]


@pytest.mark.gpu
def test_bench_prints_one_compact_parseable_line():
    """the driver's own command shape, short: the LAST stdout line parses, is below 4 KB and carries roofline + cpu_baseline + the digest;
    the full records are in bench_extras.json"""
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--extras-budget", "12"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096
    j = json.loads(lines[0])
    for k in REQUIRED:
        assert k in j, k
    assert j["config"]["identical_to_cpu_oracle"] is True
    assert j["roofline"]["frac"] > 0 and j["cpu_baseline"]["value"] > 0
    assert isinstance(j.get("other"), dict) and j["other"]
    full = json.load(open(os.path.join(ROOT, "bench_extras.json")))
    assert full["headline"]["metric"] == j["metric"] and full["other_workloads"]
