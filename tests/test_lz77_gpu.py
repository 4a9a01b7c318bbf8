"""GPU parity of the backward-reference stage: the command list produced by the HIP kernels must be
identical to the one the oracle (CPU restatement of the reference) produces."""
import pytest

import synth
from cmp_lz77 import check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import gpulib
    return gpulib.lib()


def test_alice_q5(L):
    assert check("alice", synth.alice(), 5, 22, lib=L)


@pytest.mark.parametrize("q", [6, 7, 8])
def test_alice_other_qualities(L, q):
    assert check("alice", synth.alice(), q, 22, lib=L)


def test_markov_h6(L):
    # > 4 MiB with size_hint => H6 (encode.rs:863-876), several meta-blocks
    assert check("markov6M", synth.markov_text(6 << 20), 5, 22, lib=L)


def test_random_and_zeros(L):
    assert check("random300k", synth.random_bytes(300000), 5, 22, lib=L)
    assert check("zeros300k", bytes(300000), 5, 22, lib=L)


def test_mixed(L):
    assert check("mixed2M", synth.mixed(2 << 20), 5, 22, lib=L)


def test_catable_shard_with_prefix(L):
    a = synth.alice()
    h = len(a) // 2
    assert check("alice shard1", a[h:], 5, 22, size_hint=0, catable=True, prefix=a[:h], lib=L)


@pytest.mark.parametrize("seg", [1024, 65536])
def test_segment_sizes(L, seg):
    assert check("alice", synth.alice(), 5, 22, seg=seg, lib=L)


def test_stored_flags_equal_reference_table(L):
    import cmp_flags
    for name, d in cmp_flags.tricky_inputs():
        assert cmp_flags.stored_flags_match(L, d), name
        assert check(name, d, 5, 22, lib=L)


def test_mixed_16M(L):
    # mixed content across several meta-blocks (zero runs, binary records, text, random)
    assert check("mixed16M", synth.mixed(16 << 20), 5, 22, lib=L)


def test_quality9_h9(L):
    a = synth.alice()
    assert check("alice q9", a, 9, 22, lib=L)
    assert check("alice q9 w16", a, 9, 16, lib=L)
    assert check("markov4M q9 w18", synth.markov_text(4 << 20), 9, 18, lib=L)
    assert check("mixed2M q9", synth.mixed(2 << 20), 9, 22, lib=L)
    import cmp_flags
    assert cmp_flags.stored_flags_match(L, bytes(2 << 20), 9, 18)


def test_silesia_like_and_enwik_like(L):
    # SURVEY 8d C3 / C4 content: XML-wrapped text; a mix of text, XML, binary records, zero fill, hex and random
    # stretches (literal sprees whose phase the resolver predicts, distance caches that pass through whole segments)
    assert check("enwik8M", synth.enwik_like(8 << 20), 5, 22, lib=L)
    assert check("silesia16M", synth.silesia_like(16 << 20, min_segment=64 << 10, max_segment=1 << 20), 5, 22, lib=L)
    assert check("binary4M", synth.silesia_like(4 << 20, only=60), 5, 22, lib=L)
    assert check("hex4M", synth.silesia_like(4 << 20, only=85), 5, 22, lib=L)
    assert check("silesia4M q9", synth.silesia_like(4 << 20, min_segment=64 << 10, max_segment=512 << 10), 9, 22, lib=L)


def test_distance_cache_check(L):
    import check_cache_cases
    check_cache_cases.run(L)


def test_lazy_probe_across_a_segment_boundary(L):
    import os
    d = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lazy_probe_across_segments.bin"), "rb").read()
    assert check("lazy probe", d, 5, 22, seg=256, lib=L)


def test_h5_past_the_first_ring_revolution(L):
    """1 MiB at lgwin 17: the ring buffer (256 KiB) goes round three times under an H5 hasher, whose StoreRangeOptBatch files
    masked positions (mod.rs:1163-1232) -- parsed by a live chain (lz77_live.h), checked against the oracle as it is"""
    assert check("markov1M-w17", synth.markov_text(1 << 20), 5, 17, lib=L)
    assert check("mixed1M-w17-q7", synth.mixed(1 << 20), 7, 17, lib=L)
    assert check("stretches2M-w18", synth.stretches(2 << 20), 5, 18, lib=L)


def test_round_counts_at_64MiB(L):
    """How many passes of the host resolver the speculative parse needs (segments cut as EncodeStream cuts them; the device
    schedules launches in between by itself while the parse settles fast, bursts).  Guards against a scheduling change that
    silently costs rounds; the streams themselves are checked by test_large_gpu.py.  Measured in round 3: text 2, Silesia-like
    6, random 10 (synth.mixed: 64 -- the pieces of such a mix are coupled through the hash table)."""
    import emu
    limits = [("text", synth.markov_text(64 << 20), 3), ("silesia", synth.silesia_like(64 << 20), 8), ("random", synth.random_bytes(64 << 20), 12)]
    for name, data, limit in limits:
        mbs, st = emu.lz77_trace(L, data, 5, 22, len(data), False, b"", 0)
        assert st["rounds"] <= limit, (name, st["rounds"])


def test_checkpoint_and_burst_switches_leave_the_parse_alone(L):
    """the command list with chains stopping at checkpoints and launches scheduled on the device (the default) equals the
    oracle's -- and so does the one of a child process with both switched off (they are read once per process)"""
    import os
    import subprocess
    import sys
    child = ("import sys; sys.path.insert(0, %r); import synth, gpulib; from cmp_lz77 import check; "
             "assert check('mixed6M', synth.mixed(6 << 20, seed=21), 5, 22, lib=gpulib.lib(), seg=1024); "
             "assert check('text12M', synth.markov_text(12 << 20, 22), 5, 22, lib=gpulib.lib(), seg=2048); print('ok')") % os.path.dirname(os.path.abspath(__file__))
    for extra in ({}, {"BROTLI_MI355X_NO_SPLICE": "1", "BROTLI_MI355X_BURST": "0"}, {"BROTLI_MI355X_NO_CHECKPOINTS": "1", "BROTLI_MI355X_NO_SPEC": "1"}):
        env = dict(os.environ)
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "ok" in out.stdout, (extra, out.stdout[-2000:], out.stderr[-2000:])


def test_quad_that_straddles_the_end_of_the_first_ring_revolution(L):
    from test_emu_parity import _quad_straddles_ring_end
    _quad_straddles_ring_end(L)


def test_group_kernel_opt_in():
    """k_parse_groups (lz77_groups.h: four chains per wavefront, round 0 and the warm-up), an opt-in experiment: with
    BROTLI_MI355X_GROUPS_MIN=0 every qualifying launch goes through it and the command lists must still equal the oracle's
    (the switch is read once per process, hence the child process)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); import torch, gpulib, synth; from cmp_lz77 import check; L = gpulib.lib(); "
            "ok = check('alice', synth.alice(), 5, 22, lib=L) and check('markov6M', synth.markov_text(6 << 20), 5, 22, lib=L) and "
            "check('mixed2M', synth.mixed(2 << 20), 5, 22, lib=L) and check('random300k', synth.random_bytes(300000), 5, 22, lib=L) and "
            "check('zeros300k', bytes(300000), 5, 22, lib=L); sys.exit(0 if ok else 1)" % here)
    env = dict(os.environ, BROTLI_MI355X_GROUPS_MIN="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-2000:]


def test_candidate_rows_equal_a_rebuild_after_every_update():
    """the flip-cell filters and the listed / incremental row maintenance exist in the HIP build only: with
    BROTLI_MI355X_SELFTEST_ROWS the rows are compared with a rebuild from the flags after EVERY update (Lz77Stage::SelfTestRows), on
    random, mixed, Silesia-like and text input, byte identity on top (tools/rows_selftest_gpu.py quick; the switch is read once per
    process, hence the child process)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rows_selftest_gpu.py"), "quick"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"ROWS SELFTEST OK" in p.stdout, p.stdout.decode()[-2000:]
