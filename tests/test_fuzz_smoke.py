"""A short run of the randomised sweeps (tests/fuzz_gpu.py, tests/fuzz_api.py) inside the suites: on the host emulation
build for `-m "not gpu"`, on the device for `-m gpu`.  The long sweeps are run by hand (README)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, *args, **env):
    r = subprocess.run([sys.executable, os.path.join(HERE, script)] + [str(a) for a in args], cwd=HERE, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500, env=dict(os.environ, **env))
    tail = r.stdout.decode(errors="replace")[-2000:]
    assert r.returncode == 0, tail
    return tail


def test_one_shot_sweep_emulation():
    import emu
    emu.build()
    assert "0 mismatches" in _run("fuzz_gpu.py", 60, 5, "emu", 0.3)


def test_api_sweep_emulation():
    import emu
    emu.build()
    assert " 0 failures" in _run("fuzz_api.py", 40, 5, "emu")


def test_api_sweep_quality_9_5_emulation():
    """the same stream operations at quality 10 + BROTLI_PARAM_Q9_5 (the quality >= 10 meta-block builder, row b10)"""
    import emu
    emu.build()
    assert " 0 failures" in _run("fuzz_api.py", 30, 9, "emu", FUZZ_Q9_5="1")


def test_api_sweep_quality_10_11_emulation():
    """the stream operations that do not cut a stream into pieces, the multi-shard entry and custom dictionaries at qualities 10 and 11
    proper (row f1: H10 + Zopfli, zopfli_device.h)"""
    import emu
    emu.build()
    assert " 0 failures" in _run("fuzz_api.py", 25, 12, "emu", FUZZ_ZOPFLI="10")
    assert " 0 failures" in _run("fuzz_api.py", 25, 13, "emu", FUZZ_ZOPFLI="11")


def test_api_sweep_quality_2_4_emulation():
    """every stream operation, the multi-shard entry and custom dictionaries at qualities 2 .. 4 (row f3: the BasicHasher family,
    quick_device.h), windows from lgwin 10"""
    import emu
    emu.build()
    assert " 0 failures" in _run("fuzz_api.py", 100, 21, "emu", FUZZ_QUICK="1")
    assert " 0 failures" in _run("fuzz_api.py", 100, 22, "emu", FUZZ_QUICK="1", FUZZ_TINY="1")


def test_api_sweep_quality_0_1_emulation():
    """the stream operations, extra parameters and the writer pattern at qualities 0 and 1 (row f3: the fragment compressors,
    fragment_device.h), windows from lgwin 10"""
    import emu
    emu.build()
    assert " 0 failures" in _run("fuzz_api.py", 200, 31, "emu", FUZZ_FRAGMENT="1")
    assert " 0 failures" in _run("fuzz_api.py", 150, 32, "emu", FUZZ_FRAGMENT="1", FUZZ_TINY="1")


@pytest.mark.gpu
def test_api_sweep_quality_0_1_device():
    assert " 0 failures" in _run("fuzz_api.py", 16, 31, FUZZ_FRAGMENT="1", FUZZ_MAXN="150000")
    assert " 0 failures" in _run("fuzz_api.py", 20, 32, FUZZ_FRAGMENT="1", FUZZ_TINY="1")


@pytest.mark.gpu
def test_api_sweep_quality_2_4_device():
    assert " 0 failures" in _run("fuzz_api.py", 16, 21, FUZZ_QUICK="1", FUZZ_MAXN="150000")
    assert " 0 failures" in _run("fuzz_api.py", 20, 22, FUZZ_QUICK="1", FUZZ_TINY="1")


@pytest.mark.gpu
def test_api_sweep_quality_10_11_device():
    assert " 0 failures" in _run("fuzz_api.py", 8, 12, FUZZ_ZOPFLI="10", FUZZ_TINY="1")
    assert " 0 failures" in _run("fuzz_api.py", 4, 14, FUZZ_ZOPFLI="11")


@pytest.mark.gpu
def test_api_sweep_quality_9_5_device():
    assert " 0 failures" in _run("fuzz_api.py", 40, 9, FUZZ_Q9_5="1")
    assert " 0 failures" in _run("fuzz_api.py", 25, 10, FUZZ_Q9_5="11")  # quality 11: the 512-deep H5 / H6 rings


@pytest.mark.gpu
def test_one_shot_sweep_device():
    assert "0 mismatches" in _run("fuzz_gpu.py", 200, 6)


@pytest.mark.gpu
def test_api_sweep_device():
    assert " 0 failures" in _run("fuzz_api.py", 300, 6)
