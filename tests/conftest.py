import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

# PyTorch-ROCm brings its own copy of the HIP / HSA runtime.  A process that initialises the system runtime first (through the
# encoder library) and torch's second ends up with two runtimes, and torch then finds no device ("No HIP GPUs are available",
# profiles/r04_torch_check.log).  Loaded first, torch's copy is the one the dynamic loader hands to the library as well.  A full
# run of the suite imports torch while collecting (test_concat_ends.py, test_multi_gloo.py); this makes a run of selected files
# behave the same.  (bench.py and multi.py import torch before they load the library.)
try:
    import torch  # noqa: F401
except ImportError:
    pass


# Every resolver pass that takes blocks over from the pass before it (Lz77Stage::ResolvePass) is run again over every block and
# compared, in the emulation build and the device library alike (read once per process; a few per cent of a test's time).
os.environ.setdefault("BROTLI_MI355X_SELFTEST_RESOLVE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def alice():
    import synth
    return synth.alice()
