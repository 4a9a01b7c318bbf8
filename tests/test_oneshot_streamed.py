"""BrotliEncoderCompress above BROTLI_MI355X_ONESHOT_STREAM_ABOVE goes through the stream state machine in batches (bounded
device memory, no 2 GiB limit; cabi.cpp CompressOneShotStreamed).  The stream must be the one-shot stream of the oracle
(encoder_compress, encode.rs:1436-1538).  The thresholds are read once per process, so the check runs in a child with both
scaled down to a few MiB; test_large_gpu.py holds the full-size case (3 GiB)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import os, sys
sys.path.insert(0, %(tests)r)
sys.path.insert(0, os.path.join(%(root)r, "rust-brotli_amd"))
import synth, orc
kind = sys.argv[1]
if kind == "emu":
    import emu, importlib.util
    emu.build()
    spec = importlib.util.spec_from_file_location("brotli_mi355x_emu", os.path.join(%(root)r, "rust-brotli_amd", "brotli_mi355x", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    os.environ["BROTLI_MI355X_LIB"] = os.path.join(emu.EMU_DIR, "libbrotli_emu.so")
    spec.loader.exec_module(mod)
    lib = mod.Library(os.path.join(emu.EMU_DIR, "libbrotli_emu.so"))
else:
    import brotli_mi355x
    lib = brotli_mi355x.default_library()
cases = [(synth.markov_text(5 << 20, 3), 5, 22), (synth.mixed(3 << 20, seed=4), 5, 18), (synth.markov_text((2 << 20) + 65536, 5), 7, 20),
         (synth.random_bytes(1 << 21, 6) + synth.markov_text(1 << 20, 7), 5, 22), (synth.markov_text(1 << 21, 8), 9, 22),
         # qualities 2 .. 4 take the stream machine from 64 MiB on (here: from the scaled-down threshold): the table travels
         (synth.markov_text(3 << 20, 9), 2, 22), (synth.mixed(3 << 20, seed=10), 3, 18), (synth.markov_text((2 << 20) + 12345, 11), 4, 20)]
for data, q, w in cases:
    got = lib.compress(data, q, w)
    ref = orc.compress(data, q, w)
    assert got == ref, (len(data), q, w, len(got), len(ref))
print("ok")
"""


def _run(kind):
    env = dict(os.environ)
    env["BROTLI_MI355X_ONESHOT_STREAM_ABOVE"] = "300000"
    env["BROTLI_MI355X_STREAM_BATCH"] = str(1 << 20)
    out = subprocess.run([sys.executable, "-c", CHILD % {"tests": HERE, "root": ROOT}, kind], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_oneshot_through_the_stream_machine_emulation():
    _run("emu")


@pytest.mark.gpu
def test_oneshot_through_the_stream_machine_gpu():
    _run("gpu")
