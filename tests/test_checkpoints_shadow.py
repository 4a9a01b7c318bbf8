"""Checkpoints (lz77_chain.h): a chain of a list launch may restart from a record of the segment's last parse and stop where
it finds itself in a recorded state, splicing old and new halves.  The emulation library built with -DBR_DEBUG_SPLICE=1
parses every such segment again from its entry to its end and compares exit and commands ("SHADOW MISMATCH"), and checks
every restart record against a parse from the entry to the record ("RESTART RECORD STALE").  48 MiB of text with the splice
kernel on every list launch: past the point where the static dictionary is switched off (39 MB), where the host refreshes
entry guesses with a dry run -- the regime in which both bugs of round 3 lived."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r"""
import sys, ctypes
sys.path.insert(0, %(tests)r)
import synth, emu
from cmp_lz77 import check
L = emu.bind_trace(ctypes.CDLL(%(lib)r))
assert check("text48M", synth.markov_text(48 << 20), 5, 22, lib=L, seg=2048)
assert check("mixed4M", synth.mixed(4 << 20, seed=31), 5, 22, lib=L, seg=512)
print("ok")
"""


def test_spliced_and_restarted_segments_equal_a_parse_from_their_entry():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu"), "libbrotli_emu_shadow.so"])
    env = dict(os.environ)
    env["BROTLI_MI355X_SPLICE_SHARE"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD % {"tests": HERE, "lib": os.path.join(HERE, "emu", "libbrotli_emu_shadow.so")}], env=env, capture_output=True,
                         text=True, timeout=2400)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "MISMATCH" not in out.stderr and "STALE" not in out.stderr, out.stderr[-3000:]
