"""Concurrent callers: every host thread of a caller works on its own HIP stream (hipStreamPerThread) and its own device / pinned
memory pools; the calls of 48 short-lived threads must give the oracle's bytes, and the process must come down cleanly afterwards --
thread-local pools used to call into the HIP runtime from the destructors of dying threads and corrupted the heap now and then
("malloc_consolidate(): unaligned fastbin chunk detected" at exit; round 5, device_runtime.hip Orphans)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = """
import faulthandler, sys, threading
faulthandler.enable()
sys.path.insert(0, %r)
import synth, orc, test_cabi
lib = test_cabi._load("gpu")
data = synth.alice()
want = {q: orc.compress(data, q, 22) for q in (1, 5)}
assert lib.compress(data, 5, 22) == want[5]
bad = []
def work(q):
    for _ in range(4):
        if lib.compress(data, q, 22) != want[q]:
            bad.append(q)
for rep in range(3):
    ts = [threading.Thread(target=work, args=(5 if i %% 4 else 1,)) for i in range(48)]
    for t in ts: t.start()
    for t in ts: t.join()
assert not bad, bad
print("ok")
"""


@pytest.mark.gpu
def test_many_short_lived_threads_gpu():
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", CHILD % HERE], capture_output=True, text=True, timeout=900)
        noise = [w for w in ("malloc", "corrupt", "double free", "Aborted", "Segmentation") if w in r.stderr]
        assert r.returncode == 0 and "ok" in r.stdout and not noise, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
