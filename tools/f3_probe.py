"""tools/f3_probe.py -- qualities 0..4 on the GPU: time and identity for 1 MiB of text and of random bytes (run through gpurun).
One JSON object per line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import synth  # noqa: E402
import test_cabi  # noqa: E402

lib = test_cabi._load("gpu")
cases = [("text_1MiB", synth.markov_text(1 << 20)), ("random_1MiB", synth.random_bytes(1 << 20))]
lib.compress(cases[0][1][:4096], 5, 22)  # (device start-up outside the timed calls)
for name, d in cases:
    for q in (0, 1, 2, 3, 4):
        t = time.time()
        out = lib.compress(d, q, 22)
        dt = time.time() - t
        t = time.time()
        want = orc.compress(d, q, 22)
        cpu = time.time() - t
        print(json.dumps({"workload": "q%d_%s" % (q, name), "input_bytes": len(d), "compressed_bytes": len(out), "device": lib.device_name(),
                          "seconds": round(dt, 4), "value": round(len(d) / dt / 1e6, 3), "unit": "MB/s", "host_buffers": True,
                          "identical_to_cpu_oracle": out == want,
                          "cpu_oracle": {"value": round(len(d) / cpu / 1e6, 2), "unit": "MB/s", "cores": 1, "sample": "the same input, one run"}}), flush=True)
