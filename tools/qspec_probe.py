"""tools/qspec_probe.py [MiB ...] -- qualities 2..4 on the speculative path (quick_spec.h) on the GPU: time and identity with the oracle
for text of the given sizes (default 2 and 64 MiB) through BrotliEncoderCompress(q, 22), host buffers in and out.  One JSON object per line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc, synth, test_cabi

lib = test_cabi._load("gpu")
sizes = [int(a) for a in sys.argv[1:]] or [2, 64]
lib.compress(synth.markov_text(1 << 16), 5, 22)
for mib in sizes:
    d = synth.markov_text(mib << 20)
    for q in (2, 3, 4):
        times = []
        for _ in range(3):
            t = time.time()
            out = lib.compress(d, q, 22)
            times.append(time.time() - t)
        sample = d[: min(len(d), 8 << 20)]
        t = time.time()
        want_s = orc.compress(sample, q, 22)
        cpu = time.time() - t
        same = (out == want_s) if len(sample) == len(d) else (orc.compress(d, q, 22) == out)
        dt = min(times)
        print(json.dumps({"workload": "q%d_text_%dMiB" % (q, mib), "compressed_bytes": len(out), "ms": [round(x * 1e3, 2) for x in times],
                          "value": round(len(d) / dt / 1e6, 1), "unit": "MB/s", "identical_to_cpu_oracle": same,
                          "cpu_oracle_MBps": round(len(sample) / cpu / 1e6, 1)}), flush=True)
