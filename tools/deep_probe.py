"""Quality 11 + Q9_5 (512-deep rings) probe on the GPU: the reference's 129 715-byte known answer through the C ABI with the
deep-ring kernels (profiles/r03_q11_q9_5_deep_rings_probe.log: the last ten GPU-seconds of round 3).  python tools/deep_probe.py"""
import os, sys
sys.path.insert(0, "tests")
import orc, test_cabi
lib = test_cabi._load("gpu")
d = open("tests/golden/random_then_unicode", "rb").read()
params = [(1, 11), (150, 1), (2, 22), (5, 2048 * 1024)]
e = lib.encoder(params=params)
for i in range(0, len(d), 4096):
    e.write(d[i:i + 4096])
got = e.finish()
e.close()
print("q11+Q9_5 on the device:", len(got), "identical to the oracle" if got == orc.reader_compress(d, params, chunk=4096) else "DIFFERENT", flush=True)
