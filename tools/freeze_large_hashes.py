"""Freezes the CPU oracle's output for the BASELINE.json configurations at their stated sizes.

The oracle needs minutes for these inputs, so the GPU parity tests (tests/test_large_gpu.py) do not run it: they
regenerate the same seeded input, push it through the HIP path and compare sha256(stream) with the value frozen here.
Run in the build container (CPU only):   ORC_FAST=1 python tools/freeze_large_hashes.py [case ...]
Writes tests/golden/large_hashes.json (merging with what is there).  Every entry also records sha256(input), so that a
drift of a generator is told apart from a drift of an encoder."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import large_cases  # noqa: E402
import orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "large_hashes.json")


def main():
    want = sys.argv[1:] or list(large_cases.CASES)
    frozen = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in want:
        case = large_cases.CASES[name]
        seed = None
        for seed in case.get("seeds", [None]):
            t0 = time.time()
            data = case["make"]() if seed is None else case["make"](seed)
            t1 = time.time()
            try:
                params = [(1, case["quality"]), (2, case["lgwin"])] + ([(5, case["hint"])] if case.get("hint") else [])
                if case.get("writer_chunk"):
                    out = (orc.reader_compress(data, params, chunk=case["writer_chunk"]) if case.get("hint") else
                           orc.writer_compress(data, case["quality"], case["lgwin"], chunk=case["writer_chunk"]))
                elif case.get("shards"):
                    out = orc.compress_multi(data, params, case["shards"])
                else:
                    out = orc.compress(data, case["quality"], case["lgwin"])
            except orc.ReferencePanics:
                print(name, "seed %#x: the reference fails on this sharding, trying the next seed" % seed, flush=True)
                continue
            break
        else:
            raise SystemExit("no usable seed for " + name)
        t2 = time.time()
        assert orc.decompress(out, len(data)) == data
        frozen[name] = dict(input_bytes=len(data), input_sha256=hashlib.sha256(data).hexdigest(), stream_bytes=len(out),
                            stream_sha256=hashlib.sha256(out).hexdigest(), quality=case["quality"], lgwin=case["lgwin"],
                            shards=case.get("shards", 0), oracle_seconds=round(t2 - t1, 1))
        if seed is not None:
            frozen[name]["seed"] = "%#x" % seed
        print(name, frozen[name], "generate %.1fs" % (t1 - t0), flush=True)
        json.dump(frozen, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
