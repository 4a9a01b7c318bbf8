"""Randomised parity sweep (not a test): fuzz_gpu.py <cases> <seed> [emu [max MiB]] -- on the GPU, or with `emu` on the host
emulation build (same host driver and chain code, no GPU needed).  Every case = (content, size, quality,
lgwin, segment size) -> the HIP path's stream must equal the oracle's one-shot stream."""
import os, sys, time
import synth, emu, orc
use_emu = len(sys.argv) > 3 and sys.argv[3] == "emu"
max_bytes = int(float(sys.argv[4]) * (1 << 20)) if len(sys.argv) > 4 else (3 << 20)
if use_emu:
    L = emu.lib()
else:
    import gpulib
    L = gpulib.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = synth.XorShift(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = max(6 << 20, 2 * max_bytes)
pool_text = synth.markov_text(big, 77)
pool_mixed = synth.mixed(big, 78)
pool_fine = synth.silesia_like(big, 79, min_segment=8 << 10, max_segment=256 << 10)
pool_bin = synth.silesia_like(3 << 20, 80, only=60)
pool_hex = synth.silesia_like(3 << 20, 81, only=85)
pool_rand = synth.random_bytes(3 << 20, 82)
pool_rep = synth.repeated_excerpts(max(4 << 20, max_bytes + (1 << 20)), 3)


def make(kind, n):
    if kind == 0:
        o = rng.next() % (len(pool_text) - n)
        return pool_text[o:o + n]
    if kind == 1:
        o = rng.next() % (len(pool_mixed) - n)
        return pool_mixed[o:o + n]
    if kind == 2:
        o = rng.next() % (len(pool_fine) - n)
        return pool_fine[o:o + n]
    if kind == 3:
        # stretches of random / zero / text / binary of random lengths
        out = bytearray()
        while len(out) < n:
            k = rng.next() % 5
            m = 1 + rng.next() % (200000 if k else 20000)
            src = [pool_rand, None, pool_text, pool_bin, pool_hex][k]
            if src is None:
                out += bytes(m)
            else:
                o = rng.next() % max(1, len(src) - m)
                out += src[o:o + m]
        return bytes(out[:n])
    o = rng.next() % (len(pool_rep) - n)
    return pool_rep[o:o + n]


bad = 0
t0 = time.time()
for c in range(cases):
    kind = rng.next() % 5
    n = 1 + rng.next() % max_bytes if rng.next() % 4 else 1 + rng.next() % 70000
    if os.environ.get("FUZZ_TINY"):
        n = rng.next() % 300
    q = 5 + rng.next() % 5
    w = [17, 18, 20, 22, 24][rng.next() % 5]
    seg = [0, 0, 256, 512, 1024, 4096][rng.next() % 6]
    if os.environ.get("FUZZ_SEG"):
        seg = int(os.environ["FUZZ_SEG"])
    d = make(kind, n)
    try:
        out, st = emu.encode_stream(L, d, [(1, q), (2, w), (5, len(d))], segment_bytes=seg)
    except RuntimeError as e:
        bad += 1
        print("ERROR case %d kind %d n %d q %d w %d seg %d: %s" % (c, kind, n, q, w, seg, e), flush=True)
        open("/tmp/fuzz_fail_%d_k%d_q%d_w%d_seg%d.bin" % (c, kind, q, w, seg), "wb").write(d)
        continue
    # (the one-shot entry point answers an empty input with the single byte 6; the stream path compared here does not)
    want = orc.compress(d, q, w) if len(d) else orc.stream_compress(d, [(1, q), (2, w)])[0]
    ok = out == want
    if not ok:
        bad += 1
        print("MISMATCH case %d kind %d n %d q %d w %d seg %d" % (c, kind, n, q, w, seg), flush=True)
        open("/tmp/fuzz_fail_%d_k%d_q%d_w%d_seg%d.bin" % (c, kind, q, w, seg), "wb").write(d)
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
