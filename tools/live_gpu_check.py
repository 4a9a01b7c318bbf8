"""GPU bring-up of the live chains (lz77_live.h): product library vs the unmodified oracle on inputs with masked H5 entries,
and how long a live parse takes.  Usage: python tools/live_gpu_check.py [parity|speed] ..."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "rust-brotli_amd"))
import orc, synth, emu, gpulib
Q, W, SH = 1, 2, 5
L = gpulib.lib()

def chk(name, data, q, w):
    t = time.time()
    out, st = emu.encode_stream(L, data, [(Q, q), (W, w), (SH, len(data))])
    te = time.time() - t
    ref = orc.compress(data, q, w)
    ok = out == ref
    print("%-14s q%d w%d n=%d %s rounds=%d blocks=%d lz77=%.1fms total=%.2fs" % (name, q, w, len(data), "OK" if ok else "FAIL", st["lz77_rounds"], st["num_segments"], st["ms_lz77"], te), flush=True)
    return ok

mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
ok = True
if mode == "parity":
    a = synth.alice()
    ok &= chk("alice", a, 5, 22)
    ok &= chk("text1M", synth.markov_text(1 << 20), 5, 17)
    ok &= chk("mixed1M", synth.mixed(1 << 20), 5, 17)
    ok &= chk("text1.5M", synth.markov_text(2 << 20)[:1500000], 5, 18)
    d = synth.mixed(1 << 20)
    for q in (6, 7, 8):
        ok &= chk("mixed1M", d, q, 17)
    ok &= chk("text6M", synth.markov_text(6 << 20), 5, 17)
    ok &= chk("silesia4M", synth.silesia_like(4 << 20, 0x77, min_segment=1 << 16, max_segment=1 << 19), 5, 17)
    ok &= chk("stretches3M", synth.stretches(3 << 20), 5, 17)
    ok &= chk("zeros2M", bytes(2 << 20), 5, 17)
    print("ALL OK" if ok else "FAILED")
else:
    # a shard-like stream: H5 (size hint <= 1 MiB is what a shard's hasher sees), lgwin 22, text
    n = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 32 << 20
    data = synth.markov_text(n)
    for rep in range(2):
        t = time.time()
        out, st = emu.encode_stream(L, data, [(Q, 5), (W, 22), (SH, 1 << 20)])
        print("text %d MiB H5 lgwin22: rounds=%d blocks=%d lz77=%.1f ms total=%.2fs -> %.1f MB/s, %d bytes" % (n >> 20, st["lz77_rounds"], st["num_segments"], st["ms_lz77"], time.time() - t, n / (time.time() - t) / 1e6, len(out)), flush=True)
sys.exit(0 if ok else 1)
