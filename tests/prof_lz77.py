"""LZ77-stage timing probe on the GPU (not a test): prof_lz77.py <MiB> <segment bytes, 0 = auto> <kind>"""
import sys, time
import synth, emu, gpulib
L = gpulib.lib()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
n = mb << 20
d = {"text": synth.markov_text, "random": synth.random_bytes, "mixed": synth.mixed, "silesia": synth.silesia_like, "enwik": synth.enwik_like,
     "zeros": lambda n: bytes(n), "silesia_fine": lambda n: synth.silesia_like(n, min_segment=64 << 10, max_segment=2 << 20)}[kind](n)
for it in range(2):
    t = time.time()
    mbs, st = emu.lz77_trace(L, d, 5, 22, len(d), False, b"", seg)
    dt = time.time() - t
    print("%s n=%d MiB seg=%d wall=%.3fs rounds=%d parsed=%d searches=%d cmds=%d ms: %s" % (
        kind, mb, seg, dt, st['rounds'], st['segments_parsed'], st['searches'], st['total_cmds'],
        " ".join("%s=%.1f" % (a, b) for a, b in st['ms'].items())))
