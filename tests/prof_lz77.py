"""LZ77-stage timing probe on the GPU (not a test)."""
import sys, time
import synth, emu, gpulib
L = gpulib.lib()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = synth.markov_text(mb << 20)
for it in range(2):
    t = time.time()
    mbs, st = emu.lz77_trace(L, d, 5, 22, len(d), False, b"", seg)
    dt = time.time() - t
    print("n=%d MiB seg=%d wall=%.3fs rounds=%d parsed=%d searches=%d cmds=%d ms=%s" % (mb, seg, dt, st['rounds'], st['segments_parsed'], st['searches'], st['total_cmds'], st['ms']))
