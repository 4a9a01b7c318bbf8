"""Quality 9.5 probe (GPU): three inputs through the product library's flat entry point, compared with the oracle, with the
stage times of the call (BROTLI_MI355X_PROFILE=1 adds the LZ77 stage's own split on stderr).  phases[1..3] = census +
distance parameters + FindBlocks, ClusterBlocks, context maps (encoder.cpp).  Used for profiles/r03_q9_5_probe_*.log:
    BROTLI_MI355X_PROFILE=1 python tools/hq_probe.py
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hq -o hq -- python tools/hq_probe.py"""
import sys, time, os
sys.path.insert(0, 'tests')
import gpulib, emu, orc, synth
L = gpulib.lib()
Q, W, SH = 1, 2, 5
d = open('tests/golden/random_then_unicode', 'rb').read()
P = [(Q, 10), (150, 1)]
for name, data, params in (("rtu", d, P + [(W, 24), (SH, 2048 * 1024)]), ("alice", synth.alice(), P + [(W, 22)]),
                           ("mixed1M", synth.mixed(1 << 20), P + [(W, 22)])):
    t = time.time()
    c, st = emu.encode_stream(L, data, params)
    dt = time.time() - t
    o, _ = orc.stream_compress(data, params)
    print(name, len(data), len(c), "identical" if c == o else "DIFFERENT", "%.2fs" % dt, [round(x, 1) for x in st["ms_phase"][:9]], flush=True)
