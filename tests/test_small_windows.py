"""lgwin <= 16 at qualities 5..8 (and quality 11 + Q9_5): ChooseHasher picks hasher types 40 / 41 / 42 (encode.rs:855-862),
which have no implementation in the reference -- BrotliMakeHasher falls through to InitializeH6 with the untouched default
hasher parameters (encode.rs:1096-1114, 348-355): bucket_bits 15, 256-deep rings, hash_len 5, 16 last distances.  The product
maps them onto its H6 chains at that depth (the deep-ring kernels).  Byte identity with the oracle; CPU: emulation build,
-m gpu: the product library."""
import glob
import os

import pytest

import synth

Q, W, SH, Q9_5 = 1, 2, 5, 150
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases(small):
    a = synth.alice()
    for q in (5, 6, 7, 8):
        for w in ((16,) if small else (10, 13, 16)):
            yield "alice q%d w%d" % (q, w), a, [(Q, q), (W, w)]
    yield "alice q5 w14 hint", a, [(Q, 5), (W, 14), (SH, len(a))]
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))):
        d = open(f, "rb").read()
        for q, w in ((5, 16), (7, 14), (8, 16)):
            yield "%s q%d w%d" % (os.path.basename(f), q, w), d, [(Q, q), (W, w), (SH, len(d))]
    m = synth.mixed(2 << 20)
    yield "mixed 2 MiB q5 w16", m, [(Q, 5), (W, 16)]
    yield "mixed 2 MiB q8 w15", m, [(Q, 8), (W, 15)]
    t = synth.markov_text(3 << 20)
    yield "markov 3 MiB q6 w16 hint", t, [(Q, 6), (W, 16), (SH, len(t))]
    yield "alice q11 + Q9_5 w16", a, [(Q, 11), (Q9_5, 1), (W, 16)]
    yield "alice catable appendable q7 w16", a, [(Q, 7), (W, 16), (167, 1), (168, 1)]


def _run(L, small):
    from cmp_stream import check_bytes
    bad = [name for name, data, params in _cases(small) if not check_bytes(L, name, data, params)]
    assert not bad, bad


def test_small_windows_emu():
    import emu
    _run(emu.lib(), small=False)


@pytest.mark.gpu
def test_small_windows_gpu():
    import gpulib
    _run(gpulib.lib(), small=False)
