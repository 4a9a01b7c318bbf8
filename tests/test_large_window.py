"""BROTLI_PARAM_LARGE_WINDOW (lgwin 25..30, encode.rs:519-553 SanitizeParams, metablock.rs:28-60 BrotliInitDistanceParams with
the 62-bit-class distance alphabet, the 14-bit stream header of EncodeWindowBits): byte identity with the oracle at qualities
5..9 and "9.5".  The far-copy case puts a megabyte of random bytes 19 MiB in front of its repetition: only a window above
2^24 can reach it, so the stream uses distances no ordinary window has.  CPU: emulation build; -m gpu: the product library."""
import pytest

import synth

Q, LARGE, W, SH, Q9_5 = 1, 6, 2, 5, 150


def _cases(far):
    a = synth.alice()
    for q in (5, 7, 9):
        for w in (25, 28, 30):
            yield "alice q%d w%d" % (q, w), a, [(Q, q), (LARGE, 1), (W, w)]
    m = synth.mixed(3 << 20)
    yield "mixed q5 w26", m, [(Q, 5), (LARGE, 1), (W, 26)]
    yield "mixed q5 w30 hint", m, [(Q, 5), (LARGE, 1), (W, 30), (SH, len(m))]
    yield "mixed q6 w22 with the large-window alphabet", m, [(Q, 6), (LARGE, 1), (W, 22)]
    yield "alice q10 + Q9_5 w28", a, [(Q, 10), (Q9_5, 1), (LARGE, 1), (W, 28)]
    yield "alice catable appendable w27", a, [(Q, 5), (LARGE, 1), (W, 27), (167, 1), (168, 1)]
    if far:
        r = synth.random_bytes(1 << 20, 9)
        d = r + synth.markov_text(18 << 20, 5) + r + synth.markov_text(1 << 20, 6)
        yield "far copy w25 (H6)", d, [(Q, 5), (LARGE, 1), (W, 25), (SH, len(d))]


def _run(L, far):
    from cmp_stream import check_bytes
    bad = [name for name, data, params in _cases(far) if not check_bytes(L, name, data, params)]
    assert not bad, bad


def test_large_window_emu():
    import emu
    _run(emu.lib(), far=False)


@pytest.mark.gpu
def test_large_window_gpu():
    import gpulib
    import orc
    _run(gpulib.lib(), far=True)
    # (the far copy is really taken: the same input in a 2^24 window comes out a megabyte larger)
    r = synth.random_bytes(1 << 20, 9)
    d = r + synth.markov_text(18 << 20, 5) + r + synth.markov_text(1 << 20, 6)
    small = orc.stream_compress(d, [(Q, 5), (W, 24), (SH, len(d))])[0]
    large = orc.stream_compress(d, [(Q, 5), (LARGE, 1), (W, 25), (SH, len(d))])[0]
    assert len(large) + 900000 < len(small)
