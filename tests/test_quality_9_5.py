"""Quality "9.5" (SURVEY row b10): BROTLI_PARAM_Q9_5 with quality 10 keeps the greedy H9 search of quality 9 and runs the
quality >= 10 meta-block builder behind it -- the distance-parameter search, BrotliSplitBlock (FindBlocks / ClusterBlocks),
context histograms, BrotliClusterHistograms, BrotliPopulationCost (metablock.rs:133-307, block_splitter.rs, cluster.rs,
bit_cost.rs:76-211) -- all on the device (rust-brotli_amd/csrc/metablock_hq.h).

The reference holds two exact sizes for this path: src/bin/integration_tests.rs:397-428, random_then_unicode through
roundtrip_helper(.., 10, 28, q9_5) == 130 036 bytes and (.., 11, 22, q9_5) == 129 715 bytes (4096-byte reads, size hint
2 MiB; quality 11 searches 512-deep H5 / H6 rings instead of H9).  Everything else is byte identity with the oracle
(oracle/orc_hq_metablock.c, itself pinned on those sizes and on 47 488 / 46 493).

CPU: the emulation build (the same item code compiled for the host).  -m gpu: the product library."""
import glob
import os

import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
Q, W, MODE, SH, Q9_5, NO_CTX = 1, 2, 0, 5, 150, 4
P = [(Q, 10), (Q9_5, 1)]


def _kat(lib):
    d = open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read()
    # roundtrip_helper: lgwin 28 without large_window is cut back to 24 by the parameter check (encode.rs:657-676)
    for w in (28, 24):
        e = lib.encoder(params=P + [(W, w), (SH, 2048 * 1024)])
        for i in range(0, len(d), 4096):
            e.write(d[i:i + 4096])
        got = e.finish()
        e.close()
        assert len(got) == 130036
        assert got == orc.reader_compress(d, P + [(W, w), (SH, 2048 * 1024)], chunk=4096)
        assert orc.decompress(got, len(d)) == d


def _cases(small):
    a = synth.alice()
    yield "alice w22 hint", a, P + [(W, 22), (SH, len(a))], b""
    yield "alice w16", a, P + [(W, 16), (SH, len(a))], b""
    yield "alice no hint", a, P + [(W, 22)], b""
    for mode in (3, 4, 6):  # forced context modes (ChooseContextMode, encode.rs:1357-1377); 0-2 and 5 take the UTF-8 census
        yield "alice mode %d" % mode, a, P + [(W, 22), (MODE, mode)], b""
    yield "alice no literal contexts", a, P + [(W, 22), (NO_CTX, 1)], b""
    h = len(a) // 2
    yield "alice second half behind the first as dictionary", a[h:], P + [(W, 22), (167, 1), (168, 1)], a[:h]
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))):
        d = open(f, "rb").read()
        yield os.path.basename(f), d, P + [(W, 22), (SH, len(d))], b""
    yield "random 300k (stored raw)", synth.random_bytes(300000), P + [(W, 22)], b""
    yield "zeros 300k", bytes(300000), P + [(W, 22)], b""
    yield "stretches 1 MiB", synth.stretches(1 << 20, 9), P + [(W, 22)], b""
    yield "Silesia-like 2 MiB", synth.silesia_like(2 << 20, min_segment=1 << 18, max_segment=1 << 20), P + [(W, 22)], b""
    yield "mixed %s" % ("1 MiB" if small else "3 MiB"), synth.mixed((1 if small else 3) << 20), P + [(W, 22)], b""
    n = (2 if small else 5) << 20  # several meta-blocks, one with the signed context mode if the census says so
    d = synth.markov_text(n)
    yield "markov text", d, P + [(W, 22), (SH, n)], b""


def _identity(L, small):
    from cmp_stream import check_bytes
    bad = [name for name, data, params, prefix in _cases(small) if not check_bytes(L, name, data, params, prefix=prefix)]
    assert not bad, bad


def _flushes(lib):
    """FLUSH in the middle of a stream: every piece is built by the quality >= 10 builder, the carry goes on"""
    d = synth.mixed(1 << 20, 7)
    e = lib.encoder(params=P + [(W, 20)])
    pieces = []
    cuts = [300000, 700001]
    last = 0
    for c in cuts:
        pieces.append(e.flush(d[last:c]))
        last = c
    e._stream(2, d[last:])
    pieces.append(bytes(e._out))
    e.close()
    want = orc.stream_with_flushes(d, P + [(W, 20)], cuts, write_size=1 << 30)
    assert [len(p) for p in pieces] == [len(p) for p in want]
    assert pieces == want


_STREAMED = r"""
import sys
sys.path.insert(0, %(tests)r)
import orc, synth, test_cabi
lib = test_cabi._load(%(kind)r)
Q, W, SH = 1, 2, 5
for name, d, w, chunk in (("markov 6 MiB, lgwin 18", synth.markov_text(6 << 20, 11), 18, 65536),
                          ("mixed 4 MiB, lgwin 20", synth.mixed(4 << 20, 12), 20, 100003)):
    params = [(Q, %(quality)d), (150, 1), (W, w), (SH, 2 << 20)]
    e = lib.encoder(params=params)
    early = 0
    for i in range(0, len(d), chunk):
        e.write(d[i:i + chunk])
        early = max(early, len(e._out))
    got = e.finish()
    e.close()
    assert got == orc.reader_compress(d, params, chunk=chunk), name
    assert early > len(got) // 2, (name, "PROCESS handed nothing out", early)
    print("OK", name, len(got), early)
"""


def _streamed(kind, quality=10):
    """bounded-memory streaming (BROTLI_OPERATION_PROCESS hands out the meta-blocks that are complete, the window is trimmed):
    every piece is built by the quality >= 10 builder; the batch is turned down so that a few MiB go through several pieces"""
    import subprocess
    import sys
    env = dict(os.environ, BROTLI_MI355X_STREAM_BATCH=str(1 << 20))
    r = subprocess.run([sys.executable, "-c", _STREAMED % dict(tests=HERE, kind=kind, quality=quality)], env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_streamed_in_pieces_emu():
    _streamed("emu")
    _streamed("emu", 11)  # 512-deep rings


@pytest.mark.gpu
def test_streamed_in_pieces_gpu():
    _streamed("gpu")
    _streamed("gpu", 11)  # 512-deep rings (the deep chain kernels)


def test_reference_kat_130036_emu():
    import test_cabi
    _kat(test_cabi._load("emu"))


def test_identity_with_the_oracle_emu():
    import emu
    _identity(emu.lib(), small=False)


def test_flushes_emu():
    import test_cabi
    _flushes(test_cabi._load("emu"))


def _quality_11(lib, L, wide):
    """Quality 11 + Q9_5 selects H5 / H6 with 512-deep rings and 16 cache candidates (encode.rs:863-893): the chain kernels
    instantiated with the deep candidate scratch (ChainScratchT<.., kDeep>, lz77_chain.h).  The reference's second known answer
    for the path: random_then_unicode through roundtrip_helper(.., 11, 22, q9_5) == 129 715 bytes
    (src/bin/integration_tests.rs:397-428)."""
    from cmp_stream import check_bytes
    d = open(os.path.join(GOLDEN, "random_then_unicode"), "rb").read()
    params = [(Q, 11), (Q9_5, 1), (W, 22), (SH, 2048 * 1024)]
    e = lib.encoder(params=params)
    for i in range(0, len(d), 4096):
        e.write(d[i:i + 4096])
    got = e.finish()
    e.close()
    assert len(got) == 129715
    assert got == orc.reader_compress(d, params, chunk=4096)
    if not wide:
        return
    a = synth.alice()
    assert check_bytes(L, "alice q11 + Q9_5", a, [(Q, 11), (Q9_5, 1), (W, 22)])
    assert check_bytes(L, "alice q11 + Q9_5, lgwin 18", a, [(Q, 11), (Q9_5, 1), (W, 18), (SH, len(a))])
    assert check_bytes(L, "mixed 2 MiB q11 + Q9_5", synth.mixed(2 << 20), [(Q, 11), (Q9_5, 1), (W, 20)])
    m = synth.markov_text(6 << 20)
    assert check_bytes(L, "markov 6 MiB q11 + Q9_5 (H6)", m, [(Q, 11), (Q9_5, 1), (W, 22), (SH, len(m))])


def test_quality_11_with_q9_5_emu():
    """on the emulation build: the known answer and a wider identity set (host logic, scalar chain code at depth 512, the
    meta-block builder at quality 11)"""
    import emu
    import test_cabi
    _quality_11(test_cabi._load("emu"), emu.lib(), wide=True)


@pytest.mark.gpu
def test_quality_11_with_q9_5_gpu():
    """on the device: the known answer and the same identity set as the emulation test -- H5 and H6 at ring depth 512,
    lgwin 18 / 20 / 22 (the deep chain kernels: k_parse_segments<.., deep>, k_recheck_searches, k_parse_live, k_live_verify)"""
    import gpulib
    import test_cabi
    _quality_11(test_cabi._load("gpu"), gpulib.lib(), wide=True)


@pytest.mark.gpu
def test_reference_kat_130036_gpu():
    import test_cabi
    _kat(test_cabi._load("gpu"))


@pytest.mark.gpu
def test_identity_with_the_oracle_gpu():
    import gpulib
    _identity(gpulib.lib(), small=True)


@pytest.mark.gpu
def test_flushes_gpu():
    import test_cabi
    _flushes(test_cabi._load("gpu"))
