"""GPU byte parity of complete streams: HIP encoder output == oracle output."""
import glob
import os

import pytest

import synth
from cmp_stream import check_bytes

pytestmark = pytest.mark.gpu
Q, W, SH = 1, 2, 5
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def L():
    import gpulib
    return gpulib.lib()


@pytest.mark.parametrize("q", [5, 6, 7, 8])
def test_alice(L, q):
    a = synth.alice()
    assert check_bytes(L, "alice", a, [(Q, q), (W, 22), (SH, len(a))])


def test_alice_variants(L):
    a = synth.alice()
    assert check_bytes(L, "alice", a, [(Q, 5), (W, 22)])
    assert check_bytes(L, "alice", a, [(Q, 5), (W, 18), (SH, len(a))])
    assert check_bytes(L, "alice", a, [(Q, 5), (W, 22), (168, 1), (169, 1)])
    assert check_bytes(L, "alice", a, [(Q, 5), (W, 22), (167, 1)])
    assert check_bytes(L, "alice", a, [(Q, 6), (W, 22), (168, 1), (172, 1)])


def test_small_fixtures(L):
    for f in sorted(glob.glob(os.path.join(GOLDEN, "small", "*"))):
        d = open(f, "rb").read()
        assert check_bytes(L, os.path.basename(f), d, [(Q, 5), (W, 22), (SH, len(d))])
        assert check_bytes(L, os.path.basename(f), d, [(Q, 5), (W, 22), (167, 1)])


def test_markov_h6_multi_metablock(L):
    d = synth.markov_text(6 << 20)
    assert check_bytes(L, "markov6M", d, [(Q, 5), (W, 22), (SH, len(d))])


def test_random_zeros_mixed(L):
    assert check_bytes(L, "random300k", synth.random_bytes(300000), [(Q, 5), (W, 22)])
    assert check_bytes(L, "zeros300k", bytes(300000), [(Q, 5), (W, 22)])
    assert check_bytes(L, "mixed3M", synth.mixed(3 << 20), [(Q, 5), (W, 22)])


def test_shard_with_prefix(L):
    a = synth.alice()
    h = len(a) // 2
    assert check_bytes(L, "alice shard1", a[h:], [(Q, 5), (W, 22), (167, 1), (168, 1)], prefix=a[:h])


def test_quality9_kat_51737(L):
    """the reference's own known answer (src/enc/encode.rs:3091): alice29 at quality 9, lgwin 16, one shot -> 51 737 B"""
    import emu
    a = synth.alice()
    out, _ = emu.encode_stream(L, a, [(Q, 9), (W, 16), (SH, len(a))])
    assert len(out) == 51737
    assert check_bytes(L, "alice q9 w16", a, [(Q, 9), (W, 16), (SH, len(a))])
    assert check_bytes(L, "alice q9", a, [(Q, 9), (W, 22), (SH, len(a))])
    d = synth.markov_text(5 << 20)
    assert check_bytes(L, "markov5M q9", d, [(Q, 9), (W, 22), (SH, len(d))])


def test_input_on_which_the_reference_fails(L):
    """see tests/test_emu_parity.py: a match cut to one byte at the end of the custom dictionary makes the reference panic;
    the HIP path must refuse the input with a message (it used to fault on the copy-length table)"""
    import emu
    x = open(os.path.join(GOLDEN, "copy_of_length_one.bin"), "rb").read()
    with pytest.raises(RuntimeError, match="reference encoder fails"):
        emu.encode_stream(L, x[64:], [(Q, 6), (W, 20)], prefix=x[:64])
    assert check_bytes(L, "no boundary", x, [(Q, 6), (W, 20)], verbose=False)


def test_catable_stream_recheck_at_block_starts(L):
    """the GPU twin of test_emu_parity.py::test_catable_stream_recheck_at_block_starts (round 6, API sweep seed 62 case 213): the first
    two positions of every block of a catable stream were re-checked against the end of the block in front (k_recheck_searches)"""
    d = open(os.path.join(GOLDEN, "catable_recheck_at_block_start.bin"), "rb").read()
    for q in (6, 7, 8, 9):
        assert check_bytes(L, "catable recheck q%d" % q, d, [(1, q), (2, 24), (5, len(d)), (167, 1)], seg=512)
