"""tools/synth_gen.c must produce exactly what the pure-Python generators of synth.py produce."""
import synth


def test_c_generators_match_python():
    assert synth._c() is not None, "tools/libsynth.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    for n, seed in ((0, 1), (1, 2), (7, 0), (4096, 0x5EED000000000002), (300001, 0x1234), (1 << 20, 0x5EED000000000005)):
        assert synth.markov_text(n, seed) == synth.markov_text_py(n, seed)
        assert synth.random_bytes(n, seed) == synth.random_bytes_py(n, seed)
