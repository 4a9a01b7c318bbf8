"""Compares the device's final per-position "stored in the hash table" flags with the truth recorded by the oracle's
hash table (test hook orc_set_debug_store_map).  Byte parity of the stream only exercises the flags that some later
search happened to depend on; this checks every position."""
import ctypes

import emu
import orc


def stored_flags_match(L, data, quality=5, lgwin=22, segment_bytes=4096):
    O = orc.lib()
    truth = ctypes.create_string_buffer(max(1, len(data)))
    O.orc_set_debug_store_map(truth, len(data))
    try:
        orc.compress(data, quality, lgwin)
    finally:
        O.orc_set_debug_store_map(None, 0)
    mine = ctypes.create_string_buffer(max(1, len(data)))
    L.brotli_mi355x_debug_set_flag_dump(mine, len(data))
    try:
        emu.lz77_trace(L, data, quality, lgwin, len(data), False, b"", segment_bytes)
    finally:
        L.brotli_mi355x_debug_set_flag_dump(None, 0)
    t, m = truth.raw, mine.raw
    bad = [i for i in range(len(data)) if (t[i] ^ m[i]) & 1]
    if bad:
        print("stored flags differ at %d positions, first %s" % (len(bad), bad[:8]))
    return not bad


def tricky_inputs():
    """inputs whose parse has steps that run across segment and block boundaries: long copies and copy extensions
    (zero / periodic runs), sparse-store jumps (incompressible stretches), and text in between"""
    import synth
    text = synth.markov_text(400000)
    rnd = synth.random_bytes(150000)
    yield "text+zeros+text", text[:150000] + bytes(200000) + text[150000:300000]
    yield "random+period+random", rnd[:70000] + (rnd[1000:1321] * 700)[:180001] + rnd[70000:]
    yield "zeros+random+zeros+text", bytes(70001) + rnd[:66000] + bytes(131073) + text[:60000]
    yield "text+random+text", text[:100000] + rnd[:140000] + text[100000:200000]
    # regression: a chain that first started from a wrong (too early) position left its flags behind in a stretch that
    # the final parse covers with an extended copy of the previous block
    yield "repeated excerpts", synth.repeated_excerpts(3 << 20, 1)[:1700000]
