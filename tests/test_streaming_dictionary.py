"""Streams with a custom dictionary, catable and appendable streams, piece by piece: FLUSH anywhere (also inside
the two raw first bytes of a catable stream), and BROTLI_OPERATION_PROCESS handing out the meta-blocks that are complete while
only a window of the stream is kept -- the bytes of the reference's stream encoder fed the same way (encode.rs:1196-1270 for
the dictionary, :2283-2333 for the raw first bytes of a catable stream, mod.rs:42-54 for the dictionary-end rule, which the
reference keeps applying at the same RING index for the whole stream).  Until round 3 such streams were buffered whole until
FINISH and could not be flushed.

CPU: the emulation build with the batch size turned down (BROTLI_MI355X_STREAM_BATCH) so that small inputs go through many
pieces.  -m gpu: the product library."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

_DRIVER = r'''
import sys
sys.path.insert(0, %(tests)r)
import orc, synth, test_cabi
lib = test_cabi._load(%(kind)r)
Q, W, CAT, APP, MAGIC, ALIGN = 1, 2, 167, 168, 169, 172


def run(name, d, params, cuts, dic=None, write=0, expect_early=False):
    e = lib.encoder(params=params, dictionary=dic)
    pieces, pos, early = [], 0, 0
    for c in list(cuts) + [None]:
        end = len(d) if c is None else c
        if write:
            while end - pos > write:
                e.write(d[pos:pos + write])
                pos += write
                early = max(early, len(e._out))
        if c is None:
            e._stream(2, d[pos:])
            pieces.append(bytes(e._out))
        else:
            pieces.append(e.flush(d[pos:end]))
        pos = end
    e.close()
    want = orc.stream_with_flushes(d, params, list(cuts), write_size=write, dictionary=dic)
    assert [len(p) for p in pieces] == [len(p) for p in want], (name, [len(p) for p in pieces], [len(p) for p in want])
    assert pieces == want, name
    if expect_early:
        assert early > 100000, (name, "PROCESS handed nothing out by itself", early)  # (bytes seen before a FLUSH / FINISH asked)
    print("OK %%s: %%d -> %%s bytes" %% (name, len(d), [len(p) for p in pieces]))


a = synth.alice()
m = synth.markov_text(3 << 20, 5)
mix = synth.mixed(2 << 20, 3)
# FLUSH with a custom dictionary
run("dictionary, two flushes", a[50000:], [(Q, 5), (W, 22)], [30000, 70000], dic=a[:50000])
run("dictionary q6 w18, flush twice at one point", m[200000:1500000], [(Q, 6), (W, 18)], [400000, 400001, 900000], dic=m[:200000])
run("dictionary q9", a[30000:], [(Q, 9), (W, 20)], [60000], dic=a[:30000])
# the raw first bytes of a catable stream against flushes at 0, 1, 2, 3 bytes
for cuts in ([1, 5000], [1, 3], [1, 2, 5000], [0, 1, 1, 2, 9], [2, 70000], [3, 4, 5], [100000]):
    run("catable, flushes at %%r" %% (cuts,), a, [(Q, 5), (W, 22), (CAT, 1)], cuts)
run("catable + magic number", a, [(Q, 6), (W, 18), (CAT, 1), (MAGIC, 1)], [1, 40000])
run("appendable", mix, [(Q, 5), (W, 20), (APP, 1)], [100, 1 << 20])
run("appendable + byte align, flush twice at one point", a, [(Q, 5), (W, 22), (APP, 1), (ALIGN, 1)], [70000, 70000])
run("dictionary + catable", a[1000:], [(Q, 5), (W, 22), (CAT, 1)], [40000], dic=a[:1000])
# a flush with nothing to search yet behind a dictionary: the table holds the bare dictionary, and the context bytes of the
# first meta-block still read as 0 (13 literal contexts with this size hint, so they matter)
for q in (5, 6, 9):
    run("dictionary q%%d, flush before any input" %% q, m[300000:400000], [(Q, q), (W, 22), (5, 5 << 20)], [0, 0, 50000], dic=m[:300000])
run("dictionary + catable, flushes at 0, 1, 2, 3", a[5000:], [(Q, 5), (W, 22), (CAT, 1)], [0, 1, 2, 3, 40000], dic=a[:5000])
run("dictionary of three bytes, flush at 0", a[3:], [(Q, 5), (W, 22)], [0, 40000], dic=a[:3])
# PROCESS in bounded memory: the window is trimmed behind the dictionary, the dictionary-end rule stays
run("dictionary, 64 KiB writes, lgwin 17", m[100000:], [(Q, 5), (W, 17)], [1 << 20, 2 << 20], dic=m[:100000], write=65536, expect_early=True)
run("dictionary, odd writes, no flush", m[100000:], [(Q, 5), (W, 17)], [], dic=m[:100000], write=100003, expect_early=True)
run("catable, 64 KiB writes", m, [(Q, 5), (W, 18), (CAT, 1)], [], write=65536, expect_early=True)
run("appendable, 70000-byte writes", mix, [(Q, 6), (W, 18), (APP, 1)], [1 << 20], write=70000, expect_early=True)
'''


def _run(kind, batch):
    env = dict(os.environ, BROTLI_MI355X_STREAM_BATCH=str(batch))
    code = _DRIVER % dict(tests=HERE, kind=kind)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=3000)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_dictionary_catable_appendable_streams_piece_by_piece_emu():
    _run("emu", 512 << 10)


@pytest.mark.gpu
def test_dictionary_catable_appendable_streams_piece_by_piece_gpu():
    _run("gpu", 512 << 10)
