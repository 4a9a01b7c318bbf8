"""Bounded-memory streaming through the C ABI: BROTLI_OPERATION_PROCESS encodes the meta-blocks that are complete in the
input buffered so far and hands their bytes out before FLUSH / FINISH (BrotliEncoderHasMoreOutput is true mid-stream),
keeps only a ring buffer's worth of the stream as the window of what follows, and still produces exactly the bytes of
the reference's stream encoder fed with the same writes (orc_writer_compress / compress_stream of the oracle).

CPU: cabi.cpp / encoder.cpp / lz77_stage.cpp linked against the emulation seam; the batch size is turned down with
BROTLI_MI355X_STREAM_BATCH so that small inputs go through many pieces.  -m gpu: the product library."""
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
Q, W = 1, 2
cases = %(cases)s
for case in cases:
    name, gen, q, w, write = case[:5]
    hint = case[5] if len(case) > 5 else 0  # BROTLI_PARAM_SIZE_HINT set by the caller (decides H5 / H6, encode.rs:863-893)
    expect_early = case[6] if len(case) > 6 else True  # (a stream shorter than one batch is encoded by FINISH)
    data = eval(gen)
    e = lib.encoder(params=[(Q, q), (W, w)] + ([(5, hint)] if hint else []))
    early = 0  # output produced by PROCESS alone
    for i in range(0, len(data), write):
        e.write(data[i:i + write])
        early = len(e._out)
    mid = early
    got = e.finish()
    e.close()
    want = orc.reader_compress(data, [(Q, q), (W, w), (5, hint)], chunk=write) if hint else orc.writer_compress(data, q, w, chunk=write)
    assert got == want, (name, len(got), len(want))
    assert orc.decompress(got, len(data)) == data
    print("OK %%s: %%d -> %%d bytes, %%d handed out before FINISH" %% (name, len(data), len(got), mid))
    assert mid > len(got) // 2 or not expect_early, (name, "PROCESS did not produce output mid-stream", mid)
# FLUSH in the middle of a stream that has already been trimmed to its window, then more input
if not %(flush_part)r:
    sys.exit(0)
data = synth.markov_text(3 << 20, 5)
e = lib.encoder(params=[(Q, 5), (W, 17)])
pieces = []
# (fed like orc.stream_with_flushes feeds the oracle: the last write of every piece comes WITH the FLUSH / FINISH operation)
for i in range(0, (2 << 20) - 65536, 65536):
    e.write(data[i:i + 65536])
pieces.append(e.flush(data[(2 << 20) - 65536:2 << 20]))
for i in range(2 << 20, len(data) - 65536, 65536):
    e.write(data[i:i + 65536])
e._stream(2, data[len(data) - 65536:])
pieces.append(bytes(e._out))
e.close()
want = orc.stream_with_flushes(data, [(Q, 5), (W, 17)], [2 << 20], write_size=65536)
assert [len(p) for p in pieces] == [len(p) for p in want], ([len(p) for p in pieces], [len(p) for p in want])
assert pieces == want
print("OK flush in a trimmed stream")
'''


def _run(kind, cases, batch, wrap_shift=0, flush_part=True):
    env = dict(os.environ, BROTLI_MI355X_STREAM_BATCH=str(batch))
    code = _DRIVER % dict(tests=HERE, kind=kind, cases=repr(cases), flush_part=flush_part)
    if wrap_shift:
        # scale the position wrap of the reference (3, 5, 7 ... GiB) down to MiB, in the product and in the oracle
        env["BROTLI_MI355X_TEST_WRAP_SHIFT"] = str(wrap_shift)
        code = code.replace("lib = test_cabi._load", "import ctypes; ctypes.c_int.in_dll(orc.lib(), 'orc_test_wrap_shift').value = %d\nlib = test_cabi._load" % wrap_shift)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=3000)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


CASES = [
    # lgwin 17: 256 KiB ring, meta-blocks of at most 256 KiB -> with 512 KiB batches a 3 MiB stream is ~6 pieces, and the
    # window is trimmed from the second piece on (the u16 ring counters run on across the pieces: key_counts)
    ("markov 3 MiB q5 w17", "synth.markov_text(3 << 20, 3)", 5, 17, 65536),
    ("mixed 3 MiB q5 w17 odd writes", "synth.mixed(3 << 20, 4)", 5, 17, 100003),
    ("stretches 2 MiB q6 w18", "synth.stretches(2 << 20, 9)", 6, 18, 70000),
    ("markov 2 MiB q9 w17", "synth.markov_text(2 << 20, 8)", 9, 17, 65536),
]


def test_streaming_pieces_emu():
    _run("emu", CASES, 512 << 10)


@pytest.mark.gpu
def test_streaming_pieces_gpu():
    _run("gpu", CASES + [("markov 24 MiB q5 w18", "synth.markov_text(24 << 20, 6)", 5, 18, 1 << 20)], 512 << 10)


@pytest.mark.gpu
def test_streaming_default_batches_gpu():
    """the default 64 MiB batches at lgwin 22, fed in 4 MiB writes: a 200 MiB stream with a size hint (H6), and a 40 MiB one
    without (H5: masked ring entries from 8 MiB on, parsed by a live chain)"""
    _run("gpu", [("markov 200 MiB q5 w22 hinted", "synth.markov_text(200 << 20, 7)", 5, 22, 4 << 20, 1 << 30),
                 ("markov 40 MiB q5 w22", "synth.markov_text(40 << 20, 7)", 5, 22, 4 << 20, 0, False)], 64 << 20)


WRAP_CASES = [
    # unit 1 MiB: the hasher is reset when the stream passes 3, 5 and 7 MiB (encode.rs:1623-1631, 1705-1710) -- once
    # inside a piece, once exactly between two pieces, with the candidate rows (q5) and with the rank structures (q7)
    ("markov 8 MiB q5 w17, wraps at 3/5/7 MiB", "synth.markov_text(8 << 20, 21)", 5, 17, 65536),
    ("mixed 6 MiB q7 w17, wraps at 3/5 MiB", "synth.mixed(6 << 20, 22)", 7, 17, 50000),
]


def test_streaming_pieces_emu_small_batches():
    """trimmed windows and carried masked flags (StreamCarry) with batches as small as the ring"""
    _run("emu", [("markov 1.5 MiB q5 w17", "synth.markov_text(1536 << 10, 3)", 5, 17, 65536),
                 ("mixed 1 MiB q7 w17 odd writes", "synth.mixed(1 << 20, 4)", 7, 17, 100003)], 256 << 10, flush_part=False)


def test_hasher_reset_at_position_wrap_emu():
    _run("emu", WRAP_CASES, 512 << 10, wrap_shift=20)


@pytest.mark.gpu
def test_hasher_reset_at_position_wrap_gpu():
    _run("gpu", WRAP_CASES, 512 << 10, wrap_shift=20)
