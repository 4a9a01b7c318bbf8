"""Randomised sweep over the stream operations and the multi-shard entry point (not a test): fuzz_api.py <cases> <seed>
[emu].  FLUSH at random cut points, EMIT_METADATA, BrotliEncoderCompressMulti with 1..9 shards; expected bytes from the
oracle.  FUZZ_Q9_5=1 (or =11): every case at quality 10 (11) with BROTLI_PARAM_Q9_5 (the quality >= 10 meta-block builder, row b10).
FUZZ_ZOPFLI=10 (or =11): every case at quality 10 (11) proper (row f1: H10 + Zopfli, zopfli_device.h); inputs up to 300 KB.
FUZZ_FRAGMENT=1: every case at quality 0 or 1 (row f3: the fragment compressors, fragment_device.h), catable streams, custom
dictionaries and shards (the ring-buffer path of these qualities) included; no metadata block behind input on a catable stream (the
reference does not return from that call).
FUZZ_QUICK=1: every case at quality 2, 3 or 4 (row f3: the BasicHasher family, quick_device.h), windows down to lgwin 10."""
import os, sys, time
import synth, orc
import test_cabi
kind = "emu" if len(sys.argv) > 3 and sys.argv[3] == "emu" else "gpu"
lib = test_cabi._load(kind)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = synth.XorShift(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
Q, W, MAGIC = 1, 2, 169
pools = [synth.markov_text(3 << 20, 31), synth.mixed(3 << 20, 32), synth.silesia_like(3 << 20, 33, min_segment=8 << 10, max_segment=200 << 10),
         synth.stretches(3 << 20, 34), synth.repeated_excerpts(3 << 20, 35)]
bad = 0
panics = 0
t0 = time.time()
for c in range(cases):
    pool = pools[rng.next() % len(pools)]
    n = 1 + rng.next() % (1200000 if rng.next() % 3 else 5000)
    if os.environ.get("FUZZ_TINY"):
        n = rng.next() % 300
    if os.environ.get("FUZZ_MAXN"):
        n = 1 + n % int(os.environ["FUZZ_MAXN"])
    o = rng.next() % (len(pool) - n)
    d = pool[o:o + n]
    q = 5 + rng.next() % 5
    q95 = bool(os.environ.get("FUZZ_Q9_5"))
    if q95:
        q = 11 if os.environ["FUZZ_Q9_5"] == "11" else 10  # (11: the 512-deep H5 / H6 rings)
    zopfli = os.environ.get("FUZZ_ZOPFLI")
    if zopfli:
        q = int(zopfli)
        if n > 300000:
            n = 1 + n % 300000
            d = pool[o:o + n]
    w = [17, 18, 20, 22, 24][rng.next() % 5]
    if os.environ.get("FUZZ_QUICK"):
        q = 2 + rng.next() % 3
        w = [10, 13, 16, 17, 18, 20, 22, 24][rng.next() % 8]
    mode = rng.next() % 6
    fragment = bool(os.environ.get("FUZZ_FRAGMENT"))
    if fragment:
        q = rng.next() % 2
        w = [10, 13, 16, 17, 18, 20, 22, 24][rng.next() % 8]
        mode = rng.next() % 6
    if q95 and mode == 5:
        mode = 4  # (orc.writer_compress takes quality and window only)
    base = [(Q, q), (W, w)] + ([(150, 1)] if q95 else [])
    extra = []
    for pid in (167, 168, 169, 172):  # catable, appendable, magic number, byte align
        if rng.next() % 4 == 0:
            extra.append((pid, 1))
    if rng.next() % 3 == 0:
        extra.append((5, [1, 1000, 1 << 20, (1 << 20) + 1, 5 << 20][rng.next() % 5]))  # size hint
    if os.environ.get("FUZZ_TRACE"):
        print("case %d n %d q %d w %d mode %d" % (c, n, q, w, mode), flush=True)
        open(os.environ["FUZZ_TRACE"], "wb").write(d)
    # flushes on catable / appendable streams and streams with a custom dictionary (row f4): a third of the flush cases
    fparams, fdic = base, None
    if mode in (1, 2) and rng.next() % 3 == 0:
        fparams = base + [x for x in extra if x[0] != 5]
        if rng.next() % 2:
            mdic = 2 + rng.next() % 300000
            odic = rng.next() % (len(pool) - mdic)
            fdic = pool[odic:odic + mdic]

    def flushed(ops):
        e = lib.encoder(params=fparams, dictionary=fdic)
        pieces, pos = [], 0
        for item in ops:
            if isinstance(item, tuple):
                cpos, meta = item
                if cpos > pos:
                    e.write(d[pos:cpos])
                pieces.append(e.emit_metadata(meta))
            else:
                cpos = item
                pieces.append(e.flush(d[pos:cpos]))
            pos = cpos
        e.write(d[pos:])
        pieces.append(e.finish())
        e.close()
        return pieces

    if mode == 0:
        nt = 1 + rng.next() % 9
        what = "multi nt=%d" % nt
        product = lambda: bytes(lib.BrotliCompress(d, dict(base), nt))
        oracle = lambda: orc.compress_multi(d, base, nt) if nt > 1 else orc.stream_compress(d, base)[0]
    elif mode == 3:
        # custom LZ77 dictionary (BrotliEncoderSetCustomDictionary)
        m = 1 + rng.next() % 400000
        o2 = rng.next() % (len(pool) - m)
        dic = pool[o2:o2 + m]
        what = "dictionary %d B" % m
        if os.environ.get("FUZZ_TRACE"):
            open(os.environ["FUZZ_TRACE"] + ".dict", "wb").write(dic)

        def product():
            e = lib.encoder(params=base, dictionary=dic)
            e.write(d)
            got = e.finish()
            e.close()
            return got
        oracle = lambda: orc.stream_compress(d, base, prefix=dic, continuation=False)[0]
    elif mode == 5:
        # CompressorWriter feeding pattern (src/enc/writer.rs:183-313): PROCESS calls of one buffer each, then FINISH
        chunk = [1000, 4096, 65536, 100000, 1 << 20][rng.next() % 5]
        what = "writer chunk %d" % chunk

        def product():
            e = lib.encoder(params=[(Q, q), (W, w)])
            for i in range(0, len(d), chunk):
                e.write(d[i:i + chunk])
            got = e.finish()
            e.close()
            return got
        oracle = lambda: orc.writer_compress(d, q, w, chunk=chunk)
    elif mode == 4:
        # one FINISH with extra stream parameters (everything offered in one call, as the oracle helper does)
        what = "params %r" % (extra,)

        def product():
            e = lib.encoder(params=base + extra)
            e.write(d)
            got = e.finish()
            e.close()
            return got
        oracle = lambda: orc.stream_compress(d, base + extra)[0]
    else:
        ncut = 1 + rng.next() % 4
        cuts = sorted(rng.next() % (n + 1) for _ in range(ncut))
        ops = []
        # (quality 0 / 1: a metadata block behind input on a catable stream -- also one with a custom dictionary -- never returns in
        # the reference, encode.rs:2621-2629 with :2335-2389)
        no_meta = fragment and (fdic is not None or any(k == 167 for k, _ in fparams))
        for cut in cuts:
            if mode == 2 and rng.next() % 2 and not no_meta:
                ops.append((cut, bytes([65 + (rng.next() % 26)]) * (2 + rng.next() % 300)))
            else:
                ops.append(cut)
        what = "ops %r params %r dictionary %s" % ([x if not isinstance(x, tuple) else (x[0], len(x[1])) for x in ops], fparams[2:],
                                                   len(fdic) if fdic else None)
        product = lambda: flushed(ops)
        oracle = lambda: orc.stream_with_flushes(d, fparams, ops, dictionary=fdic)
    if os.environ.get("FUZZ_ONLY") and c != int(os.environ["FUZZ_ONLY"]):
        continue
    # an input on which the reference itself fails (it panics on a copy of length 1, see orc.ReferencePanics) must make
    # the product fail too, with the message that says so
    try:
        want = oracle()
    except orc.ReferencePanics:
        want = "reference fails"
    try:
        got = product()
    except Exception as ex:
        # (BrotliEncoderCompressMulti with the output bound of the binding: a stream that outgrows it fails the call on both sides)
        # (... and a quality 0 / 1 shard that outgrows BrotliEncoderMaxCompressedSize of its length: compress_part's fixed buffer in the
        # reference, a refusal here -- DESIGN.md section 3.10)
        got = "reference fails" if ("reference encoder fails" in str(ex) or "insufficient output space" in str(ex) or
                                    "a shard of quality 0 / 1 outgrows" in str(ex)) else "EXCEPTION %r" % (ex,)
    ok = got == want
    if want == "reference fails":
        panics += 1
    elif ok and mode != 3 and fdic is None:  # (a stream that refers to a custom dictionary needs that dictionary to decode)
        try:
            ok = orc.decompress(got if isinstance(got, bytes) else b"".join(got), len(d)) == d
        except RuntimeError as ex:
            ok = False
            what += " -> identical to the oracle, but: %s" % ex
    if not ok and isinstance(got, str):
        what += " -> " + got[:200]
    if not ok:
        bad += 1
        print("FAIL case %d n %d q %d w %d %s" % (c, n, q, w, what), flush=True)
        open("/tmp/fuzzapi_fail_%d.bin" % c, "wb").write(d)
print("%d cases, %d failures, %d on which the reference fails (and so does the product), %.1f s" % (cases, bad, panics, time.time() - t0))
sys.exit(1 if bad else 0)
