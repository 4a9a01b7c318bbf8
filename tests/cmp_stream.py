"""byte parity of a full stream: product encoder (emu or GPU library) vs the oracle"""
import glob, os, time
import orc, emu, synth


def check_bytes(L, name, data, params, prefix=b"", seg=4096, verbose=True):
    t = time.time()
    c, st = emu.encode_stream(L, data, params, prefix=prefix, segment_bytes=seg)
    te = time.time() - t
    o, _ = orc.stream_compress(data, params, prefix=prefix if prefix else None)
    ok = c == o
    msg = ""
    if not ok:
        i = next((i for i, (x, y) in enumerate(zip(c, o)) if x != y), min(len(c), len(o)))
        msg = " first diff at byte %d (sizes %d vs %d)" % (i, len(c), len(o))
        if not prefix:
            try:
                msg += " decodes=%s" % (orc.decompress(c, len(data)) == data)
            except Exception as e:
                msg += " decode error %s" % e
    if verbose:
        print("%-26s n=%-9d %s %s out=%d rounds=%d retries=%d %.2fs%s" % (name, len(data), params, "OK" if ok else "FAIL", len(c), st["lz77_rounds"], st["fallback_retries"], te, msg))
    return ok


if __name__ == "__main__":
    L = emu.lib()
    allok = True
    Q, W, SH = 1, 2, 5
    for f in sorted(glob.glob("/root/reference/testdata/*")):
        b = os.path.basename(f)
        if "compressed" in b and b not in ("compressed_file", "compressed_repeated"):
            continue
        d = open(f, "rb").read()
        for q in (5, 7):
            allok &= check_bytes(L, b, d, [(Q, q), (W, 22), (SH, len(d))])
        allok &= check_bytes(L, b, d, [(Q, 5), (W, 22)])  # size_hint from the single write
        allok &= check_bytes(L, b, d, [(Q, 5), (W, 18), (SH, len(d))])
        allok &= check_bytes(L, b, d, [(Q, 5), (W, 22), (168, 1), (169, 1)])  # appendable + magic
        allok &= check_bytes(L, b, d, [(Q, 5), (W, 22), (167, 1)])  # catable
        allok &= check_bytes(L, b, d, [(Q, 6), (W, 22), (168, 1), (172, 1)])  # appendable + byte_align
    print("ALL OK" if allok else "SOME FAILED")
