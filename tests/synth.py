"""Deterministic synthetic inputs: xorshift64* PRNG, word-bigram Markov English-like text."""
import os
MASK = (1 << 64) - 1


class XorShift:
    def __init__(self, seed):
        self.x = seed & MASK or 0x9E3779B97F4A7C15

    def next(self):
        x = self.x
        x ^= x >> 12
        x ^= (x << 25) & MASK
        x ^= x >> 27
        self.x = x
        return (x * 0x2545F4914F6CDD1D) & MASK


_ALICE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alice29.txt")


def alice():
    for p in (_ALICE, "/root/reference/testdata/alice29.txt"):
        if os.path.exists(p):
            return open(p, "rb").read()
    raise FileNotFoundError("alice29.txt fixture")


_clib = None
_markov_handle = None


def _c():
    """tools/libsynth.so (tools/synth_gen.c): the same generators at several hundred MB/s; None if it is not built"""
    global _clib
    if _clib is None:
        import ctypes
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "libsynth.so")
        if not os.environ.get("SYNTH_PURE_PYTHON") and not os.path.exists(path):
            import subprocess
            try:
                subprocess.check_call(["make", "-s", "-C", os.path.dirname(path)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except Exception:
                pass
        if os.environ.get("SYNTH_PURE_PYTHON") or not os.path.exists(path):
            _clib = False
        else:
            L = ctypes.CDLL(path)
            L.synth_xorshift.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.c_char_p]
            L.synth_xorshift.restype = None
            L.synth_markov_new.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
            L.synth_markov_new.restype = ctypes.c_void_p
            L.synth_markov_text.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_char_p]
            L.synth_markov_text.restype = None
            _clib = L
    return _clib or None


def markov_text(nbytes, seed=0x5EED000000000002):
    L = _c()
    if L is not None:
        import ctypes
        global _markov_handle
        if _markov_handle is None:
            a = alice()
            _markov_handle = L.synth_markov_new(a, len(a))
        buf = ctypes.create_string_buffer(max(1, nbytes))
        L.synth_markov_text(_markov_handle, seed & MASK, nbytes, buf)
        return buf.raw[:nbytes]
    return markov_text_py(nbytes, seed)


def markov_text_py(nbytes, seed=0x5EED000000000002):
    toks = alice().split()
    nxt = {}
    for a, b in zip(toks, toks[1:]):
        nxt.setdefault(a, []).append(b)
    rng = XorShift(seed)
    out = bytearray()
    cur = toks[0]
    col = 0
    while len(out) < nbytes:
        out += cur
        col += len(cur)
        if col >= 70:
            out += b"\n"
            col = 0
        else:
            out += b" "
            col += 1
        cands = nxt.get(cur)
        if not cands:
            cur = toks[rng.next() % len(toks)]
        else:
            cur = cands[rng.next() % len(cands)]
    return bytes(out[:nbytes])


def random_bytes(nbytes, seed=0x5EED000000000005):
    L = _c()
    if L is not None:
        import ctypes
        buf = ctypes.create_string_buffer(max(1, nbytes))
        L.synth_xorshift(seed & MASK, nbytes, buf)
        return buf.raw[:nbytes]
    return random_bytes_py(nbytes, seed)


def random_bytes_py(nbytes, seed=0x5EED000000000005):
    rng = XorShift(seed)
    out = bytearray()
    while len(out) < nbytes:
        out += rng.next().to_bytes(8, "little")
    return bytes(out[:nbytes])


def mixed(nbytes, seed=0x5EED000000000004):
    """Silesia-like mix: text, binary records, zero fill, hex, random"""
    rng = XorShift(seed)
    out = bytearray()
    text = markov_text(min(nbytes, 1 << 20), seed ^ 0x1111)
    while len(out) < nbytes:
        kind = rng.next() % 100
        seglen = 4096 + rng.next() % 65536
        if kind < 40:
            off = rng.next() % max(1, len(text) - seglen)
            out += text[off:off + seglen]
        elif kind < 55:
            for i in range(seglen // 8):
                out += (i * 7 + (rng.next() & 0xfff)).to_bytes(4, "little") + (i).to_bytes(4, "little")
        elif kind < 65:
            out += bytes(seglen)
        elif kind < 75:
            out += random_bytes(seglen // 2, rng.next()).hex().encode()
        else:
            out += random_bytes(seglen, rng.next())
    return bytes(out[:nbytes])


def repeated_excerpts(nbytes, seed=1, source_bytes=1 << 20):
    """Random-offset excerpts (4..68 KiB) of one 1 MiB text, concatenated: long exact repeats whose copies run across
    segment and block boundaries and get continued by extend_last_command, with more than a ring's worth of earlier
    occurrences of most hash keys."""
    text = markov_text(source_bytes, 0x1234)
    rng = XorShift(seed)
    out = bytearray()
    while len(out) < nbytes:
        seglen = 4096 + rng.next() % 65536
        off = rng.next() % (len(text) - seglen)
        out += text[off:off + seglen]
    return bytes(out[:nbytes])


def enwik_like(nbytes, seed=0x5EED000000000003):
    """the Markov text wrapped in <page> records of 2..40 KiB with [[links]]"""
    rng = XorShift(seed)
    pool = markov_text(min(nbytes, 8 << 20) + (64 << 10), seed ^ 0x3333)
    out = bytearray()
    page = 0
    at = 0
    while len(out) < nbytes:
        page += 1
        want = 2048 + rng.next() % (38 << 10)
        if at + want > len(pool):
            pool = markov_text(len(pool), rng.next())
            at = 0
        body = bytearray(pool[at:at + want])
        at += want
        # turn some words into [[links]]
        words = body.split(b" ")
        step = 23 + rng.next() % 17
        for i in range(step, len(words), step):
            if words[i] and b"\n" not in words[i]:
                words[i] = b"[[" + words[i] + b"]]"
        title = b" ".join(words[:3]).replace(b"\n", b" ")
        out += b"<page><title>" + title + b"</title><id>" + str(page).encode() + b"</id><text>" + b" ".join(words) + b"</text></page>\n"
    return bytes(out[:nbytes])


def silesia_plan(nbytes, seed=0x5EED000000000004, min_segment=1 << 20, max_segment=32 << 20, only=None):
    """the pieces of silesia_like(): list of (kind, offset, length, sub-seed); no content is generated"""
    rng = XorShift(seed)
    plan = []
    at = 0
    while at < nbytes:
        kind = rng.next() % 100
        if only is not None:
            kind = only
        seglen = min(min_segment + rng.next() % (max_segment - min_segment + 1), nbytes - at)
        sub = rng.next()
        plan.append((kind, at, seglen, sub))
        at += seglen
    return plan


def _silesia_piece(kind, seglen, sub):
    import numpy as np
    g = np.random.Generator(np.random.PCG64(sub))
    if kind < 40:
        return markov_text(seglen, sub)
    if kind < 60:
        return enwik_like(seglen, sub)
    if kind < 75:
        n = seglen // 8 + 1
        rec = np.empty((n, 2), dtype="<u4")
        rec[:, 0] = np.arange(n, dtype=np.uint32) * 3 + (sub & 0xffff)
        rec[:, 1] = (np.float32(1.0) + g.integers(0, 4096, n).astype(np.float32) / np.float32(4096.0)).view(np.uint32)
        return rec.tobytes()[:seglen]
    if kind < 85:
        return bytes(seglen)
    if kind < 95:
        raw = np.frombuffer(g.integers(0, 256, seglen // 2 + 1, dtype=np.uint8).tobytes().hex().encode(), dtype=np.uint8)
        # lines of 64 hex digits: "  0x" + digits + ",\n"
        nlines = (len(raw) + 63) // 64
        padded = np.zeros(nlines * 64, dtype=np.uint8)
        padded[:len(raw)] = raw
        lines = np.empty((nlines, 70), dtype=np.uint8)
        lines[:, 0:4] = np.frombuffer(b"  0x", dtype=np.uint8)
        lines[:, 4:68] = padded.reshape(nlines, 64)
        lines[:, 68:70] = np.frombuffer(b",\n", dtype=np.uint8)
        out = lines.tobytes()
        short = nlines * 64 - len(raw)
        if short:  # the last line holds fewer digits
            out = out[:len(out) - 2 - short] + b",\n"
        return out[:seglen]
    return g.integers(0, 256, seglen, dtype=np.uint8).tobytes()


def silesia_range(plan, lo, hi):
    """bytes [lo, hi) of the input described by `plan` (only the pieces that overlap the range are generated)"""
    out = bytearray()
    for kind, at, seglen, sub in plan:
        if at + seglen <= lo or at >= hi:
            continue
        piece = _silesia_piece(kind, seglen, sub)
        out += piece[max(lo, at) - at:min(hi, at + seglen) - at]
    return bytes(out)


def silesia_like(nbytes, seed=0x5EED000000000004, min_segment=1 << 20, max_segment=32 << 20, only=None):
    """segments of min_segment..max_segment bytes; 40 % fresh Markov text, 20 % enwik-style XML, 15 % binary
    records (LE u32 counters + floats with 12 bits of entropy), 10 % zero fill, 10 % hex / source-like, 5 % random"""
    return silesia_range(silesia_plan(nbytes, seed, min_segment, max_segment, only), 0, nbytes)


def stretches(nbytes, seed=1):
    """Stretches of random bytes, zero fill, text, binary records and hex of random lengths (1 B .. 200 KB): literal
    sprees next to compressible data, rare distance-cache hits inside the sprees (found by the fuzz sweep: at
    qualities 7-8 the cache contributes 10 candidate distances, not 4)."""
    rng = XorShift(seed)
    text = markov_text(1 << 20, seed ^ 0x77)
    rand = random_bytes(1 << 20, seed ^ 0x78)
    binary = silesia_like(1 << 20, seed ^ 0x79, only=60)
    hexa = silesia_like(1 << 20, seed ^ 0x7a, only=85)
    out = bytearray()
    while len(out) < nbytes:
        k = rng.next() % 5
        m = 1 + rng.next() % (200000 if k else 20000)
        src = [rand, None, text, binary, hexa][k]
        if src is None:
            out += bytes(m)
        else:
            o = rng.next() % max(1, len(src) - m)
            out += src[o:o + m]
    return bytes(out[:nbytes])
