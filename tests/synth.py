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


def markov_text(nbytes, seed=0x5EED000000000002):
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


def silesia_like(nbytes, seed=0x5EED000000000004, min_segment=1 << 20, max_segment=32 << 20, only=None):
    """segments of min_segment..max_segment bytes; 40 % fresh Markov text, 20 % enwik-style XML, 15 % binary
    records (LE u32 counters + floats with 12 bits of entropy), 10 % zero fill, 10 % hex / source-like, 5 % random"""
    import numpy as np
    rng = XorShift(seed)
    out = bytearray()
    while len(out) < nbytes:
        kind = rng.next() % 100
        if only is not None:
            kind = only
        seglen = min(min_segment + rng.next() % (max_segment - min_segment + 1), nbytes - len(out))
        sub = rng.next()
        g = np.random.Generator(np.random.PCG64(sub))
        if kind < 40:
            out += markov_text(seglen, sub)
        elif kind < 60:
            out += enwik_like(seglen, sub)
        elif kind < 75:
            n = seglen // 8 + 1
            rec = np.empty((n, 2), dtype="<u4")
            rec[:, 0] = np.arange(n, dtype=np.uint32) * 3 + (sub & 0xffff)
            rec[:, 1] = (np.float32(1.0) + g.integers(0, 4096, n).astype(np.float32) / np.float32(4096.0)).view(np.uint32)
            out += rec.tobytes()[:seglen]
        elif kind < 85:
            out += bytes(seglen)
        elif kind < 95:
            raw = g.integers(0, 256, seglen // 2 + 1, dtype=np.uint8).tobytes().hex().encode()
            lines = bytearray()
            for i in range(0, len(raw), 64):
                lines += b"  0x" + raw[i:i + 64] + b",\n"
            out += lines[:seglen]
        else:
            out += g.integers(0, 256, seglen, dtype=np.uint8).tobytes()
    return bytes(out[:nbytes])


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
