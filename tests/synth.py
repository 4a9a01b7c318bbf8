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
