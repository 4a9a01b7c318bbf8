"""A brotli stream PARSER (RFC 7932) in plain Python: decodes a stream into its meta-block headers, prefix codes, block
splits, context maps and the insert&copy command list, reconstructing the bytes on the way.

Test infrastructure: it lets the tests say WHERE two encoders' streams for the same input part ways (first differing
command / header field) instead of only "the bytes differ" -- used to compare the CPU oracle with Google's
libbrotlienc.so.1 (tests/test_oracle_vs_libbrotlienc.py).  Slow (pure Python): meant for inputs of a few hundred KB.
Static dictionary words and transforms come from the system libbrotlicommon.so.1."""
import ctypes

_INS_BASE = [0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594]
_INS_EXTRA = [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24]
_COPY_BASE = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118]
_COPY_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24]
_BLOCK_LEN = [(1, 2), (5, 2), (9, 2), (13, 2), (17, 3), (25, 3), (33, 3), (41, 3), (49, 4), (65, 4), (81, 4), (97, 4),
              (113, 5), (145, 5), (177, 5), (209, 5), (241, 6), (305, 6), (369, 7), (497, 8), (753, 9), (1265, 10),
              (2289, 11), (4337, 12), (8433, 13), (16625, 24)]
_CL_ORDER = [1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15]
_CL_PREFIX_LEN = [2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4]
_CL_PREFIX_VAL = [0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5]


class _Tables:
    _inst = None

    def __init__(self):
        import os
        import re
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        src = open(os.path.join(root, "tables", "brotli_tables.h")).read()

        def arr(name):
            m = re.search(name + r"\[\d+\]\s*=\s*\{([^}]*)\}", src)
            return [int(x, 0) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
        self.utf8 = arr("kBrotliUTF8ContextLookup")       # [0..255] = Lut0, [256..511] = Lut1
        self.signed = arr("kBrotliSigned3BitContextLookup")  # Lut2
        self.common = ctypes.CDLL("libbrotlicommon.so.1")
        self.common.BrotliGetDictionary.restype = ctypes.c_void_p
        self.common.BrotliGetTransforms.restype = ctypes.c_void_p
        self.common.BrotliTransformDictionaryWord.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int,
                                                              ctypes.c_void_p, ctypes.c_int]

        class Dict(ctypes.Structure):
            _fields_ = [("size_bits_by_length", ctypes.c_uint8 * 32), ("offsets_by_length", ctypes.c_uint32 * 32),
                        ("data_size", ctypes.c_size_t), ("data", ctypes.c_void_p)]
        d = Dict.from_address(self.common.BrotliGetDictionary())
        self.size_bits = list(d.size_bits_by_length)
        self.offsets = list(d.offsets_by_length)
        self.dict_data_addr = d.data
        self.transforms = self.common.BrotliGetTransforms()

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = _Tables()
        return cls._inst

    def word(self, length, index, transform):
        dst = ctypes.create_string_buffer(64)
        n = self.common.BrotliTransformDictionaryWord(dst, self.dict_data_addr + self.offsets[length] + length * index,
                                                      length, self.transforms, transform)
        return dst.raw[:n]


class _Bits:
    """LSB-first bit reader over a bytes object"""

    def __init__(self, data):
        self.data = data
        self.pos = 0
        self.nbits = len(data) * 8

    def peek(self, n):
        byte = self.pos >> 3
        chunk = int.from_bytes(self.data[byte:byte + 5], "little")
        return (chunk >> (self.pos & 7)) & ((1 << n) - 1)

    def read(self, n):
        if n == 0:
            return 0
        if self.pos + n > self.nbits:
            raise ValueError("stream truncated")
        if n <= 32:
            r = self.peek(n)
        else:
            byte = self.pos >> 3
            chunk = int.from_bytes(self.data[byte:byte + (n >> 3) + 2], "little")
            r = (chunk >> (self.pos & 7)) & ((1 << n) - 1)
        self.pos += n
        return r

    def align(self):
        pad = (-self.pos) % 8
        if self.read(pad) != 0:
            raise ValueError("non-zero padding")


class _Code:
    """canonical prefix code given code lengths; decode reads bit by bit (first bit = most significant code bit)"""

    def __init__(self, lengths):
        self.lengths = lengths
        nz = [(l, s) for s, l in enumerate(lengths) if l]
        self.single = None
        if len(nz) == 0:
            raise ValueError("empty prefix code")
        if len(nz) == 1:
            self.single = nz[0][1]
            return
        nz.sort()
        self.table = {}
        code = 0
        prev = nz[0][0]
        for l, s in nz:
            code <<= (l - prev)
            prev = l
            self.table[(l, code)] = s
            code += 1
        self.maxlen = nz[-1][0]

    def decode(self, br):
        if self.single is not None:
            return self.single
        code = 0
        for l in range(1, self.maxlen + 1):
            code = (code << 1) | br.read(1)
            s = self.table.get((l, code))
            if s is not None:
                return s
        raise ValueError("invalid prefix code")


def _read_prefix_code(br, alphabet):
    hskip = br.read(2)
    if hskip == 1:
        nsym = br.read(2) + 1
        bits = max(1, (alphabet - 1).bit_length())
        syms = [br.read(bits) for _ in range(nsym)]
        if len(set(syms)) != nsym or any(s >= alphabet for s in syms):
            raise ValueError("bad simple prefix code")
        lengths = [0] * alphabet
        if nsym == 1:
            lengths[syms[0]] = 1  # zero bits in the stream; _Code treats a lone symbol as length 0
            return dict(kind="simple", symbols=syms, lengths=lengths), _Code(lengths)
        if nsym == 2:
            for s in syms:
                lengths[s] = 1
        elif nsym == 3:
            lengths[syms[0]] = 1
            lengths[syms[1]] = lengths[syms[2]] = 2
        else:
            if br.read(1) == 0:
                for s in syms:
                    lengths[s] = 2
            else:
                lengths[syms[0]] = 1
                lengths[syms[1]] = 2
                lengths[syms[2]] = lengths[syms[3]] = 3
        return dict(kind="simple", symbols=syms, lengths=lengths), _Code(lengths)
    cl = [0] * 18
    space = 32
    num = 0
    for i in range(hskip, 18):
        ix = br.peek(4)
        br.read(_CL_PREFIX_LEN[ix])
        v = _CL_PREFIX_VAL[ix]
        cl[_CL_ORDER[i]] = v
        if v:
            space -= 32 >> v
            num += 1
            if space <= 0:
                break
    if not (num == 1 or space == 0):
        raise ValueError("bad code length code")
    clcode = _Code(cl)
    lengths = [0] * alphabet
    i = 0
    prev = 8
    rep = 0
    rep_len = 0
    space = 32768
    while i < alphabet and space > 0:
        s = clcode.decode(br)
        if s < 16:
            lengths[i] = s
            i += 1
            rep = 0
            if s:
                prev = s
                space -= 32768 >> s
        else:
            extra = 2 if s == 16 else 3
            new_len = prev if s == 16 else 0
            if rep_len != new_len:
                rep = 0
                rep_len = new_len
            old = rep
            if rep > 0:
                rep = (rep - 2) << extra
            rep += br.read(extra) + 3
            delta = rep - old
            if i + delta > alphabet:
                raise ValueError("repeat runs past the alphabet")
            for _ in range(delta):
                lengths[i] = rep_len
                i += 1
            if rep_len:
                space -= delta * (32768 >> rep_len)
    if space != 0:
        raise ValueError("code lengths do not fill the code space (%d left)" % space)
    return dict(kind="complex", hskip=hskip, code_length_lengths=cl, lengths=lengths), _Code(lengths)


def _read_varlen_1_256(br):
    if br.read(1) == 0:
        return 1
    n = br.read(3)
    return (1 << n) + 1 + br.read(n)


def _read_block_len(code, br):
    s = code.decode(br)
    base, extra = _BLOCK_LEN[s]
    return base + br.read(extra)


def _read_context_map(br, size, ntrees):
    rlemax = 0
    if br.read(1):
        rlemax = br.read(4) + 1
    desc, code = _read_prefix_code(br, ntrees + rlemax)
    cmap = []
    while len(cmap) < size:
        s = code.decode(br)
        if s == 0:
            cmap.append(0)
        elif s <= rlemax:
            n = (1 << s) + br.read(s)
            if len(cmap) + n > size:
                raise ValueError("context map run too long")
            cmap.extend([0] * n)
        else:
            cmap.append(s - rlemax)
    imtf = br.read(1)
    if imtf:
        mtf = list(range(256))
        for i, v in enumerate(cmap):
            x = mtf[v]
            cmap[i] = x
            del mtf[v]
            mtf.insert(0, x)
    return dict(rlemax=rlemax, imtf=imtf, code=desc), cmap


def parse(stream, max_metablocks=None):
    """-> dict(wbits, metablocks=[...], output=bytes).  Every meta-block dict holds the header fields and `cmds`:
    a list of (insert_len, copy_len, distance_symbol or None (implicit last distance), distance, out_pos)."""
    T = _Tables.get()
    br = _Bits(stream)
    if br.read(1) == 0:
        wbits = 16
    else:
        n = br.read(3)
        if n:
            wbits = 17 + n
        else:
            m = br.read(3)
            if m == 0:
                wbits = 17
            elif m == 1:
                raise ValueError("large window streams are not supported")
            else:
                wbits = 8 + m
    max_backward = (1 << wbits) - 16
    out = bytearray()
    dist_rb = [16, 15, 11, 4]  # last distances, dist_rb[-1] is the most recent ... kept as list with idx
    rb_idx = 0
    metablocks = []
    while True:
        mb = dict(bit_offset=br.pos, out_pos=len(out))
        is_last = br.read(1)
        mb["is_last"] = is_last
        if is_last and br.read(1):
            mb["empty"] = True
            metablocks.append(mb)
            break
        nib = br.read(2)
        if nib == 3:
            if br.read(1):
                raise ValueError("reserved bit set")
            nbytes = br.read(2)
            skip = 0
            if nbytes:
                skip = br.read(8 * nbytes) + 1
                if nbytes > 1 and (skip - 1) >> (8 * (nbytes - 1)) == 0:
                    raise ValueError("non-minimal metadata length")
            br.align()
            mb["metadata"] = bytes((br.read(8) for _ in range(skip)))
            metablocks.append(mb)
            if is_last:
                break
            continue
        mlen = br.read(4 * (nib + 4)) + 1
        mb["mlen"] = mlen
        if not is_last and br.read(1):
            br.align()
            mb["uncompressed"] = True
            start = br.pos // 8
            out += stream[start:start + mlen]
            br.pos += 8 * mlen
            metablocks.append(mb)
            continue
        mb["uncompressed"] = False
        nbl = []
        bt_code = []
        bl_code = []
        blen = []
        for _ in range(3):
            n = _read_varlen_1_256(br)
            nbl.append(n)
            if n >= 2:
                d1, c1 = _read_prefix_code(br, n + 2)
                d2, c2 = _read_prefix_code(br, 26)
                bt_code.append(c1)
                bl_code.append(c2)
                blen.append(_read_block_len(c2, br))
            else:
                bt_code.append(None)
                bl_code.append(None)
                blen.append(1 << 28)
        mb["nbltypes"] = tuple(nbl)
        mb["first_block_len"] = tuple(b if b < (1 << 28) else None for b in blen)
        npostfix = br.read(2)
        ndirect = br.read(4) << npostfix
        mb["npostfix"], mb["ndirect"] = npostfix, ndirect
        mb["context_modes"] = [br.read(2) for _ in range(nbl[0])]
        ntrees_l = _read_varlen_1_256(br)
        mb["ntrees_l"] = ntrees_l
        if ntrees_l >= 2:
            mb["cmap_l_desc"], cmap_l = _read_context_map(br, 64 * nbl[0], ntrees_l)
        else:
            cmap_l = [0] * (64 * nbl[0])
        ntrees_d = _read_varlen_1_256(br)
        mb["ntrees_d"] = ntrees_d
        if ntrees_d >= 2:
            mb["cmap_d_desc"], cmap_d = _read_context_map(br, 4 * nbl[2], ntrees_d)
        else:
            cmap_d = [0] * (4 * nbl[2])
        mb["cmap_l"], mb["cmap_d"] = cmap_l, cmap_d
        lit_codes, cmd_codes, dist_codes = [], [], []
        mb["lit_lengths"], mb["cmd_lengths"], mb["dist_lengths"] = [], [], []
        for _ in range(ntrees_l):
            d, c = _read_prefix_code(br, 256)
            lit_codes.append(c)
            mb["lit_lengths"].append(d["lengths"])
        for _ in range(nbl[1]):
            d, c = _read_prefix_code(br, 704)
            cmd_codes.append(c)
            mb["cmd_lengths"].append(d["lengths"])
        dist_alphabet = 16 + ndirect + (48 << npostfix)
        for _ in range(ntrees_d):
            d, c = _read_prefix_code(br, dist_alphabet)
            dist_codes.append(c)
            mb["dist_lengths"].append(d["lengths"])
        mb["data_bit_offset"] = br.pos
        btype = [0, 0, 0]
        prev_btype = [1, 1, 1]
        cmds = []
        splits = ([], [], [])  # (type, length) per block and category
        for cat in range(3):
            splits[cat].append([0, blen[cat] if blen[cat] < (1 << 28) else None])

        def switch(cat):
            code = bt_code[cat].decode(br)
            n = nbl[cat]
            if code == 0:
                t = prev_btype[cat]
            elif code == 1:
                t = (btype[cat] + 1) % n
            else:
                t = code - 2
            prev_btype[cat] = btype[cat]
            btype[cat] = t
            blen[cat] = _read_block_len(bl_code[cat], br)
            splits[cat].append([t, blen[cat]])

        produced = 0
        postfix_mask = (1 << npostfix) - 1
        while produced < mlen:
            if blen[1] == 0:
                switch(1)
            blen[1] -= 1
            sym = cmd_codes[btype[1]].decode(br)
            ridx = sym >> 6
            implicit = ridx < 2
            if ridx >= 2:
                ridx -= 2
            icode = (((0x29850 >> (ridx * 2)) & 3) << 3) | ((sym >> 3) & 7)
            ccode = (((0x26244 >> (ridx * 2)) & 3) << 3) | (sym & 7)
            insert_len = _INS_BASE[icode] + br.read(_INS_EXTRA[icode])
            copy_len = _COPY_BASE[ccode] + br.read(_COPY_EXTRA[ccode])
            cmd_pos = len(out)
            for _ in range(insert_len):
                if blen[0] == 0:
                    switch(0)
                blen[0] -= 1
                p1 = out[-1] if len(out) >= 1 else 0
                p2 = out[-2] if len(out) >= 2 else 0
                mode = mb["context_modes"][btype[0]]
                if mode == 0:
                    ctx = p1 & 0x3f
                elif mode == 1:
                    ctx = p1 >> 2
                elif mode == 2:
                    ctx = T.utf8[p1] | T.utf8[256 + p2]
                else:
                    ctx = (T.signed[p1] << 3) | T.signed[p2]
                out.append(lit_codes[cmap_l[64 * btype[0] + ctx]].decode(br))
            produced += insert_len
            if produced >= mlen:
                cmds.append((insert_len, 0, None, 0, cmd_pos))
                break
            if implicit:
                dsym = None
                distance = dist_rb[(rb_idx - 1) & 3]
            else:
                if blen[2] == 0:
                    switch(2)
                blen[2] -= 1
                dctx = 3 if copy_len > 4 else copy_len - 2
                dsym = dist_codes[cmap_d[4 * btype[2] + dctx]].decode(br)
                if dsym < 16:
                    idx = [1, 2, 3, 4, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2][dsym]
                    off = [0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3][dsym]
                    distance = dist_rb[(rb_idx - idx) & 3] + off
                    if distance <= 0:
                        raise ValueError("non-positive distance")
                elif dsym < 16 + ndirect:
                    distance = dsym - 15
                else:
                    v = dsym - ndirect - 16
                    nbits = 1 + (v >> (npostfix + 1))
                    hcode = v >> npostfix
                    lcode = v & postfix_mask
                    offset = ((2 + (hcode & 1)) << nbits) - 4
                    distance = ((offset + br.read(nbits)) << npostfix) + lcode + ndirect + 1
            max_distance = min(len(out), max_backward)
            if distance <= max_distance:
                if dsym != 0 and dsym is not None:
                    dist_rb[rb_idx & 3] = distance
                    rb_idx += 1
                start = len(out) - distance
                for i in range(copy_len):
                    out.append(out[start + i])
                produced += copy_len
            else:
                if not 4 <= copy_len <= 24:
                    raise ValueError("bad dictionary word length")
                wid = distance - max_distance - 1
                shift = T.size_bits[copy_len]
                word = T.word(copy_len, wid & ((1 << shift) - 1), wid >> shift)
                out += word
                produced += len(word)
            cmds.append((insert_len, copy_len, dsym, distance, cmd_pos))
        if produced != mlen:
            raise ValueError("meta-block length mismatch")
        mb["cmds"] = cmds
        mb["splits"] = splits
        metablocks.append(mb)
        if is_last or (max_metablocks and len(metablocks) >= max_metablocks):
            break
    return dict(wbits=wbits, metablocks=metablocks, output=bytes(out))


def first_difference(a, b):
    """where two parses of streams for the same input part ways: ('identical',) or (metablock index, what, detail)"""
    for i, (x, y) in enumerate(zip(a["metablocks"], b["metablocks"])):
        for key in ("is_last", "mlen", "uncompressed", "nbltypes", "npostfix", "ndirect", "context_modes", "ntrees_l",
                    "ntrees_d", "cmap_l", "cmap_d"):
            if x.get(key) != y.get(key):
                return (i, key, (x.get(key), y.get(key)))
        if x.get("uncompressed") or "cmds" not in x:
            continue
        for j, (c, d) in enumerate(zip(x["cmds"], y["cmds"])):
            if c != d:
                return (i, "command", (j, c, d))
        if len(x["cmds"]) != len(y["cmds"]):
            return (i, "command count", (len(x["cmds"]), len(y["cmds"])))
        if x["splits"] != y["splits"]:
            for cat, name in enumerate(("literal", "command", "distance")):
                if x["splits"][cat] != y["splits"][cat]:
                    return (i, name + " block split", (len(x["splits"][cat]), len(y["splits"][cat])))
        for key in ("lit_lengths", "cmd_lengths", "dist_lengths"):
            if x[key] != y[key]:
                return (i, key, None)
    if len(a["metablocks"]) != len(b["metablocks"]):
        return (min(len(a["metablocks"]), len(b["metablocks"])), "metablock count", (len(a["metablocks"]), len(b["metablocks"])))
    return ("identical",)
