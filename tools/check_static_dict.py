#!/usr/bin/env python3
"""Cross-check oracle/orc_static_dict.c against the reference text of BrotliFindAllStaticDictionaryMatches
(src/enc/static_dict.rs:309-1300): the ORDER of character tests and of the (transform id, length offset) pairs of every
AddMatch call must agree.  Build container only (reads /root/reference); a development aid, not a test."""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rs = open("/root/reference/src/enc/static_dict.rs").read()
rs = rs[rs.index("pub fn BrotliFindAllStaticDictionaryMatches"):rs.index("#[cfg(test)]")]
rs = re.sub(r"//[^\n]*", "", rs)
c = open(os.path.join(ROOT, "oracle", "orc_static_dict.c")).read()
c = c[c.index("int orc_find_all_static_dictionary_matches"):]
c = re.sub(r"/\*.*?\*/", "", c, flags=re.S)


def unesc(s):
    return {"\\n": "\n", "\\t": "\t", "\\'": "'", '\\"': '"'}.get(s, s)


rs_chars = [unesc(m) for m in re.findall(r"b'((?:\\.|[^'])+)'", rs)] 
rs_chars += []
c_chars = [unesc(m) for m in re.findall(r"'((?:\\.|[^'])+)'", c)]
# the Rust spells the two UTF-8 bytes of U+00A0 as 0xc2i32 / 0xa0i32 and tests b' ' once more for `is_space`
print("char tests: rust %d, c %d" % (len(rs_chars), len(c_chars)))


def rust_calls(src):
    out = []
    for m in re.finditer(r"AddMatch\(", src):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        body = src[m.end():i - 1]
        nums = [int(x) for x in re.findall(r"(?<![\w\[])(\d+)(?:usize|i32)?", body)]
        out.append(nums)
    return out


def c_calls(src):
    out = []
    for m in re.finditer(r"(ADD|add_match)\(", src):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        body = src[m.end():i - 1]
        nums = [int(x) for x in re.findall(r"(?<![\w\[])(\d+)", body)]
        out.append(nums)
    return out


r, k = rust_calls(rs), c_calls(c)
print("AddMatch calls: rust %d, c %d" % (len(r), len(k)))
bad = 0
for i, (a, b) in enumerate(zip(r, k)):
    # ADD(t, 0) in C is AddMatch(.., l, l) in Rust and ADD(1, 1) is AddMatch(id + n, l + 1, l): compare as sets without 0
    if set(a) - {0} != set(b) - {0}:
        print("call %d differs: rust %s c %s" % (i, a, b))
        bad += 1
ok_chars = rs_chars == c_chars
if not ok_chars:
    for i, (a, b) in enumerate(zip(rs_chars, c_chars)):
        if a != b:
            print("first char difference at %d: rust %r c %r (context rust %r c %r)" % (i, a, b, rs_chars[i-3:i+3], c_chars[i-3:i+3]))
            break
print("OK" if not bad and ok_chars and len(r) == len(k) else "MISMATCH")
