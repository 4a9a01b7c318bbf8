"""The BASELINE.json configurations at their stated sizes: seeded inputs + encoder settings.
tools/freeze_large_hashes.py runs the CPU oracle over them once; tests/test_large_gpu.py compares the HIP path with the
frozen hashes."""
import synth

CASES = {
    # configs[1]: 64 MiB English-like text, q5 / lgwin 22 (H6)
    "c2_text_64MiB_q5": dict(make=lambda: synth.markov_text(64 << 20), quality=5, lgwin=22),
    # the same generator at 256 MiB: 4 KiB-segment regime of the speculative parse
    "text_256MiB_q5": dict(make=lambda: synth.markov_text(256 << 20), quality=5, lgwin=22),
    # configs[2]: 256 MiB enwik-style corpus, q9 / lgwin 22 (H9 greedy)
    "c3_enwik_256MiB_q9": dict(make=lambda: synth.enwik_like(256 << 20), quality=9, lgwin=22),
    # configs[4]: 1 GiB xorshift64* (incompressible), q5
    "c5_xorshift_1GiB_q5": dict(make=lambda: synth.random_bytes(1 << 30, 0x5EED000000000005), quality=5, lgwin=22),
    # zero fill, 1 GiB (north_star's fifth distribution)
    "zero_1GiB_q5": dict(make=lambda: bytes(1 << 30), quality=5, lgwin=22),
    # configs[3] at a quarter of its size: Silesia-like mix, BrotliEncoderCompressMulti with 8 shards of 128 MiB, the caller
    # stating the size of the input like c/brotli.c does (BROTLI_PARAM_SIZE_HINT; the reference's own multi tests pass it too,
    # src/ffi/multicompress/test.rs:40): every shard gets an H6 hasher (encode.rs:863-893).
    # `seeds`: the reference encoder FAILS on some shardings of this kind of data (a match cut to one byte at the end of the
    # custom dictionary; the oracle raises ReferencePanics): the first seed it accepts is frozen.
    "c4_silesia_1GiB_multi8_hinted": dict(make=lambda seed: synth.silesia_like(1 << 30, seed), quality=5, lgwin=22, shards=8, hint=1 << 30,
                                          seeds=[0x5EED000000000004 + i for i in range(16)]),
    # The same call WITHOUT a size hint: the shards' hashers are chosen in set_custom_dictionary before any size is known --
    # H5, whose StoreRangeOptBatch files masked ring entries once a shard (with its prefix) has passed the 8 MiB ring buffer
    # (mod.rs:1163-1232).  Such a shard is parsed by one live chain (lz77_live.h), at the speed of one wavefront: 8 shards of
    # 16 MiB keep the test short.
    "c4_silesia_128MiB_multi8_h5": dict(make=lambda seed: synth.silesia_like(128 << 20, seed, min_segment=256 << 10, max_segment=8 << 20),
                                        quality=5, lgwin=22, shards=8, seeds=[0x5EED000000000004 + i for i in range(16)]),
    # (configs[3] itself -- 4 GiB, 8 shards of 512 MiB -- cannot be frozen: the reference FAILS on it, on every one of the
    # 13 seeds tried, also when no shard boundary lies in periodic data.  fix_unbroken_len (mod.rs:42-54) is applied to
    # ring-buffer indices, so the "no match across the custom-dictionary end" rule returns with every revolution of the
    # 8 MiB ring -- 64 times per 512 MiB shard -- and each time there is a chance that a last-distance match is cut to ONE
    # byte and still wins (score 2070 > 2020), which GetCopyLengthCode cannot encode (command.rs:91-93).)
}


# bench.py --gpus N (weak scaling): one stream of N x 64 MiB text, compress_multi with N shards and the size of the stream as
# BROTLI_PARAM_SIZE_HINT (the N = 1 step is BrotliEncoderCompress, which sets it too)
for _n in (2, 4, 8):
    CASES["text_%dx64MiB_multi%d_hinted" % (_n, _n)] = dict(make=(lambda n=_n: synth.markov_text(n * (64 << 20))), quality=5, lgwin=22, shards=_n,
                                                           hint=_n * (64 << 20), bench_only=True)


# a single stream longer than the reference's 32-bit position wrap (hasher reset at 3 GiB), fed in 4 MiB writes with
# PROCESS, then FINISH without input (BrotliCompressCustomIo's pattern), BROTLI_PARAM_SIZE_HINT = 1 GiB (H6: without a hint
# the stream runs under H5 with masked ring entries, i.e. at the speed of one live chain -- tests/test_streaming.py has
# that case at 40 MiB): bounded-memory streaming at full scale
CASES["stream_4GiB_q5_w22_hinted"] = dict(make=lambda: synth.markov_text(4 << 30, 0x5EED00000000000A), quality=5, lgwin=22, writer_chunk=4 << 20,
                                          hint=1 << 30, bench_only=True)


# BrotliEncoderCompress on more than 2 GiB (round 2 refused it): the call goes through the stream state machine in 64 MiB
# batches (cabi.cpp CompressOneShotStreamed) and must produce the one-shot stream of the reference, size hint = (u32) size
CASES["oneshot_2304MiB_q5_w22"] = dict(make=lambda: synth.markov_text(9 << 28, 0x5EED00000000000B), quality=5, lgwin=22, bench_only=True)


def make_input(name, frozen=None):
    """the input of case `name`; multi-shard cases use the seed recorded in tests/golden/large_hashes.json"""
    case = CASES[name]
    if "seeds" in case:
        return case["make"](int(frozen[name]["seed"], 16))
    return case["make"]()
