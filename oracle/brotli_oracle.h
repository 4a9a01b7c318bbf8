/* oracle/brotli_oracle.h -- CPU restatement of rust-brotli's encoder path (qualities 0..11, incl. "9.5").
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (rust-brotli_amd/, include/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Parity status: the reference (Rust) cannot be built in this environment (no cargo/rustc).  The oracle is pinned by
 * every exact number the reference's own tests hold for this path (tests/test_oracle.py):
 *   encoder_compress(q9, lgwin16, alice29.txt) == 51737 bytes                    (src/enc/encode.rs:3073-3099)
 *   random_then_unicode, q10 + q9_5 / q11 + q9_5  == 130036 / 129715 bytes       (src/bin/integration_tests.rs:397-428)
 *   alice29.txt, q10 / q11 (H10 + Zopfli)         == 47488 / 46493 bytes         (src/bin/integration_tests.rs:401-449)
 * by the compress_multi size bounds of src/bin/test_threading.rs:91-110, by byte identity with Google's libbrotlienc
 * 1.0.9 at qualities 5..8 modulo four documented source differences (tests/test_oracle_vs_libbrotlienc.py), and by
 * round trips through an independent decoder (libbrotlidec).
 * Files: orc_lz77.c (hashers H5/H6/H9, greedy parse), orc_zopfli.c (H10, Zopfli), orc_static_dict.c (all-matches
 * dictionary search), orc_metablock.c (greedy meta-block builder, Huffman, bit stream), orc_hq_metablock.c (quality >= 10
 * block splitter + clustering), orc_encode.c (stream state machine), orc_multi.c (compress_multi + BroCatli).
 * Qualities 2..4 (BasicHasher H2/H3/H4/H54, store_meta_block_fast / _trivial) are byte-identical to libbrotlienc 1.0.9
 * modulo two more documented source differences, qualities 0 and 1 (orc_fragment.c: compress_fragment,
 * compress_fragment_two_pass) modulo one each; their catable forms (the quality 0 / 1 branch of encode_data, encode.rs:2335-2389) exist in
 * the Rust sources only.
 */
#ifndef BROTLI_ORACLE_H_
#define BROTLI_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* parameter ids: reference src/enc/parameters.rs:3-33 */
enum {
  ORC_PARAM_MODE = 0,
  ORC_PARAM_QUALITY = 1,
  ORC_PARAM_LGWIN = 2,
  ORC_PARAM_LGBLOCK = 3,
  ORC_PARAM_DISABLE_LITERAL_CONTEXT_MODELING = 4,
  ORC_PARAM_SIZE_HINT = 5,
  ORC_PARAM_LARGE_WINDOW = 6,
  ORC_PARAM_Q9_5 = 150,
  ORC_PARAM_LITERAL_BYTE_SCORE = 154,
  ORC_PARAM_CATABLE = 167,
  ORC_PARAM_APPENDABLE = 168,
  ORC_PARAM_MAGIC_NUMBER = 169,
  ORC_PARAM_NO_DICTIONARY = 170,
  ORC_PARAM_FAVOR_EFFICIENCY = 171,
  ORC_PARAM_BYTE_ALIGN = 172,
  ORC_PARAM_BARE_STREAM = 173
};

enum { ORC_OP_PROCESS = 0, ORC_OP_FLUSH = 1, ORC_OP_FINISH = 2, ORC_OP_EMIT_METADATA = 3 };

typedef struct OrcEncoder OrcEncoder;

/* one LZ77 command: reference src/enc/command.rs:12-21 (16 bytes) */
typedef struct OrcCommand {
  uint32_t insert_len_;
  uint32_t copy_len_;
  uint32_t dist_extra_;
  uint16_t cmd_prefix_;
  uint16_t dist_prefix_;
} OrcCommand;

/* Instrumentation counters (roofline inputs) */
typedef struct OrcStats {
  uint64_t positions_searched; /* S: FindLongestMatch calls */
  uint64_t positions_stored;   /* hash-table inserts */
  uint64_t commands;           /* K */
  uint64_t literals;           /* L */
  uint64_t metablocks;
  uint64_t uncompressed_metablocks;
  uint64_t dict_lookups, dict_matches;
} OrcStats;

/* meta-block trace callback: called once per emitted compressed/uncompressed meta-block with the
   final command list (after the trailing insert-only command was added).  kind: 0 compressed,
   1 uncompressed (should_compress false), 2 uncompressed (size fallback). */
typedef void (*OrcMetablockTrace)(void* opaque, int kind, uint64_t start_pos, size_t bytes,
                                  const OrcCommand* cmds, size_t n_cmds,
                                  const int32_t dist_cache_after[4]);

OrcEncoder* orc_encoder_create(void);
void orc_encoder_destroy(OrcEncoder* s);
int orc_encoder_set_parameter(OrcEncoder* s, int param, uint32_t value);
void orc_encoder_set_trace(OrcEncoder* s, OrcMetablockTrace cb, void* opaque);
const OrcStats* orc_encoder_stats(const OrcEncoder* s);
/* reference encode.rs:1196-1270 */
void orc_encoder_set_custom_dictionary(OrcEncoder* s, size_t size, const uint8_t* dict,
                                       int is_multithreading_file_continue);
/* reference encode.rs:2873-2995 */
int orc_encoder_compress_stream(OrcEncoder* s, int op, size_t* available_in, const uint8_t** next_in,
                                size_t* available_out, uint8_t** next_out, size_t* total_out);
int orc_encoder_is_finished(const OrcEncoder* s);
int orc_encoder_has_more_output(const OrcEncoder* s);
const uint8_t* orc_encoder_take_output(OrcEncoder* s, size_t* size);

size_t orc_max_compressed_size(size_t input_size);
size_t orc_max_compressed_size_multi(size_t input_size, size_t num_threads);

/* one-shot: reference encode.rs:1436-1538 (encoder_compress) */
int orc_encoder_compress(int quality, int lgwin, int mode, size_t input_size, const uint8_t* input,
                         size_t* encoded_size, uint8_t* encoded, OrcStats* stats_out);

/* "CompressorWriter" feeding pattern: size_hint left 0, input offered in `chunk`-byte writes,
   then FINISH (reference src/enc/writer.rs:183-313).  chunk==0 means one write of everything. */
int orc_writer_compress(int quality, int lgwin, size_t chunk, size_t input_size, const uint8_t* input,
                        size_t* encoded_size, uint8_t* encoded, OrcStats* stats_out,
                        OrcMetablockTrace cb, void* opaque);

/* BrotliCompressCustomIo feeding pattern (src/enc/mod.rs:225-345): parameters set, PROCESS per `chunk` bytes read, FINISH
   without input at the end.  This is how the reference's integration tests (src/bin/integration_tests.rs:310-338) feed
   the encoder, so their exact size pins apply to it. */
int orc_reader_compress(const int* param_keys, const uint32_t* param_values, size_t num_params, size_t chunk,
                        size_t input_size, const uint8_t* input, size_t* encoded_size, uint8_t* encoded,
                        OrcStats* stats_out, OrcMetablockTrace cb, void* opaque);

/* BrotliCompressCustomIo feeding pattern (src/enc/mod.rs:225-345): parameters set, PROCESS per `chunk` bytes read, FINISH
   without input at the end.  This is how the reference's integration tests (src/bin/integration_tests.rs:310-338) feed
   the encoder, so their exact size pins apply to it. */
int orc_reader_compress(const int* param_keys, const uint32_t* param_values, size_t num_params, size_t chunk,
                        size_t input_size, const uint8_t* input, size_t* encoded_size, uint8_t* encoded,
                        OrcStats* stats_out, OrcMetablockTrace cb, void* opaque);

/* compress_multi: reference src/enc/threading/mod.rs:333-661 + src/concat/mod.rs.  Runs the
   shards sequentially on one thread (results are independent of thread scheduling). */
int orc_compress_multi(const int* param_keys, const uint32_t* param_values, size_t num_params,
                       size_t input_size, const uint8_t* input, size_t* encoded_size,
                       uint8_t* encoded, size_t num_threads);

/* glibc-compatible log2f restated (used to check the device implementation) */
float orc_log2f_restated(float x);
uint32_t orc_log2f_check_all(void);
/* f32 BitsEntropy, reference src/enc/bit_cost.rs:13-42 */
float orc_bits_entropy(const uint32_t* population, size_t size);

#ifdef __cplusplus
}
#endif
/* test hook (orc_lz77.c): map[ix] |= 1 for every position inserted into the hash table, |= 2 for every searched one */
void orc_set_debug_store_map(uint8_t* map, size_t size);

#endif
