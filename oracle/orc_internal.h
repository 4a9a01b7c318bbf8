/* oracle/orc_internal.h -- shared declarations of the CPU oracle (test infrastructure only). */
#ifndef ORC_INTERNAL_H_
#define ORC_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "brotli_oracle.h"

typedef OrcCommand Command;

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ---- tables (tables/brotli_tables.h, generated data) ---- */
extern const float* orc_logs_16(void);
extern const float* orc_logs_8(void);
extern const uint16_t* orc_static_dictionary_hash(void);
extern const uint8_t* orc_dictionary_data(void);
extern const uint32_t* orc_dictionary_offsets_by_length(void);
extern const uint8_t* orc_dictionary_size_bits_by_length(void);
extern const uint32_t* orc_ins_base(void);
extern const uint32_t* orc_ins_extra(void);
extern const uint32_t* orc_copy_base(void);
extern const uint32_t* orc_copy_extra(void);
extern const uint8_t* orc_utf8_context_lookup(void);
extern const uint8_t* orc_signed3_context_lookup(void);

/* ---- params: reference src/enc/backward_references/mod.rs:56-131 ---- */
typedef struct {
  int type_;
  int bucket_bits;
  int block_bits;
  int hash_len;
  int num_last_distances_to_check;
  int literal_byte_score;
} HasherParams;

typedef struct {
  uint32_t distance_postfix_bits;
  uint32_t num_direct_distance_codes;
  uint32_t alphabet_size;
  size_t max_distance;
} DistanceParams;

typedef struct {
  DistanceParams dist;
  int mode;
  int quality;
  int q9_5;
  int lgwin;
  int lgblock;
  size_t size_hint;
  int disable_literal_context_modeling;
  HasherParams hasher;
  int large_window;
  int byte_align;
  int bare_stream;
  int catable;
  int use_dictionary;
  int appendable;
  int magic_number;
  int favor_cpu_efficiency;
} EncoderParams;

/* ---- hasher (H5/H5q5/H6 = AdvHasher, H9): mod.rs:598-917, 919-1813 ---- */
typedef struct {
  int kind; /* 0 uninit, 5 (H5 family, 32-bit hash of 4 bytes), 6 (H6), 9 (H9) */
  HasherParams params;
  int is_prepared_;
  size_t dict_num_lookups;
  size_t dict_num_matches;
  uint32_t literal_byte_score; /* H9Opts */
  /* geometry */
  int bucket_bits, block_bits;
  uint32_t block_size, block_mask;
  uint64_t hash_mask; /* H6 */
  size_t bucket_count;
  uint16_t* num;
  uint32_t* buckets;
  /* BasicHasher (kinds 2, 3, 4, 54): mod.rs:237-596 */
  int sweep, basic_use_dictionary, basic_hash_len;
  /* H10 (kind 10): hash_to_binary_tree.rs:108-122 */
  size_t window_mask_;
  uint32_t invalid_pos_;
  size_t ringbuffer_break;
  uint32_t* forest;
} Hasher;

typedef struct {
  size_t len;
  size_t len_x_code;
  size_t distance;
  uint64_t score;
} HasherSearchResult;

void orc_hasher_free(Hasher* h);
void orc_choose_hasher(EncoderParams* params);                 /* encode.rs:834-893 */
void orc_hasher_setup(Hasher* h, EncoderParams* params, size_t ringbuffer_break, const uint8_t* data, size_t position,
                      size_t input_size, int is_last);        /* encode.rs:1125-1161 */
void orc_hasher_reset(Hasher* h);                              /* encode.rs:1118-1123 */
void orc_hasher_stitch(Hasher* h, size_t num_bytes, size_t position, const uint8_t* rb, size_t mask,
                       OrcStats* st);                          /* mod.rs:210-222 */
void orc_hasher_prepend_dictionary(Hasher* h, EncoderParams* params, size_t ringbuffer_break, size_t size,
                                   const uint8_t* dict, OrcStats* st); /* encode.rs:1163-1194 */
/* H10 + Zopfli (orc_zopfli.c) and the all-matches dictionary search (orc_static_dict.c) */
int orc_h10_init(Hasher* h, const EncoderParams* params, size_t ringbuffer_break);
int orc_h10_prepare(Hasher* h);
void orc_h10_store(Hasher* h, const uint8_t* data, size_t mask, size_t ix);
void orc_h10_stitch(Hasher* h, size_t num_bytes, size_t position, const uint8_t* ringbuffer, size_t ringbuffer_mask);
void orc_create_zopfli_backward_references(size_t num_bytes, size_t position, const uint8_t* ringbuffer,
                                           size_t ringbuffer_mask, size_t ringbuffer_break, const EncoderParams* params,
                                           Hasher* hasher, int32_t* dist_cache, size_t* last_insert_len,
                                           Command* commands, size_t* num_commands, size_t* num_literals);
void orc_create_hq_zopfli_backward_references(size_t num_bytes, size_t position, const uint8_t* ringbuffer,
                                              size_t ringbuffer_mask, size_t ringbuffer_break,
                                              const EncoderParams* params, Hasher* hasher, int32_t* dist_cache,
                                              size_t* last_insert_len, Command* commands, size_t* num_commands,
                                              size_t* num_literals);
int orc_find_all_static_dictionary_matches(const uint8_t* data, size_t min_length, size_t max_length,
                                           uint32_t* matches); /* static_dict.rs:309-1300 */
uint16_t orc_combine_length_codes(uint16_t inscode, uint16_t copycode, int use_last_distance); /* command.rs:110-125 */
extern int orc_reference_would_panic;

/* mod.rs:2376-2552 (+dispatcher 2553-2803) */
void orc_create_backward_references(size_t num_bytes, size_t position, const uint8_t* ringbuffer,
                                    size_t ringbuffer_mask, size_t ringbuffer_break /*0=none*/,
                                    const EncoderParams* params, Hasher* hasher, int32_t* dist_cache,
                                    size_t* last_insert_len, Command* commands, size_t* num_commands,
                                    size_t* num_literals, OrcStats* st);

/* command.rs */
void orc_command_init(Command* self, const DistanceParams* dist, size_t insertlen, size_t copylen,
                      size_t copylen_code, size_t distance_code);
void orc_command_init_insert(Command* self, size_t insertlen);
uint16_t orc_get_insert_length_code(size_t insertlen);
uint16_t orc_get_copy_length_code(size_t copylen);
void orc_get_length_code(size_t insertlen, size_t copylen, int use_last_distance, uint16_t* code);
uint32_t orc_command_restore_distance_code(const Command* self, const DistanceParams* dist);
static inline uint32_t orc_command_copy_len(const Command* c) { return c->copy_len_ & 0x01ffffffu; }
uint32_t orc_command_copy_len_code(const Command* c);

static inline uint32_t orc_log2_floor_nonzero(uint64_t v) { return 63u ^ (uint32_t)__builtin_clzll(v); }

/* bit_cost.rs:13-42, util.rs:17-25 */
float orc_fast_log2(uint64_t v);
float orc_shannon_entropy(const uint32_t* population, size_t size, size_t* total);
float orc_bits_entropy_impl(const uint32_t* population, size_t size);

/* ---- meta-block ---- */
typedef struct {
  size_t num_types;
  size_t num_blocks;
  uint8_t* types;
  uint32_t* lengths;
} BlockSplit;

#define ORC_NUM_LITERAL_SYMBOLS 256
#define ORC_NUM_COMMAND_SYMBOLS 704
#define ORC_NUM_DISTANCE_HISTO_SYMBOLS 544

typedef struct {
  BlockSplit literal_split, command_split, distance_split;
  uint32_t* literal_context_map;
  size_t literal_context_map_size;
  uint32_t* distance_context_map;
  size_t distance_context_map_size;
  uint32_t* literal_histograms; /* [n][256] */
  size_t literal_histograms_size;
  uint32_t* command_histograms; /* [n][704] */
  size_t command_histograms_size;
  uint32_t* distance_histograms; /* [n][544] */
  size_t distance_histograms_size;
} MetaBlockSplit;

void orc_metablock_destroy(MetaBlockSplit* mb);
/* metablock.rs:858-1075 */
void orc_build_meta_block_greedy(const uint8_t* ringbuffer, size_t pos, size_t mask, uint8_t prev_byte,
                                 uint8_t prev_byte2, int literal_context_mode, size_t num_contexts,
                                 const uint32_t* static_context_map, const Command* commands,
                                 size_t n_commands, MetaBlockSplit* mb);
/* metablock.rs:1076-1108 */
void orc_optimize_histograms(size_t num_distance_codes, MetaBlockSplit* mb);
/* brotli_bit_stream.rs:2035-2261 */
void orc_store_meta_block(const uint8_t* input, size_t start_pos, size_t length, size_t mask,
                          uint8_t prev_byte, uint8_t prev_byte2, int is_last, const EncoderParams* params,
                          int literal_context_mode, const Command* commands, size_t n_commands,
                          MetaBlockSplit* mb, size_t* storage_ix, uint8_t* storage);
/* brotli_bit_stream.rs:2775-2833 */
/* brotli_bit_stream.rs:2345-2465 (quality 3) and :2578-2742 (quality 2) */
void orc_store_meta_block_trivial(const uint8_t* input, size_t start_pos, size_t length, size_t mask, int is_last,
                                  const EncoderParams* params, const Command* commands, size_t n_commands,
                                  size_t* storage_ix, uint8_t* storage);
void orc_store_meta_block_fast(const uint8_t* input, size_t start_pos, size_t length, size_t mask, int is_last,
                               const EncoderParams* params, const Command* commands, size_t n_commands,
                               size_t* storage_ix, uint8_t* storage);
void orc_store_uncompressed_meta_block(int is_final_block, const uint8_t* input, size_t position,
                                       size_t mask, size_t len, size_t* storage_ix, uint8_t* storage);
void orc_write_bits(unsigned n_bits, uint64_t bits, size_t* pos, uint8_t* array);
void orc_write_padding_meta_block(size_t* storage_ix, uint8_t* storage);
void orc_write_empty_last_meta_block(size_t* storage_ix, uint8_t* storage);
void orc_write_metadata_meta_block(const EncoderParams* params, size_t* storage_ix, uint8_t* storage);

/* metablock.rs:133-307 (quality >= 10), orc_hq_metablock.c */
void orc_build_meta_block(const uint8_t* ringbuffer, size_t pos, size_t mask, EncoderParams* params, uint8_t prev_byte,
                          uint8_t prev_byte2, Command* cmds, size_t num_commands, int literal_context_mode,
                          MetaBlockSplit* mb);
float orc_population_cost(const uint32_t* data, size_t data_size, size_t total_count); /* bit_cost.rs:76-211 */
void orc_init_distance_params(EncoderParams* params, uint32_t npostfix, uint32_t ndirect); /* metablock.rs:28-60 */
void orc_prefix_encode_copy_distance(size_t distance_code, size_t num_direct_codes, uint64_t postfix_bits,
                                     uint16_t* code, uint32_t* extra_bits); /* command.rs:134-173 */
uint32_t orc_command_distance_context(const Command* c);                   /* command.rs:28-36 */
int orc_is_mostly_utf8(const uint8_t* data, size_t pos, size_t mask, size_t length, float min_fraction);

/* entropy coder pieces shared with orc_fragment.c */
void orc_create_huffman_tree(const uint32_t* data, size_t length, int tree_limit, uint8_t* depth);
void orc_store_huffman_tree(const uint8_t* depths, size_t num, size_t* storage_ix, uint8_t* storage);
void orc_convert_bit_depths_to_symbols(const uint8_t* depth, size_t len, uint16_t* bits);
void orc_build_and_store_huffman_tree_fast(const uint32_t* histogram, size_t histogram_total, size_t max_bits,
                                           uint8_t* depth, uint16_t* bits, size_t* storage_ix, uint8_t* storage);
/* compress_fragment_two_pass.rs (quality 1), orc_fragment.c */
void orc_compress_fragment_two_pass(const uint8_t* input, size_t input_size, int is_last, uint32_t* command_buf,
                                    uint8_t* literal_buf, int32_t* table, size_t table_size, size_t* storage_ix,
                                    uint8_t* storage);

/* compress_fragment.rs (quality 0) + InitCommandPrefixCodes (encode.rs:627-659), orc_fragment.c */
void orc_init_command_prefix_codes(uint8_t* cmd_depths, uint16_t* cmd_bits, uint8_t* cmd_code, size_t* cmd_code_numbits);
void orc_compress_fragment_fast(const uint8_t* input, size_t input_size, int is_last, int32_t* table, size_t table_size,
                                uint8_t* cmd_depth, uint16_t* cmd_bits, size_t* cmd_code_numbits, uint8_t* cmd_code,
                                size_t* storage_ix, uint8_t* storage);

uint8_t orc_context(uint8_t p1, uint8_t p2, int mode);
enum { ORC_CONTEXT_LSB6 = 0, ORC_CONTEXT_MSB6 = 1, ORC_CONTEXT_UTF8 = 2, ORC_CONTEXT_SIGNED = 3 };

#endif
