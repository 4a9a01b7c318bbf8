/* oracle/orc_lz77.c -- CPU restatement of the LZ77 half of rust-brotli's encoder hot path.
 * TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows, function by function:
 *   src/enc/backward_references/mod.rs  (H9 :598-917, AdvHasher H5/H5q5/H6 :919-1813,
 *       scoring :1151-1154,1871-1889, static dictionary probe :1891-1988,
 *       CreateBackwardReferences :2376-2552)
 *   src/enc/static_dict.rs:125-147      (FindMatchLengthWithLimit[Min4])
 *   src/enc/command.rs                  (distance code, length codes, Command::init)
 *   src/enc/encode.rs:834-893,1118-1194 (ChooseHasher, HasherReset, hasher_setup, dictionary prepend)
 */
#include "orc_internal.h"

#define BROTLI_TABLE_QUAL
#include "../tables/brotli_tables.h"

/* ------------------------------------------------------------------ tables */
const float* orc_logs_16(void) { return (const float*)(const void*)kBrotliLog2Table16_bits; }
const float* orc_logs_8(void) { return (const float*)(const void*)kBrotliLog2Table8_bits; }
const uint16_t* orc_static_dictionary_hash(void) { return kBrotliStaticDictionaryHash; }
const uint8_t* orc_dictionary_data(void) { return kBrotliDictionaryData; }
const uint32_t* orc_dictionary_offsets_by_length(void) { return kBrotliDictionaryOffsetsByLength; }
const uint8_t* orc_dictionary_size_bits_by_length(void) { return kBrotliDictionarySizeBitsByLength; }
const uint32_t* orc_ins_base(void) { return kBrotliInsBase; }
const uint32_t* orc_ins_extra(void) { return kBrotliInsExtra; }
const uint32_t* orc_copy_base(void) { return kBrotliCopyBase; }
const uint32_t* orc_copy_extra(void) { return kBrotliCopyExtra; }
const uint8_t* orc_utf8_context_lookup(void) { return kBrotliUTF8ContextLookup; }
const uint8_t* orc_signed3_context_lookup(void) { return kBrotliSigned3BitContextLookup; }

/* ------------------------------------------------------------------ loads */
static inline uint32_t load32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static inline uint64_t load64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

static const uint32_t kHashMul32 = 0x1e35a7bdu;
static const uint64_t kHashMul64Long = 0x1fe35a7bd3579bd3ull;
static const uint64_t kHashMul64 = 0x1e35a7bd1e35a7bdull;
#define IS_BASIC(h) ((h)->kind == 2 || (h)->kind == 3 || (h)->kind == 4 || (h)->kind == 54)

/* static_dict.rs:125-132 */
static size_t find_match_length_with_limit(const uint8_t* s1, const uint8_t* s2, size_t limit) {
  size_t i = 0;
  while (i + 8 <= limit) {
    uint64_t a = load64(s1 + i), b = load64(s2 + i);
    if (a != b) return i + ((size_t)__builtin_ctzll(a ^ b) >> 3);
    i += 8;
  }
  while (i < limit) {
    if (s1[i] != s2[i]) return i;
    ++i;
  }
  return limit;
}

/* static_dict.rs:134-147 */
static size_t find_match_length_with_limit_min4(const uint8_t* s1, const uint8_t* s2, size_t limit) {
  if (load32(s1) != load32(s2)) return 0;
  if (limit <= 4 || s1[4] != s2[4]) return ORC_MIN(limit, (size_t)4);
  return find_match_length_with_limit(s1 + 5, s2 + 5, limit - 5) + 5;
}

/* mod.rs:42-54 */
static size_t fix_unbroken_len(size_t unbroken_len, size_t prev_ix, size_t ring_buffer_break) {
  if (ring_buffer_break != 0) {
    if (prev_ix < ring_buffer_break && prev_ix + unbroken_len > ring_buffer_break)
      return ring_buffer_break - prev_ix;
  }
  return unbroken_len;
}

/* ------------------------------------------------------------------ command.rs */
/* command.rs:48-68 */
static size_t compute_distance_code(size_t distance, size_t max_distance, const int32_t* dist_cache) {
  if (distance <= max_distance) {
    size_t distance_plus_3 = distance + 3;
    size_t offset0 = distance_plus_3 - (size_t)(int64_t)dist_cache[0];
    size_t offset1 = distance_plus_3 - (size_t)(int64_t)dist_cache[1];
    if (distance == (size_t)(int64_t)dist_cache[0]) {
      return 0;
    } else if (distance == (size_t)(int64_t)dist_cache[1]) {
      return 1;
    } else if (offset0 < 7) {
      return (size_t)((0x09750468 >> (4 * offset0)) & 0xf);
    } else if (offset1 < 7) {
      return (size_t)((0x0fdb1ace >> (4 * offset1)) & 0xf);
    } else if (distance == (size_t)(int64_t)dist_cache[2]) {
      return 2;
    } else if (distance == (size_t)(int64_t)dist_cache[3]) {
      return 3;
    }
  }
  return distance + 16 - 1;
}

/* command.rs:71-92 */
uint16_t orc_get_insert_length_code(size_t insertlen) {
  if (insertlen < 6) {
    return (uint16_t)insertlen;
  } else if (insertlen < 130) {
    uint32_t nbits = orc_log2_floor_nonzero(insertlen - 2) - 1u;
    return (uint16_t)((nbits << 1) + ((insertlen - 2) >> nbits) + 2);
  } else if (insertlen < 2114) {
    return (uint16_t)(orc_log2_floor_nonzero(insertlen - 66) + 10);
  } else if (insertlen < 6210) {
    return 21;
  } else if (insertlen < 22594) {
    return 22;
  } else {
    return 23;
  }
}

/* Set when the restated code reaches a state in which the Rust reference panics (bounds-checked table index); the
   entry points report failure then, as the reference's FFI does after catching the panic (ffi/compressor.rs:419-434). */
int orc_reference_would_panic = 0;

/* TEST SWITCH (tests/test_oracle_vs_libbrotlienc.py only; 0 = the rust-brotli behaviour this oracle restates).
   Google's C encoder 1.0.9 differs from rust-brotli in the hash-table update after a copy: "Avoid hash poisoning with RLE
   data" (c/enc/backward_references_inc.h) starts StoreRange no earlier than position + len - 4 * distance when
   distance < len / 4; rust-brotli stores from position + 2 (backward_references/mod.rs:2516-2521).  With this switch on,
   the oracle follows the C rule, which makes its streams comparable byte for byte with libbrotlienc.so.1 and thereby pins
   every OTHER part of the restated path against an independent implementation. */
int orc_test_c109_rle_store_rule = 0;
/* Second difference in the same loop: inside a literal spree rust-brotli gives up the rest of a block as soon as a sparse
   jump would come within kMargin of its end (mod.rs:2529-2533); C 1.0.9 clamps the jump (pos_jump = min(position + 16,
   pos_end - kMargin)) and keeps storing every 4th / 2nd position up to there. */
int orc_test_c109_spree_tail = 0;

/* command.rs:94-108 */
uint16_t orc_get_copy_length_code(size_t copylen) {
  /* copylen 1 (a match cut by fix_unbroken_len) wraps to code 65535; kCopyBase[65535] panics in StoreCommandExtra */
  if (copylen < 2) {
    orc_reference_would_panic = 1;
    return 0; /* (keeps the rest of this run inside its tables; the result is discarded) */
  }
  if (copylen < 10) {
    return (uint16_t)(copylen - 2);
  } else if (copylen < 134) {
    uint32_t nbits = orc_log2_floor_nonzero(copylen - 6) - 1u;
    return (uint16_t)((nbits << 1) + ((copylen - 6) >> nbits) + 4);
  } else if (copylen < 2118) {
    return (uint16_t)(orc_log2_floor_nonzero(copylen - 70) + 12);
  } else {
    return 23;
  }
}

/* command.rs:110-125 */
uint16_t orc_combine_length_codes(uint16_t inscode, uint16_t copycode, int use_last_distance) {
  uint16_t bits64 = (uint16_t)((copycode & 0x7u) | ((inscode & 0x7u) << 3));
  if (use_last_distance && inscode < 8 && copycode < 16) {
    return (copycode < 8) ? bits64 : (uint16_t)(bits64 | 64);
  } else {
    int sub_offset = 2 * ((copycode >> 3) + 3 * (inscode >> 3));
    int offset = (sub_offset << 5) + 0x40 + ((0x520d40 >> sub_offset) & 0xc0);
    return (uint16_t)((uint16_t)offset | bits64);
  }
}

void orc_get_length_code(size_t insertlen, size_t copylen, int use_last_distance, uint16_t* code) {
  *code = orc_combine_length_codes(orc_get_insert_length_code(insertlen), orc_get_copy_length_code(copylen),
                               use_last_distance);
}

/* command.rs:134-173 */
void orc_prefix_encode_copy_distance(size_t distance_code, size_t num_direct_codes, uint64_t postfix_bits,
                                     uint16_t* code, uint32_t* extra_bits) {
  if (distance_code < 16 + num_direct_codes) {
    *code = (uint16_t)distance_code;
    *extra_bits = 0;
  } else {
    uint64_t dist = (1ull << (postfix_bits + 2)) + ((uint64_t)distance_code - 16 - (uint64_t)num_direct_codes);
    uint64_t bucket = (uint64_t)orc_log2_floor_nonzero(dist) - 1;
    uint64_t postfix_mask = (1u << postfix_bits) - 1;
    uint64_t postfix = dist & postfix_mask;
    uint64_t prefix = (dist >> bucket) & 1;
    uint64_t offset = (2 + prefix) << bucket;
    uint64_t nbits = bucket - postfix_bits;
    *code = (uint16_t)((nbits << 10) |
                       (16 + (uint64_t)num_direct_codes + ((2 * (nbits - 1) + prefix) << postfix_bits) + postfix));
    *extra_bits = (uint32_t)((dist - offset) >> postfix_bits);
  }
}

/* command.rs:273-297 */
void orc_command_init(Command* self, const DistanceParams* dist, size_t insertlen, size_t copylen,
                      size_t copylen_code, size_t distance_code) {
  self->insert_len_ = (uint32_t)insertlen;
  int8_t delta = (int8_t)((int32_t)copylen_code - (int32_t)copylen);
  self->copy_len_ = (uint32_t)copylen | ((uint32_t)(uint8_t)delta << 25);
  orc_prefix_encode_copy_distance(distance_code, dist->num_direct_distance_codes, dist->distance_postfix_bits,
                              &self->dist_prefix_, &self->dist_extra_);
  orc_get_length_code(insertlen, copylen_code, (self->dist_prefix_ & 0x3ff) == 0, &self->cmd_prefix_);
}

/* command.rs:38-44 */
void orc_command_init_insert(Command* self, size_t insertlen) {
  self->insert_len_ = (uint32_t)insertlen;
  self->copy_len_ = (uint32_t)(4 << 25);
  self->dist_extra_ = 0;
  self->dist_prefix_ = (uint16_t)((1u << 10) | 16u);
  orc_get_length_code(insertlen, 4, 0, &self->cmd_prefix_);
}

/* command.rs:176-201 */
uint32_t orc_command_restore_distance_code(const Command* self, const DistanceParams* dist) {
  if ((int)(self->dist_prefix_ & 0x3ff) < 16 + (int)dist->num_direct_distance_codes) {
    return (uint32_t)self->dist_prefix_ & 0x3ff;
  } else {
    uint32_t dcode = (uint32_t)self->dist_prefix_ & 0x3ff;
    uint32_t nbits = (uint32_t)(self->dist_prefix_ >> 10);
    uint32_t extra = self->dist_extra_;
    uint32_t postfix_mask = (1u << dist->distance_postfix_bits) - 1;
    uint32_t hcode = (dcode - dist->num_direct_distance_codes - 16u) >> dist->distance_postfix_bits;
    uint32_t lcode = (dcode - dist->num_direct_distance_codes - 16u) & postfix_mask;
    uint32_t offset = ((2u + (hcode & 1)) << nbits) - 4u;
    return ((offset + extra) << dist->distance_postfix_bits) + lcode + dist->num_direct_distance_codes + 16u;
  }
}

/* brotli_bit_stream.rs:1923-1929 */
uint32_t orc_command_copy_len_code(const Command* c) {
  uint32_t modifier = c->copy_len_ >> 25;
  int32_t delta = (int32_t)(int8_t)(uint8_t)(modifier | ((modifier & 0x40) << 1));
  return (uint32_t)((int32_t)(c->copy_len_ & 0x01ffffffu) + delta);
}

/* ------------------------------------------------------------------ hasher lifecycle */
void orc_hasher_free(Hasher* h) {
  free(h->num);
  free(h->buckets);
  free(h->forest);
  memset(h, 0, sizeof(*h));
}

/* encode.rs:834-893 */
/* TEST SWITCH, see orc_test_c109_rle_store_rule: C 1.0.9 (c/enc/quality.h ChooseHasher) takes H6 from size_hint >= 1 MiB
   (rust-brotli: > 4 MiB, encode.rs:863-865) and gives H5 14 bucket bits at every size below quality 7 (rust-brotli: only
   up to 1 MiB, encode.rs:880). */
int orc_test_c109_hasher_choice = 0;

void orc_choose_hasher(EncoderParams* params) {
  HasherParams* hp = &params->hasher;
  if (orc_test_c109_hasher_choice && params->quality >= 5 && params->quality <= 8 && params->lgwin > 16) {
    const int h6 = params->size_hint >= (1u << 20) && params->lgwin >= 19;
    hp->type_ = h6 ? 6 : 5;
    hp->block_bits = params->quality - 1;
    hp->bucket_bits = h6 ? 15 : (params->quality < 7 ? 14 : 15);
    if (h6) hp->hash_len = 5;
    hp->num_last_distances_to_check = params->quality < 7 ? 4 : 10;
    return;
  }
  if (params->quality >= 10 && !params->q9_5) {
    hp->type_ = 10;
  } else if (params->quality == 10 || params->quality == 9) {
    hp->type_ = 9;
    hp->num_last_distances_to_check = 16;
    hp->block_bits = 8;
    hp->bucket_bits = 15;
    hp->hash_len = 4;
  } else if (params->quality == 4 && params->size_hint >= (1u << 20)) {
    hp->type_ = 54;
  } else if (params->quality < 5) {
    hp->type_ = params->quality;
  } else if (params->lgwin <= 16) {
    hp->type_ = params->quality < 7 ? 40 : (params->quality < 9 ? 41 : 42);
  } else if (((params->q9_5 && params->size_hint > (1u << 20)) || params->size_hint > (1u << 22)) &&
             params->lgwin >= 19) {
    hp->type_ = 6;
    hp->block_bits = ORC_MIN(params->quality - 1, 9);
    hp->bucket_bits = 15;
    hp->hash_len = 5;
    hp->num_last_distances_to_check = params->quality < 7 ? 4 : (params->quality < 9 ? 10 : 16);
  } else {
    hp->type_ = 5;
    hp->block_bits = ORC_MIN(params->quality - 1, 9);
    hp->bucket_bits = (params->quality < 7 && params->size_hint <= (1u << 20)) ? 14 : 15;
    hp->num_last_distances_to_check = params->quality < 7 ? 4 : (params->quality < 9 ? 10 : 16);
  }
}

/* encode.rs:968-1117 (InitializeH5/H6/H9, BrotliMakeHasher). Types 40/41/42 are not implemented by
   the reference and fall back to InitializeH6 with whatever hasher params are current (:1115). */
static int make_hasher(Hasher* h, const EncoderParams* params, size_t ringbuffer_break) {
  int t = params->hasher.type_;
  memset(h, 0, sizeof(*h));
  h->params = params->hasher;
  if (t == 10) return orc_h10_init(h, params, ringbuffer_break); /* InitializeH10, orc_zopfli.c */
  h->literal_byte_score = params->hasher.literal_byte_score ? (uint32_t)params->hasher.literal_byte_score : 540u;
  if (t == 9) {
    h->kind = 9;
    h->bucket_bits = 15;
    h->block_bits = 8;
  } else if (t == 5) {
    h->kind = 5;
    h->bucket_bits = params->hasher.bucket_bits;
    h->block_bits = params->hasher.block_bits;
  } else if (t == 2 || t == 3 || t == 4 || t == 54) {
    /* InitializeH2/H3/H4/H54, encode.rs:895-966 + the BasicHashComputer impls mod.rs:437-560: zeroed tables of
       (1 << BUCKET_BITS) + BUCKET_SWEEP entries */
    h->kind = t;
    h->bucket_bits = t == 54 ? 20 : (t == 4 ? 17 : 16);
    h->sweep = t == 2 ? 1 : (t == 3 ? 2 : 4);
    h->basic_use_dictionary = (t == 2 || t == 4);
    h->basic_hash_len = t == 54 ? 7 : 5;
    h->bucket_count = ((size_t)1 << h->bucket_bits) + (size_t)h->sweep;
    h->buckets = (uint32_t*)calloc(h->bucket_count, sizeof(uint32_t));
    return h->buckets != NULL;
  } else {
    h->kind = 6;
    h->bucket_bits = params->hasher.bucket_bits;
    h->block_bits = params->hasher.block_bits;
    h->hash_mask = 0xffffffffffffffffull >> (64 - 8 * params->hasher.hash_len);
  }
  h->block_size = 1u << h->block_bits;
  h->block_mask = h->block_size - 1;
  h->bucket_count = (size_t)1 << h->bucket_bits;
  h->num = (uint16_t*)calloc(h->bucket_count, sizeof(uint16_t));
  h->buckets = (uint32_t*)calloc(h->bucket_count << h->block_bits, sizeof(uint32_t));
  return h->num && h->buckets;
}

void orc_hasher_reset(Hasher* h) {
  if (h->kind != 0) h->is_prepared_ = 0;
}

static inline size_t hash_bytes(const Hasher* h, const uint8_t* data) {
  if (IS_BASIC(h)) {
    /* mod.rs:437-441, 482-486, 502-506, 527-531 */
    uint64_t v = (load64(data) << (64 - 8 * h->basic_hash_len)) * kHashMul64;
    return (size_t)(v >> (64 - h->bucket_bits));
  }
  if (h->kind == 6) {
    /* mod.rs:1138-1140, 1521-1525 */
    uint64_t v = (load64(data) & h->hash_mask) * kHashMul64Long;
    return (size_t)(uint32_t)(v >> (64 - h->bucket_bits));
  } else if (h->kind == 5) {
    /* mod.rs:990-992 */
    uint64_t v = ((uint64_t)load32(data) * (uint64_t)kHashMul32) & 0xffffffffull;
    return (size_t)(uint32_t)(v >> (32 - h->bucket_bits));
  } else {
    /* H9 mod.rs:720-724 */
    uint32_t v = load32(data) * kHashMul32;
    return (size_t)(v >> (32 - 15));
  }
}

/* TEST SWITCH (tests/test_oracle_vs_libbrotlienc.py only; 0 = rust-brotli).  C 1.0.9 (hash_longest_match_quickly_inc.h)
   spreads the BUCKET_SWEEP slots of a key eight entries apart and wraps them inside the table: position ix goes to slot
   (key + (ix & ((BUCKET_SWEEP - 1) << 3))) & BUCKET_MASK, and a search looks at (key + (i << 3)) & BUCKET_MASK.  It also
   takes a candidate only when it beats the score found so far, where rust-brotli (a port of an older C version,
   mod.rs:322-327, 391-440) uses the slots key .. key + BUCKET_SWEEP - 1 and accepts the last-distance / single-slot match
   unconditionally. */
int orc_test_c109_basic_layout = 0;
static inline size_t basic_slot(const Hasher* h, size_t key, size_t i) {
  if (orc_test_c109_basic_layout && h->sweep > 1) return (key + (i << 3)) & (((size_t)1 << h->bucket_bits) - 1);
  return key + i;
}
static inline size_t basic_slot_of(const Hasher* h, size_t key, size_t ix) {
  return basic_slot(h, key, (ix >> 3) % (size_t)h->sweep);
}

static inline size_t hash_type_length(const Hasher* h) { return (h->kind == 6 || IS_BASIC(h)) ? 8 : 4; }
static inline size_t store_lookahead(const Hasher* h) {
  return (h->kind == 6 || IS_BASIC(h)) ? 8 : (h->kind == 10 ? 128 : 4);
}

/* test hook: when set, map[ix] |= 1 for every position inserted into the hash table and |= 2 for every position
   FindLongestMatch runs on (lets tests compare the device's stored / searched flags with the truth) */
uint8_t* orc_debug_store_map = 0;
size_t orc_debug_store_map_size = 0;
void orc_set_debug_store_map(uint8_t* map, size_t size) {
  orc_debug_store_map = map;
  orc_debug_store_map_size = size;
}

/* mod.rs:1644-1656 / 879-887 */
static inline void hasher_store(Hasher* h, const uint8_t* data, size_t mask, size_t ix, OrcStats* st) {
  if (h->kind == 10) {
    orc_h10_store(h, data, mask, ix);
    st->positions_stored++;
    return;
  }
  if (IS_BASIC(h)) { /* mod.rs:322-327 */
    size_t bkey = hash_bytes(h, data + (ix & mask));
    h->buckets[basic_slot_of(h, bkey, ix)] = (uint32_t)ix;
    st->positions_stored++;
    return;
  }
  size_t key = hash_bytes(h, data + (ix & mask));
  size_t minor_ix = (size_t)(h->num[key] & h->block_mask);
  h->buckets[minor_ix + (key << h->block_bits)] = (uint32_t)ix;
  h->num[key] = (uint16_t)(h->num[key] + 1);
  st->positions_stored++;
  if (orc_debug_store_map && ix < orc_debug_store_map_size) orc_debug_store_map[ix] |= 1;
}

/* mod.rs:1491-1510 (AdvHasher::Prepare), :898-906 (H9::Prepare). Returns 1 if newly prepared. */
static int hasher_prepare(Hasher* h, int one_shot, size_t input_size, const uint8_t* data) {
  if (h->kind == 10) return orc_h10_prepare(h);
  if (h->is_prepared_ != 0) return 0;
  if (IS_BASIC(h)) { /* mod.rs:336-357 */
    size_t partial_prepare_threshold = ((size_t)4 << h->bucket_bits) >> 7;
    if (one_shot && input_size <= partial_prepare_threshold) {
      for (size_t i = 0; i < input_size; ++i) {
        size_t key = hash_bytes(h, data + i);
        for (int j = 0; j < h->sweep; ++j) h->buckets[key + (size_t)j] = 0;
      }
    } else {
      memset(h->buckets, 0, h->bucket_count * sizeof(uint32_t));
    }
    h->is_prepared_ = 1;
    return 1;
  }
  if (h->kind == 9) {
    memset(h->num, 0, h->bucket_count * sizeof(uint16_t));
  } else {
    size_t partial_prepare_threshold = h->bucket_count >> 6;
    if (one_shot && input_size <= partial_prepare_threshold) {
      for (size_t i = 0; i < input_size; ++i) h->num[hash_bytes(h, data + i)] = 0;
    } else {
      memset(h->num, 0, h->bucket_count * sizeof(uint16_t));
    }
  }
  h->is_prepared_ = 1;
  return 1;
}

/* encode.rs:1125-1161 */
void orc_hasher_setup(Hasher* h, EncoderParams* params, size_t ringbuffer_break, const uint8_t* data, size_t position,
                      size_t input_size, int is_last) {
  int one_shot = (position == 0 && is_last);
  if (h->kind == 0) {
    orc_choose_hasher(params);
    if (!make_hasher(h, params, ringbuffer_break)) abort();
    h->params = params->hasher;
    h->is_prepared_ = 1; /* relies on zero-initialised tables (encode.rs:1147) */
  } else {
    if (hasher_prepare(h, one_shot, input_size, data)) {
      if (position == 0) {
        h->dict_num_lookups = 0;
        h->dict_num_matches = 0;
      }
    }
  }
}

/* mod.rs:210-222 */
void orc_hasher_stitch(Hasher* h, size_t num_bytes, size_t position, const uint8_t* rb, size_t mask,
                       OrcStats* st) {
  if (h->kind == 10) {
    orc_h10_stitch(h, num_bytes, position, rb, mask); /* StitchToPreviousBlockH10, hq.rs:254-300 */
    return;
  }
  if (num_bytes >= hash_type_length(h) - 1 && position >= 3) {
    hasher_store(h, rb, mask, position - 3, st);
    hasher_store(h, rb, mask, position - 2, st);
    hasher_store(h, rb, mask, position - 1, st);
  }
}

/* encode.rs:1163-1194 + mod.rs:224-229 (StoreLookaheadThenStore, mask = usize::MAX) */
void orc_hasher_prepend_dictionary(Hasher* h, EncoderParams* params, size_t ringbuffer_break, size_t size,
                                   const uint8_t* dict, OrcStats* st) {
  orc_hasher_setup(h, params, ringbuffer_break, dict, 0, size, 0);
  size_t overlap = store_lookahead(h) - 1;
  if (size > overlap) {
    for (size_t i = 0; i < size - overlap; ++i) hasher_store(h, dict, ~(size_t)0, i, st);
  }
}

/* ------------------------------------------------------------------ static dictionary (mod.rs:1891-1988) */
static const uint32_t kCutoffTransformsCount = 10;
static const uint64_t kCutoffTransforms = 0x071b520ada2d3200ull;

static inline uint64_t backward_reference_score(size_t copy_length, size_t backward, uint32_t lbs) {
  /* mod.rs:1878-1889 */
  return (uint64_t)(30 * 8 * 8) + (uint64_t)((size_t)(lbs >> 2) * copy_length) -
         30ull * (uint64_t)orc_log2_floor_nonzero((uint64_t)backward);
}
static inline uint64_t backward_reference_score_using_last_distance(size_t copy_length, uint32_t lbs) {
  /* mod.rs:1871-1876 */
  return ((uint64_t)(lbs >> 2)) * (uint64_t)copy_length + (uint64_t)(30 * 8 * 8) + 15;
}
static inline uint64_t backward_reference_penalty_using_last_distance(size_t distance_short_code) {
  /* mod.rs:1151-1154 */
  return 39ull + ((0x0001ca10ull >> (distance_short_code & 0x0e)) & 0x0e);
}

static int test_static_dictionary_item(size_t item, const uint8_t* data, size_t max_length, size_t max_backward,
                                       size_t max_distance, uint32_t lbs, HasherSearchResult* out) {
  size_t len = item & 0x1f;
  size_t dist = item >> 5;
  size_t offset = (size_t)kBrotliDictionaryOffsetsByLength[len] + len * dist;
  if (len > max_length) return 0;
  size_t matchlen = find_match_length_with_limit(data, &kBrotliDictionaryData[offset], len);
  if (matchlen + kCutoffTransformsCount <= len || matchlen == 0) return 0;
  size_t backward;
  {
    uint64_t cut = (uint64_t)(len - matchlen);
    size_t transform_id = (size_t)((cut << 2) + ((kCutoffTransforms >> (cut * 6)) & 0x3f));
    backward = max_backward + dist + 1 + (transform_id << kBrotliDictionarySizeBitsByLength[len]);
  }
  if (backward > max_distance) return 0;
  uint64_t score = backward_reference_score(matchlen, backward, lbs);
  if (score < out->score) return 0;
  out->len = matchlen;
  out->len_x_code = len ^ matchlen;
  out->distance = backward;
  out->score = score;
  return 1;
}

static int search_in_static_dictionary(Hasher* h, const uint8_t* data, size_t max_length, size_t max_backward,
                                       size_t max_distance, HasherSearchResult* out, int shallow, OrcStats* st) {
  int is_match_found = 0;
  if (h->dict_num_matches < (h->dict_num_lookups >> 7)) return 0;
  size_t key = (size_t)((load32(data) * kHashMul32) >> (32 - 14)) << 1; /* Hash14 << 1 */
  for (int i = 0; i < (shallow ? 1 : 2); ++i, ++key) {
    size_t item = kBrotliStaticDictionaryHash[key];
    h->dict_num_lookups++;
    st->dict_lookups++;
    if (item != 0) {
      if (test_static_dictionary_item(item, data, max_length, max_backward, max_distance,
                                      h->literal_byte_score, out)) {
        h->dict_num_matches++;
        st->dict_matches++;
        is_match_found = 1;
      }
    }
  }
  return is_match_found;
}

/* ------------------------------------------------------------------ AdvHasher::FindLongestMatch (mod.rs:1684-1812) */
static int adv_find_longest_match(Hasher* h, int use_dictionary, const uint8_t* data, size_t ring_buffer_mask,
                                  size_t ring_buffer_break, const int32_t* distance_cache, size_t cur_ix,
                                  size_t max_length, size_t max_backward, size_t gap, size_t max_distance,
                                  HasherSearchResult* out, OrcStats* st) {
  const uint32_t lbs = h->literal_byte_score;
  const size_t cur_ix_masked = cur_ix & ring_buffer_mask;
  int is_match_found = 0;
  uint64_t best_score = out->score;
  size_t best_len = out->len;
  const uint8_t* cur_data = data + cur_ix_masked;
  out->len = 0;
  out->len_x_code = 0;
  st->positions_searched++;
  for (size_t i = 0; i < (size_t)h->params.num_last_distances_to_check; ++i) {
    size_t backward = (size_t)(int64_t)distance_cache[i];
    size_t prev_ix = cur_ix - backward;
    if (prev_ix >= cur_ix || backward > max_backward) continue;
    prev_ix &= ring_buffer_mask;
    if (cur_ix_masked + best_len > ring_buffer_mask || prev_ix + best_len > ring_buffer_mask ||
        cur_data[best_len] != data[prev_ix + best_len])
      continue;
    size_t unbroken_len = find_match_length_with_limit(data + prev_ix, cur_data, max_length);
    if (unbroken_len >= 3 || (unbroken_len == 2 && i < 2)) {
      size_t len = fix_unbroken_len(unbroken_len, prev_ix, ring_buffer_break);
      uint64_t score = backward_reference_score_using_last_distance(len, lbs);
      if (best_score < score) {
        if (i != 0) score -= backward_reference_penalty_using_last_distance(i);
        if (best_score < score) {
          best_score = score;
          best_len = len;
          out->len = best_len;
          out->distance = backward;
          out->score = best_score;
          is_match_found = 1;
        }
      }
    }
  }
  {
    size_t key = hash_bytes(h, cur_data);
    uint16_t num_copy = h->num[key];
    uint32_t* bucket = &h->buckets[key << h->block_bits];
    if (num_copy != 0) {
      int32_t dn = (int32_t)num_copy - (int32_t)h->block_size;
      size_t down = dn > 0 ? (size_t)dn : 0;
      size_t i = num_copy;
      while (i > down) {
        --i;
        size_t prev_ix = bucket[i & h->block_mask];
        size_t backward = cur_ix - prev_ix;
        prev_ix &= ring_buffer_mask;
        if (cur_ix_masked + best_len > ring_buffer_mask || prev_ix + best_len > ring_buffer_mask ||
            cur_data[best_len] != data[prev_ix + best_len]) {
          if (backward > max_backward) break;
          continue;
        }
        if (backward > max_backward) break;
        size_t unbroken_len = find_match_length_with_limit_min4(data + prev_ix, cur_data, max_length);
        if (unbroken_len != 0) {
          size_t len = fix_unbroken_len(unbroken_len, prev_ix, ring_buffer_break);
          uint64_t score = backward_reference_score(len, backward, lbs);
          if (best_score < score) {
            best_score = score;
            best_len = len;
            out->len = best_len;
            out->distance = backward;
            out->score = best_score;
            is_match_found = 1;
          }
        }
      }
    }
    bucket[num_copy & h->block_mask] = (uint32_t)cur_ix;
    h->num[key] = (uint16_t)(num_copy + 1);
    st->positions_stored++;
    if (orc_debug_store_map && cur_ix < orc_debug_store_map_size) orc_debug_store_map[cur_ix] |= 3;
  }
  if (!is_match_found && use_dictionary) {
    is_match_found = search_in_static_dictionary(h, cur_data, max_length, max_backward + gap, max_distance,
                                                 out, 0, st);
  }
  return is_match_found;
}

/* ------------------------------------------------------------------ H9::FindLongestMatch (mod.rs:737-877) */
static const uint8_t kDistanceCacheIndex[16] = {0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
static const int8_t kDistanceCacheOffset[16] = {0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3};
#define H9_SCORE_BASE (120u * 8u * 8u)
static const uint32_t kDistanceShortCodeCost[16] = {
    H9_SCORE_BASE + 60,  H9_SCORE_BASE - 95,  H9_SCORE_BASE - 117, H9_SCORE_BASE - 127,
    H9_SCORE_BASE - 93,  H9_SCORE_BASE - 93,  H9_SCORE_BASE - 96,  H9_SCORE_BASE - 96,
    H9_SCORE_BASE - 99,  H9_SCORE_BASE - 99,  H9_SCORE_BASE - 105, H9_SCORE_BASE - 105,
    H9_SCORE_BASE - 115, H9_SCORE_BASE - 115, H9_SCORE_BASE - 125, H9_SCORE_BASE - 125};

static inline uint64_t score_h9(size_t copy_length, size_t backward, uint32_t lbs) {
  return ((uint64_t)H9_SCORE_BASE + (uint64_t)lbs * (uint64_t)copy_length -
          120ull * (uint64_t)orc_log2_floor_nonzero((uint64_t)backward)) >> 2;
}
static inline uint64_t score_last_h9(size_t copy_length, size_t code, uint32_t lbs) {
  return ((uint64_t)lbs * (uint64_t)copy_length + (uint64_t)kDistanceShortCodeCost[code]) >> 2;
}

static int h9_find_longest_match(Hasher* h, int use_dictionary, const uint8_t* data, size_t ring_buffer_mask,
                                 size_t ring_buffer_break, const int32_t* distance_cache, size_t cur_ix,
                                 size_t max_length, size_t max_backward, size_t gap, size_t max_distance,
                                 HasherSearchResult* out, OrcStats* st) {
  const uint32_t lbs = h->literal_byte_score;
  const size_t cur_ix_masked = cur_ix & ring_buffer_mask;
  uint64_t best_score = out->score;
  size_t best_len = out->len;
  int is_match_found = 0;
  out->len_x_code = 0;
  st->positions_searched++;
  for (size_t i = 0; i < 16; ++i) {
    size_t idx = kDistanceCacheIndex[i];
    size_t backward = (size_t)(int64_t)distance_cache[idx] + (size_t)(int64_t)kDistanceCacheOffset[i];
    size_t prev_ix = cur_ix - backward;
    if (prev_ix >= cur_ix) continue;
    if (backward > max_backward) continue;
    prev_ix &= ring_buffer_mask;
    if (cur_ix_masked + best_len > ring_buffer_mask || prev_ix + best_len > ring_buffer_mask ||
        data[cur_ix_masked + best_len] != data[prev_ix + best_len])
      continue;
    size_t unbroken_len = find_match_length_with_limit(data + prev_ix, data + cur_ix_masked, max_length);
    if (unbroken_len >= 3 || (unbroken_len == 2 && i < 2)) {
      size_t len = fix_unbroken_len(unbroken_len, prev_ix, ring_buffer_break);
      uint64_t score = score_last_h9(len, i, lbs);
      if (best_score < score) {
        best_score = score;
        best_len = len;
        out->len = best_len;
        out->distance = backward;
        out->score = best_score;
        is_match_found = 1;
      }
    }
  }
  if (max_length >= 4 && cur_ix_masked + best_len <= ring_buffer_mask) {
    size_t key = hash_bytes(h, data + cur_ix_masked);
    uint32_t* bucket = &h->buckets[key << 8];
    uint16_t self_num_key = h->num[key];
    size_t down = self_num_key > 256 ? (size_t)self_num_key - 256 : 0;
    size_t i = self_num_key;
    uint8_t prev_best_val = data[cur_ix_masked + best_len];
    while (i > down) {
      --i;
      size_t prev_ix = bucket[i & 255];
      size_t backward = cur_ix - prev_ix;
      if (backward > max_backward) break;
      prev_ix &= ring_buffer_mask;
      if (prev_ix + best_len > ring_buffer_mask || prev_best_val != data[prev_ix + best_len]) continue;
      size_t unbroken_len = find_match_length_with_limit(data + prev_ix, data + cur_ix_masked, max_length);
      if (unbroken_len >= 4) {
        size_t len = fix_unbroken_len(unbroken_len, prev_ix, ring_buffer_break);
        uint64_t score = score_h9(len, backward, lbs);
        if (best_score < score) {
          best_score = score;
          best_len = len;
          out->len = best_len;
          out->distance = backward;
          out->score = best_score;
          is_match_found = 1;
          if (cur_ix_masked + best_len > ring_buffer_mask) break;
          prev_best_val = data[cur_ix_masked + best_len];
        }
      }
    }
    bucket[self_num_key & 255] = (uint32_t)cur_ix;
    h->num[key] = (uint16_t)(self_num_key + 1);
    st->positions_stored++;
    if (orc_debug_store_map && cur_ix < orc_debug_store_map_size) orc_debug_store_map[cur_ix] |= 3;
  }
  if (!is_match_found && use_dictionary) {
    is_match_found = search_in_static_dictionary(h, data + cur_ix_masked, max_length, max_backward + gap,
                                                 max_distance, out, 0, st);
  }
  return is_match_found;
}

/* StoreRange: mod.rs:328-332 with StoreRangeOptBasic :254-285 for the basic hashers (which stores the MASKED position of
   the positions it takes in fours), a plain loop for the others */
/* TEST SWITCH (tests/test_oracle_vs_libbrotlienc.py only; 0 = rust-brotli): C 1.0.9 stores a range position by position
   (hash_longest_match_quickly_inc.h StoreRange).  rust-brotli's StoreRangeOptBasic takes four positions at a time and files
   all four under the sweep slot of the FIRST one ((i >> 3) % BUCKET_SWEEP with i the chunk start, mod.rs:264-279), and
   writes the masked position. */
int orc_test_c109_basic_store_range = 0;
/* TEST SWITCH, same purpose: C stores absolute positions in StoreRange for the H5 family too; rust-brotli's StoreRangeOptBatch
   stores masked ones (below).  1 = C behaviour. */
int orc_test_c109_adv_store_range = 0;
static void hasher_store_range(Hasher* h, const uint8_t* data, size_t mask, size_t ix_start, size_t ix_end, OrcStats* st) {
  size_t i = ix_start;
  if (IS_BASIC(h) && ix_end >= ix_start + 16 && !orc_test_c109_basic_store_range) {
    size_t chunk_count = (ix_end - ix_start) / 4;
    for (size_t chunk_id = 0; chunk_id < chunk_count; ++chunk_id) {
      size_t p = (ix_start + chunk_id * 4) & mask;
      size_t off = (p >> 3) % (size_t)h->sweep;
      for (size_t b = 0; b < 4; ++b) {
        h->buckets[hash_bytes(h, data + p + b) + off] = (uint32_t)p + (uint32_t)b;
        st->positions_stored++;
      }
    }
    i = ix_start + chunk_count * 4;
  }
  if (h->kind == 5 && ix_end >= ix_start + 8 && !orc_test_c109_adv_store_range) {
    /* AdvHasher::StoreRangeOptBatch, mod.rs:1163-1232 (StoreLookahead == 4, i.e. the H5 family only): four positions at a
       time, in ascending order like Store -- but what goes into the bucket is the MASKED position.  Past the first
       revolution of the ring buffer such an entry looks further away than max_backward to FindLongestMatch
       (backward = cur_ix - entry, :1763-1775), which ends its walk through the bucket there. */
    size_t chunk_count = (ix_end - ix_start) / 4;
    for (size_t c = 0; c < chunk_count * 4; ++c) {
      size_t p = ((ix_start + (c & ~(size_t)3)) & mask) + (c & 3);
      size_t key = hash_bytes(h, data + p);
      size_t minor_ix = (size_t)(h->num[key] & h->block_mask);
      h->buckets[minor_ix + (key << h->block_bits)] = (uint32_t)p;
      h->num[key] = (uint16_t)(h->num[key] + 1);
      st->positions_stored++;
      if (orc_debug_store_map && ix_start + c < orc_debug_store_map_size) orc_debug_store_map[ix_start + c] |= 1;
    }
    i = ix_start + chunk_count * 4;
  }
  for (; i < ix_end; ++i) hasher_store(h, data, mask, i, st);
}

/* mod.rs:359-473 */
static int basic_find_longest_match(Hasher* h, int use_dictionary, const uint8_t* data, size_t ring_buffer_mask,
                                    size_t ring_buffer_break, const int32_t* distance_cache, size_t cur_ix,
                                    size_t max_length, size_t max_backward, size_t gap, size_t max_distance,
                                    HasherSearchResult* out, OrcStats* st) {
  const uint32_t lbs = h->literal_byte_score;
  const size_t best_len_in = out->len;
  const size_t cur_ix_masked = cur_ix & ring_buffer_mask;
  const size_t key = hash_bytes(h, &data[cur_ix_masked]);
  int compare_char = data[cur_ix_masked + best_len_in];
  uint64_t best_score = out->score;
  size_t best_len = best_len_in;
  const size_t cached_backward = (size_t)(int64_t)distance_cache[0];
  size_t prev_ix = cur_ix - cached_backward;
  int is_match_found = 0;
  const size_t mask32 = (size_t)(uint32_t)ring_buffer_mask;
  st->positions_searched++;
  if (orc_debug_store_map && cur_ix < orc_debug_store_map_size) orc_debug_store_map[cur_ix] |= 3;
  out->len_x_code = 0;
  if (prev_ix < cur_ix) {
    prev_ix &= mask32;
    if (compare_char == data[prev_ix + best_len]) {
      size_t unbroken_len = find_match_length_with_limit_min4(&data[prev_ix], &data[cur_ix_masked], max_length);
      if (unbroken_len != 0 &&
          !(orc_test_c109_basic_layout && !(best_score < backward_reference_score_using_last_distance(unbroken_len, lbs)))) {
        size_t len = fix_unbroken_len(unbroken_len, prev_ix, ring_buffer_break);
        best_score = backward_reference_score_using_last_distance(len, lbs);
        best_len = len;
        out->len = len;
        out->distance = cached_backward;
        out->score = best_score;
        compare_char = data[cur_ix_masked + best_len];
        if (h->sweep == 1) {
          h->buckets[key] = (uint32_t)cur_ix;
          return 1;
        }
        is_match_found = 1;
      }
    }
  }
  if (h->sweep == 1) {
    prev_ix = h->buckets[key];
    h->buckets[key] = (uint32_t)cur_ix;
    size_t backward = cur_ix - prev_ix;
    prev_ix &= mask32;
    if (compare_char != data[prev_ix + best_len_in]) return 0;
    if (backward == 0 || backward > max_backward) return 0;
    size_t unbroken_len = find_match_length_with_limit_min4(&data[prev_ix], &data[cur_ix_masked], max_length);
    if (unbroken_len != 0 &&
        !(orc_test_c109_basic_layout && !(best_score < backward_reference_score(unbroken_len, backward, lbs)))) {
      size_t len = fix_unbroken_len(unbroken_len, prev_ix, ring_buffer_break);
      out->len = len;
      out->distance = backward;
      out->score = backward_reference_score(len, backward, lbs);
      return 1;
    }
  } else {
    for (int j = 0; j < h->sweep; ++j) {
      size_t p = h->buckets[basic_slot(h, key, (size_t)j)];
      size_t backward = cur_ix - p;
      p &= mask32;
      if (compare_char != data[p + best_len]) continue;
      if (backward == 0 || backward > max_backward) continue;
      size_t unbroken_len = find_match_length_with_limit_min4(&data[p], &data[cur_ix_masked], max_length);
      if (unbroken_len != 0) {
        size_t len = fix_unbroken_len(unbroken_len, p, ring_buffer_break);
        uint64_t score = backward_reference_score(len, backward, lbs);
        if (best_score < score) {
          best_score = score;
          best_len = len;
          out->len = best_len;
          out->distance = backward;
          out->score = score;
          compare_char = data[cur_ix_masked + best_len];
          is_match_found = 1;
        }
      }
    }
  }
  if (use_dictionary && h->basic_use_dictionary && !is_match_found)
    is_match_found = search_in_static_dictionary(h, &data[cur_ix_masked], max_length, max_backward + gap, max_distance,
                                                 out, 1, st);
  h->buckets[basic_slot_of(h, key, cur_ix)] = (uint32_t)cur_ix;
  return is_match_found;
}

static inline int find_longest_match(Hasher* h, int use_dictionary, const uint8_t* data, size_t mask,
                                     size_t rb_break, const int32_t* dc, size_t cur_ix, size_t max_length,
                                     size_t max_backward, size_t gap, size_t max_distance,
                                     HasherSearchResult* out, OrcStats* st) {
  if (IS_BASIC(h))
    return basic_find_longest_match(h, use_dictionary, data, mask, rb_break, dc, cur_ix, max_length, max_backward, gap,
                                    max_distance, out, st);
  if (h->kind == 9)
    return h9_find_longest_match(h, use_dictionary, data, mask, rb_break, dc, cur_ix, max_length, max_backward,
                                 gap, max_distance, out, st);
  return adv_find_longest_match(h, use_dictionary, data, mask, rb_break, dc, cur_ix, max_length, max_backward,
                                gap, max_distance, out, st);
}

/* mod.rs:632-651 */
static void prepare_distance_cache(const Hasher* h, int32_t* distance_cache) {
  if (IS_BASIC(h)) return; /* mod.rs:297-298 */
  int num_distances = h->kind == 9 ? 16 : h->params.num_last_distances_to_check;
  if (num_distances > 4) {
    int32_t last_distance = distance_cache[0];
    distance_cache[4] = last_distance - 1;
    distance_cache[5] = last_distance + 1;
    distance_cache[6] = last_distance - 2;
    distance_cache[7] = last_distance + 2;
    distance_cache[8] = last_distance - 3;
    distance_cache[9] = last_distance + 3;
    if (num_distances > 10) {
      int32_t next_last_distance = distance_cache[1];
      distance_cache[10] = next_last_distance - 1;
      distance_cache[11] = next_last_distance + 1;
      distance_cache[12] = next_last_distance - 2;
      distance_cache[13] = next_last_distance + 2;
      distance_cache[14] = next_last_distance - 3;
      distance_cache[15] = next_last_distance + 3;
    }
  }
}

/* ------------------------------------------------------------------ CreateBackwardReferences (mod.rs:2376-2552) */
void orc_create_backward_references(size_t num_bytes, size_t position, const uint8_t* ringbuffer,
                                    size_t ringbuffer_mask, size_t ringbuffer_break,
                                    const EncoderParams* params, Hasher* hasher, int32_t* dist_cache,
                                    size_t* last_insert_len, Command* commands, size_t* num_commands,
                                    size_t* num_literals, OrcStats* st) {
  if (hasher->kind == 10) { /* dispatcher, mod.rs:2576-2620 */
    if (params->quality >= 11) {
      orc_create_hq_zopfli_backward_references(num_bytes, position, ringbuffer, ringbuffer_mask, ringbuffer_break, params,
                                               hasher, dist_cache, last_insert_len, commands, num_commands, num_literals);
    } else {
      orc_create_zopfli_backward_references(num_bytes, position, ringbuffer, ringbuffer_mask, ringbuffer_break, params,
                                            hasher, dist_cache, last_insert_len, commands, num_commands, num_literals);
    }
    return;
  }
  const int use_dictionary = params->use_dictionary;
  const size_t gap = 0;
  const size_t max_backward_limit = ((size_t)1 << params->lgwin) - 16;
  size_t new_commands_count = 0;
  size_t insert_length = *last_insert_len;
  const size_t pos_end = position + num_bytes;
  const size_t store_end = num_bytes >= store_lookahead(hasher) ? position + num_bytes - store_lookahead(hasher) + 1
                                                                : position;
  const size_t random_heuristics_window_size = params->quality < 9 ? 64 : 512;
  size_t apply_random_heuristics = position + random_heuristics_window_size;
  const uint64_t kMinScore = 30 * 8 * 8 + 100;
  const size_t htl = hash_type_length(hasher);
  prepare_distance_cache(hasher, dist_cache);
  while (position + htl < pos_end) {
    size_t max_length = pos_end - position;
    size_t max_distance = ORC_MIN(position, max_backward_limit);
    HasherSearchResult sr;
    sr.len = 0;
    sr.len_x_code = 0;
    sr.distance = 0;
    sr.score = kMinScore;
    if (find_longest_match(hasher, use_dictionary, ringbuffer, ringbuffer_mask, ringbuffer_break, dist_cache,
                           position, max_length, max_distance, gap, params->dist.max_distance, &sr, st)) {
      int delayed_backward_references_in_row = 0;
      max_length--;
      for (;;) {
        const uint64_t cost_diff_lazy = 175;
        HasherSearchResult sr2;
        sr2.len = params->quality < 5 ? ORC_MIN(sr.len - 1, max_length) : 0;
        sr2.len_x_code = 0;
        sr2.distance = 0;
        sr2.score = kMinScore;
        max_distance = ORC_MIN(position + 1, max_backward_limit);
        int is_match_found =
            find_longest_match(hasher, use_dictionary, ringbuffer, ringbuffer_mask, ringbuffer_break, dist_cache,
                               position + 1, max_length, max_distance, gap, params->dist.max_distance, &sr2, st);
        if (is_match_found && sr2.score >= sr.score + cost_diff_lazy) {
          position++;
          insert_length++;
          sr = sr2;
          if (++delayed_backward_references_in_row < 4 && position + htl < pos_end) {
            max_length--;
            continue;
          }
        }
        break;
      }
      apply_random_heuristics = position + 2 * sr.len + random_heuristics_window_size;
      max_distance = ORC_MIN(position, max_backward_limit);
      {
        size_t distance_code = compute_distance_code(sr.distance, max_distance, dist_cache);
        if (sr.distance <= max_distance && distance_code > 0) {
          dist_cache[3] = dist_cache[2];
          dist_cache[2] = dist_cache[1];
          dist_cache[1] = dist_cache[0];
          dist_cache[0] = (int32_t)sr.distance;
          prepare_distance_cache(hasher, dist_cache);
        }
        orc_command_init(&commands[new_commands_count++], &params->dist, insert_length, sr.len,
                         sr.len ^ sr.len_x_code, distance_code);
      }
      *num_literals += insert_length;
      insert_length = 0;
      {
        size_t a = position + 2, b = ORC_MIN(position + sr.len, store_end);
        if (orc_test_c109_rle_store_rule && sr.distance < (sr.len >> 2))
          a = ORC_MIN(b, ORC_MAX(a, position + sr.len - (sr.distance << 2)));
        hasher_store_range(hasher, ringbuffer, ringbuffer_mask, a, b, st);
      }
      position += sr.len;
    } else {
      insert_length++;
      position++;
      if (position > apply_random_heuristics) {
        size_t kMargin = ORC_MAX(store_lookahead(hasher) - 1, (size_t)4);
        if (orc_test_c109_spree_tail) {
          if (position > apply_random_heuristics + 4 * random_heuristics_window_size) {
            size_t pos_jump = ORC_MIN(position + 16, pos_end - kMargin);
            for (; position < pos_jump; position += 4) {
              hasher_store(hasher, ringbuffer, ringbuffer_mask, position, st);
              insert_length += 4;
            }
          } else {
            size_t margin3 = ORC_MAX(store_lookahead(hasher) - 1, (size_t)3);
            size_t pos_jump = ORC_MIN(position + 8, pos_end - margin3);
            for (; position < pos_jump; position += 2) {
              hasher_store(hasher, ringbuffer, ringbuffer_mask, position, st);
              insert_length += 2;
            }
          }
        } else if (position + 16 >= pos_end - kMargin) {
          insert_length += pos_end - position;
          position = pos_end;
        } else if (position > apply_random_heuristics + 4 * random_heuristics_window_size) {
          for (size_t i = 0; i < 4; ++i) hasher_store(hasher, ringbuffer, ringbuffer_mask, position + i * 4, st);
          insert_length += 16;
          position += 16;
        } else {
          for (size_t i = 0; i < 4; ++i) hasher_store(hasher, ringbuffer, ringbuffer_mask, position + i * 2, st);
          insert_length += 8;
          position += 8;
        }
      }
    }
  }
  insert_length += pos_end - position;
  *last_insert_len = insert_length;
  *num_commands += new_commands_count;
}
