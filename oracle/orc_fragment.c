/* oracle/orc_fragment.c -- CPU restatement of rust-brotli's quality 0 and quality 1 encoders (one-pass and two-pass
 * fragment compressors; the quality 0 half starts further down).
 * TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows src/enc/compress_fragment_two_pass.rs:
 *   EmitInsertLen :19-48, EmitDistance :50-64, EmitCopyLenLastDistance :66-115, HashBytesAtOffset :116-119,
 *   EmitCopyLen :121-144, Hash :145-149, IsMatch :151-155, CreateCommands :157-385, ShouldCompress :387-406,
 *   store_meta_block_header :416-437, BuildAndStoreCommandPrefixCode :449-517, StoreCommands :519-629,
 *   EmitUncompressedMetaBlock :631-644, compress_fragment_two_pass_impl :646-703, compress_fragment_two_pass :752-905.
 * One thing the reference does differently from the C encoder it was ported from is kept as it is: after an
 * insert-and-copy command with 4-byte matching, the third of the three table updates hashes offset 0 again (:279-281).
 */
#include <math.h>
#include <stddef.h>

#include "orc_internal.h"

static const size_t kCompressFragmentTwoPassBlockSize = (size_t)1 << 17;
/* TEST SWITCH (tests/test_oracle_vs_libbrotlienc.py only; 0 = rust-brotli): with a 2^15-entry table C 1.0.9 still matches
   4 bytes (min_match = B <= 15 ? 4 : 6, compress_fragment_two_pass.c); rust-brotli switches to 6 there
   (`$table_bits < 15`, compress_fragment_two_pass.rs:723). */
int orc_test_c109_two_pass_min_match = 0;
static const uint32_t kHashMul32 = 0x1e35a7bdu;

static inline uint64_t load64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
static inline uint32_t load32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static size_t match_length(const uint8_t* s1, const uint8_t* s2, size_t limit) {
  size_t i = 0;
  while (i < limit && s1[i] == s2[i]) ++i;
  return i;
}

static inline void emit_insert_len(uint32_t insertlen, uint32_t** commands) { /* :19-48 */
  if (insertlen < 6) {
    **commands = insertlen;
  } else if (insertlen < 130) {
    uint32_t tail = insertlen - 2;
    uint32_t nbits = orc_log2_floor_nonzero(tail) - 1;
    uint32_t prefix = tail >> nbits;
    uint32_t inscode = (nbits << 1) + prefix + 2;
    uint32_t extra = tail - (prefix << nbits);
    **commands = inscode | (extra << 8);
  } else if (insertlen < 2114) {
    uint32_t tail = insertlen - 66;
    uint32_t nbits = orc_log2_floor_nonzero(tail);
    uint32_t code = nbits + 10;
    uint32_t extra = tail - (1u << nbits);
    **commands = code | (extra << 8);
  } else if (insertlen < 6210) {
    **commands = 21u | ((insertlen - 2114) << 8);
  } else if (insertlen < 22594) {
    **commands = 22u | ((insertlen - 6210) << 8);
  } else {
    **commands = 23u | ((insertlen - 22594) << 8);
  }
  ++*commands;
}

static inline void emit_distance(uint32_t distance, uint32_t** commands) { /* :50-64 */
  uint32_t d = distance + 3;
  uint32_t nbits = orc_log2_floor_nonzero(d) - 1;
  uint32_t prefix = (d >> nbits) & 1;
  uint32_t offset = (2 + prefix) << nbits;
  uint32_t distcode = 2 * (nbits - 1) + prefix + 80;
  uint32_t extra = d - offset;
  **commands = distcode | (extra << 8);
  ++*commands;
}

static inline void emit_copy_len_last_distance(size_t copylen, uint32_t** commands) { /* :66-115 */
  if (copylen < 12) {
    **commands = (uint32_t)(copylen + 20);
    ++*commands;
  } else if (copylen < 72) {
    size_t tail = copylen - 8;
    size_t nbits = orc_log2_floor_nonzero(tail) - 1;
    size_t prefix = tail >> nbits;
    size_t code = (nbits << 1) + prefix + 28;
    size_t extra = tail - (prefix << nbits);
    **commands = (uint32_t)(code | (extra << 8));
    ++*commands;
  } else if (copylen < 136) {
    size_t tail = copylen - 8;
    size_t code = (tail >> 5) + 54;
    size_t extra = tail & 31;
    **commands = (uint32_t)(code | (extra << 8));
    ++*commands;
    **commands = 64;
    ++*commands;
  } else if (copylen < 2120) {
    size_t tail = copylen - 72;
    size_t nbits = orc_log2_floor_nonzero(tail);
    size_t code = nbits + 52;
    size_t extra = tail - ((size_t)1 << nbits);
    **commands = (uint32_t)(code | (extra << 8));
    ++*commands;
    **commands = 64;
    ++*commands;
  } else {
    size_t extra = copylen - 2120;
    **commands = (uint32_t)(63 | (extra << 8));
    ++*commands;
    **commands = 64;
    ++*commands;
  }
}

static inline void emit_copy_len(size_t copylen, uint32_t** commands) { /* :121-144 */
  if (copylen < 10) {
    **commands = (uint32_t)(copylen + 38);
  } else if (copylen < 134) {
    size_t tail = copylen - 6;
    size_t nbits = orc_log2_floor_nonzero(tail) - 1;
    size_t prefix = tail >> nbits;
    size_t code = (nbits << 1) + prefix + 44;
    size_t extra = tail - (prefix << nbits);
    **commands = (uint32_t)(code | (extra << 8));
  } else if (copylen < 2118) {
    size_t tail = copylen - 70;
    size_t nbits = orc_log2_floor_nonzero(tail);
    size_t code = nbits + 52;
    size_t extra = tail - ((size_t)1 << nbits);
    **commands = (uint32_t)(code | (extra << 8));
  } else {
    size_t extra = copylen - 2118;
    **commands = (uint32_t)(63 | (extra << 8));
  }
  ++*commands;
}

static inline uint32_t hash_bytes_at_offset(uint64_t v, int offset, size_t shift, size_t length) { /* :116-119 */
  uint64_t h = ((v >> (8 * offset)) << ((8 - length) * 8)) * (uint64_t)kHashMul32;
  return (uint32_t)(h >> shift);
}
static inline uint32_t hash2(const uint8_t* p, size_t shift, size_t length) { /* :145-149 */
  uint64_t h = (load64(p) << ((8 - length) * 8)) * (uint64_t)kHashMul32;
  return (uint32_t)(h >> shift);
}
static inline int is_match2(const uint8_t* p1, const uint8_t* p2, size_t length) { /* :151-155 */
  return load32(p1) == load32(p2) && (length == 4 || (p1[4] == p2[4] && p1[5] == p2[5]));
}

/* :157-385 */
static void create_commands(size_t input_index, size_t block_size, size_t input_size, const uint8_t* base_ip,
                            int32_t* table, size_t table_bits, size_t min_match, uint8_t** literals,
                            uint32_t** commands) {
  size_t ip_index = input_index;
  const size_t shift = 64 - table_bits;
  const size_t ip_end = input_index + block_size;
  size_t next_emit = input_index;
  int32_t last_distance = -1;
  const size_t kInputMarginBytes = 16;
  if (block_size >= kInputMarginBytes) {
    const size_t len_limit = ORC_MIN(block_size - min_match, input_size - kInputMarginBytes);
    const size_t ip_limit = input_index + len_limit;
    uint32_t next_hash;
    int goto_emit_remainder = 0;
    next_hash = hash2(&base_ip[++ip_index], shift, min_match);
    while (!goto_emit_remainder) {
      uint32_t skip = 32;
      size_t next_ip = ip_index;
      size_t candidate = 0;
      for (;;) {
        for (;;) {
          uint32_t hash = next_hash;
          uint32_t bytes_between_hash_lookups = skip >> 5;
          ++skip;
          ip_index = next_ip;
          next_ip = ip_index + bytes_between_hash_lookups;
          if (next_ip > ip_limit) {
            goto_emit_remainder = 1;
            break;
          }
          next_hash = hash2(&base_ip[next_ip], shift, min_match);
          candidate = ip_index - (size_t)(int64_t)last_distance;
          /* (the reference evaluates IsMatch first; a candidate at or beyond ip_index reads bytes that exist either way) */
          if (candidate < ip_index && is_match2(&base_ip[ip_index], &base_ip[candidate], min_match)) {
            table[hash] = (int32_t)ip_index;
            break;
          }
          candidate = (size_t)(int64_t)table[hash];
          table[hash] = (int32_t)ip_index;
          if (is_match2(&base_ip[ip_index], &base_ip[candidate], min_match)) break;
        }
        if (!(ip_index - candidate > ((size_t)1 << 18) - 16 && !goto_emit_remainder)) break;
      }
      if (goto_emit_remainder) break;
      {
        const size_t base = ip_index;
        const size_t matched =
            min_match + match_length(&base_ip[candidate + min_match], &base_ip[ip_index + min_match],
                                     ip_end - ip_index - min_match);
        const int32_t distance = (int32_t)(base - candidate);
        const int32_t insert = (int32_t)(base - next_emit);
        ip_index += matched;
        emit_insert_len((uint32_t)insert, commands);
        memcpy(*literals, &base_ip[next_emit], (size_t)insert);
        *literals += insert;
        if (distance == last_distance) {
          **commands = 64;
          ++*commands;
        } else {
          emit_distance((uint32_t)distance, commands);
          last_distance = distance;
        }
        emit_copy_len_last_distance(matched, commands);
        next_emit = ip_index;
        if (ip_index >= ip_limit) {
          goto_emit_remainder = 1;
          break;
        }
        {
          uint64_t input_bytes;
          uint32_t prev_hash, cur_hash;
          if (min_match == 4) {
            input_bytes = load64(&base_ip[ip_index - 3]);
            cur_hash = hash_bytes_at_offset(input_bytes, 3, shift, min_match);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 3);
            prev_hash = hash_bytes_at_offset(input_bytes, 1, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 2);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match); /* sic: offset 0 again (:279) */
            table[prev_hash] = (int32_t)(ip_index - 1);
          } else {
            input_bytes = load64(&base_ip[ip_index - 5]);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 5);
            prev_hash = hash_bytes_at_offset(input_bytes, 1, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 4);
            prev_hash = hash_bytes_at_offset(input_bytes, 2, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 3);
            input_bytes = load64(&base_ip[ip_index - 2]);
            cur_hash = hash_bytes_at_offset(input_bytes, 2, shift, min_match);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 2);
            prev_hash = hash_bytes_at_offset(input_bytes, 1, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 1);
          }
          candidate = (size_t)(int64_t)table[cur_hash];
          table[cur_hash] = (int32_t)ip_index;
        }
      }
      while (ip_index - candidate <= ((size_t)1 << 18) - 16 && is_match2(&base_ip[ip_index], &base_ip[candidate], min_match)) {
        const size_t base_index = ip_index;
        const size_t matched =
            min_match + match_length(&base_ip[candidate + min_match], &base_ip[ip_index + min_match],
                                     ip_end - ip_index - min_match);
        ip_index += matched;
        last_distance = (int32_t)(base_index - candidate);
        emit_copy_len(matched, commands);
        emit_distance((uint32_t)last_distance, commands);
        next_emit = ip_index;
        if (ip_index >= ip_limit) {
          goto_emit_remainder = 1;
          break;
        }
        {
          uint64_t input_bytes;
          uint32_t cur_hash, prev_hash;
          if (min_match == 4) {
            input_bytes = load64(&base_ip[ip_index - 3]);
            cur_hash = hash_bytes_at_offset(input_bytes, 3, shift, min_match);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 3);
            prev_hash = hash_bytes_at_offset(input_bytes, 1, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 2);
            prev_hash = hash_bytes_at_offset(input_bytes, 2, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 1);
          } else {
            input_bytes = load64(&base_ip[ip_index - 5]);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 5);
            prev_hash = hash_bytes_at_offset(input_bytes, 1, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 4);
            prev_hash = hash_bytes_at_offset(input_bytes, 2, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 3);
            input_bytes = load64(&base_ip[ip_index - 2]);
            cur_hash = hash_bytes_at_offset(input_bytes, 2, shift, min_match);
            prev_hash = hash_bytes_at_offset(input_bytes, 0, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 2);
            prev_hash = hash_bytes_at_offset(input_bytes, 1, shift, min_match);
            table[prev_hash] = (int32_t)(ip_index - 1);
          }
          candidate = (size_t)(int64_t)table[cur_hash];
          table[cur_hash] = (int32_t)ip_index;
        }
      }
      if (!goto_emit_remainder) next_hash = hash2(&base_ip[++ip_index], shift, min_match);
    }
  }
  if (next_emit < ip_end) {
    const uint32_t insert = (uint32_t)(ip_end - next_emit);
    emit_insert_len(insert, commands);
    memcpy(*literals, &base_ip[next_emit], insert);
    *literals += insert;
  }
}

/* :387-406 */
static int should_compress_fragment(const uint8_t* input, size_t input_size, size_t num_literals) {
  const float corpus_size = (float)input_size;
  if ((float)num_literals < 0.98f * corpus_size) return 1;
  {
    uint32_t literal_histo[256] = {0};
    const float max_total_bit_cost = corpus_size * 8.0f * 0.98f / 43.0f;
    for (size_t i = 0; i < input_size; i += 43) literal_histo[input[i]]++;
    return orc_bits_entropy_impl(literal_histo, 256) < max_total_bit_cost;
  }
}

/* :416-437 */
void orc_fragment_store_meta_block_header(size_t len, int is_uncompressed, size_t* storage_ix, uint8_t* storage) {
  uint64_t nibbles = 6;
  orc_write_bits(1, 0, storage_ix, storage);
  if (len <= (1u << 16)) {
    nibbles = 4;
  } else if (len <= (1u << 20)) {
    nibbles = 5;
  }
  orc_write_bits(2, nibbles - 4, storage_ix, storage);
  orc_write_bits((unsigned)(nibbles * 4), len - 1, storage_ix, storage);
  orc_write_bits(1, is_uncompressed ? 1 : 0, storage_ix, storage);
}

/* :449-517 */
static void build_and_store_command_prefix_code(const uint32_t* histogram, uint8_t* depth /*[128]*/,
                                                uint16_t* bits /*[128]*/, size_t* storage_ix, uint8_t* storage) {
  uint8_t cmd_depth[704];
  uint16_t cmd_bits[64];
  memset(cmd_depth, 0, sizeof(cmd_depth));
  memset(cmd_bits, 0, sizeof(cmd_bits));
  orc_create_huffman_tree(histogram, 64, 15, depth);
  orc_create_huffman_tree(&histogram[64], 64, 14, &depth[64]);
  memcpy(cmd_depth, depth + 24, 24);
  memcpy(cmd_depth + 24, depth, 8);
  memcpy(cmd_depth + 32, depth + 48, 8);
  memcpy(cmd_depth + 40, depth + 8, 8);
  memcpy(cmd_depth + 48, depth + 56, 8);
  memcpy(cmd_depth + 56, depth + 16, 8);
  orc_convert_bit_depths_to_symbols(cmd_depth, 64, cmd_bits);
  memcpy(bits, cmd_bits + 24, 16 * sizeof(uint16_t));
  memcpy(bits + 8, cmd_bits + 40, 8 * sizeof(uint16_t));
  memcpy(bits + 16, cmd_bits + 56, 8 * sizeof(uint16_t));
  memcpy(bits + 24, cmd_bits, 48 * sizeof(uint16_t));
  memcpy(bits + 48, cmd_bits + 32, 8 * sizeof(uint16_t));
  memcpy(bits + 56, cmd_bits + 48, 8 * sizeof(uint16_t));
  orc_convert_bit_depths_to_symbols(&depth[64], 64, &bits[64]);
  {
    memset(cmd_depth, 0, 64);
    memcpy(cmd_depth, depth + 24, 8);
    memcpy(cmd_depth + 64, depth + 32, 8);
    memcpy(cmd_depth + 128, depth + 40, 8);
    memcpy(cmd_depth + 192, depth + 48, 8);
    memcpy(cmd_depth + 384, depth + 56, 8);
    for (size_t i = 0; i < 8; ++i) {
      cmd_depth[128 + 8 * i] = depth[i];
      cmd_depth[256 + 8 * i] = depth[i + 8];
      cmd_depth[448 + 8 * i] = depth[i + 16];
    }
    orc_store_huffman_tree(cmd_depth, 704, storage_ix, storage);
  }
  orc_store_huffman_tree(&depth[64], 64, storage_ix, storage);
}

/* :519-629 */
static void store_commands(const uint8_t* literals, size_t num_literals, const uint32_t* commands, size_t num_commands,
                           size_t* storage_ix, uint8_t* storage) {
  static const uint32_t kNumExtraBits[128] = {
      0,  0,  0,  0,  0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  5,  5,  6,  7,  8,  9,  10, 12, 14, 24, 0,  0,
      0,  0,  0,  0,  0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  0,  0,  0,  0,  0,  0,  0,  0,  1,  1,  2,  2,
      3,  3,  4,  4,  5,  5,  6,  7,  8,  9,  10, 24, 0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
      0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  5,  5,  6,  6,  7,  7,  8,  8,  9,  9,  10, 10, 11, 11, 12, 12,
      13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 24, 24};
  static const uint32_t kInsertOffset[24] = {0,  1,  2,  3,  4,   5,   6,   8,   10,   14,   18,   26,
                                             34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
  uint8_t lit_depths[256] = {0};
  uint16_t lit_bits[256] = {0};
  uint32_t lit_histo[256] = {0};
  uint8_t cmd_depths[128] = {0};
  uint16_t cmd_bits[128] = {0};
  uint32_t cmd_histo[128] = {0};
  for (size_t i = 0; i < num_literals; ++i) lit_histo[literals[i]]++;
  orc_build_and_store_huffman_tree_fast(lit_histo, num_literals, 8, lit_depths, lit_bits, storage_ix, storage);
  for (size_t i = 0; i < num_commands; ++i) cmd_histo[commands[i] & 0xff]++;
  cmd_histo[1] += 1;
  cmd_histo[2] += 1;
  cmd_histo[64] += 1;
  cmd_histo[84] += 1;
  build_and_store_command_prefix_code(cmd_histo, cmd_depths, cmd_bits, storage_ix, storage);
  for (size_t i = 0; i < num_commands; ++i) {
    const uint32_t cmd = commands[i];
    const uint32_t code = cmd & 0xff;
    const uint32_t extra = cmd >> 8;
    orc_write_bits(cmd_depths[code], cmd_bits[code], storage_ix, storage);
    orc_write_bits(kNumExtraBits[code], extra, storage_ix, storage);
    if (code < 24) {
      const uint32_t insert = kInsertOffset[code] + extra;
      for (uint32_t j = 0; j < insert; ++j) orc_write_bits(lit_depths[literals[j]], lit_bits[literals[j]], storage_ix, storage);
      literals += insert;
    }
  }
}

/* :631-644 */
static void emit_uncompressed_meta_block(const uint8_t* input, size_t input_size, size_t* storage_ix, uint8_t* storage) {
  orc_fragment_store_meta_block_header(input_size, 1, storage_ix, storage);
  *storage_ix = (*storage_ix + 7) & ~(size_t)7;
  memcpy(&storage[*storage_ix >> 3], input, input_size);
  *storage_ix += input_size << 3;
  storage[*storage_ix >> 3] = 0;
}

/* :741-750 */
static void rewind_bit_position(size_t new_storage_ix, size_t* storage_ix, uint8_t* storage) {
  const size_t bitpos = new_storage_ix & 7;
  const size_t mask = (1u << bitpos) - 1;
  storage[new_storage_ix >> 3] &= (uint8_t)mask;
  *storage_ix = new_storage_ix;
}

/* :646-703 + :752-905 */
void orc_compress_fragment_two_pass(const uint8_t* input, size_t input_size, int is_last, uint32_t* command_buf,
                                    uint8_t* literal_buf, int32_t* table, size_t table_size, size_t* storage_ix,
                                    uint8_t* storage) {
  const size_t initial_storage_ix = *storage_ix;
  const size_t table_bits = orc_log2_floor_nonzero(table_size);
  if (table_bits >= 8 && table_bits <= 17) {
    const size_t min_match = (table_bits < 15 || (table_bits == 15 && orc_test_c109_two_pass_min_match)) ? 4 : 6;
    size_t input_index = 0, remaining = input_size;
    while (remaining > 0) {
      const size_t block_size = ORC_MIN(remaining, kCompressFragmentTwoPassBlockSize);
      uint8_t* literals = literal_buf;
      uint32_t* commands = command_buf;
      create_commands(input_index, block_size, remaining, input, table, table_bits, min_match, &literals, &commands);
      const size_t num_literals = (size_t)(literals - literal_buf);
      const size_t num_commands = (size_t)(commands - command_buf);
      if (should_compress_fragment(&input[input_index], block_size, num_literals)) {
        orc_fragment_store_meta_block_header(block_size, 0, storage_ix, storage);
        orc_write_bits(13, 0, storage_ix, storage);
        store_commands(literal_buf, num_literals, command_buf, num_commands, storage_ix, storage);
      } else {
        emit_uncompressed_meta_block(&input[input_index], block_size, storage_ix, storage);
      }
      input_index += block_size;
      remaining -= block_size;
    }
  }
  if (*storage_ix - initial_storage_ix > 31 + (input_size << 3)) {
    rewind_bit_position(initial_storage_ix, storage_ix, storage);
    emit_uncompressed_meta_block(input, input_size, storage_ix, storage);
  }
  if (is_last) {
    orc_write_bits(1, 1, storage_ix, storage);
    orc_write_bits(1, 1, storage_ix, storage);
    *storage_ix = (*storage_ix + 7) & ~(size_t)7;
  }
}

/* ================================================================== quality 0: compress_fragment.rs
 * Hash :32-35, IsMatch :37-39, BuildAndStoreLiteralPrefixCode :41-125, EmitInsertLen :133-213, ShouldUseUncompressedMode
 * :215-224, EmitUncompressedMetaBlock :236-249, EmitLongInsertLen :251-286, EmitLiterals :288-305, EmitDistance :307-334,
 * EmitCopyLenLastDistance :336-446, HashBytesAtOffset :448-451, EmitCopyLen :453-532, ShouldMergeBlock :534-558, UpdateBits
 * :560-575, BuildAndStoreCommandPrefixCode :577-648, compress_fragment_fast_impl :650-1045, compress_fragment_fast
 * :1089-1179; InitCommandPrefixCodes encode.rs:627-659. */

/* TEST SWITCH (tests/test_oracle_vs_libbrotlienc.py only; 0 = rust-brotli): ShouldMergeBlock sums in f32 with log2f here
   (floatX, compress_fragment.rs:546-556); C 1.0.9 sums in double. */
int orc_test_c109_merge_block_double = 0;

static const uint32_t kCmdHistoSeed[128] = {
    0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0};

/* encode.rs:627-659 */
void orc_init_command_prefix_codes(uint8_t* cmd_depths /*[128]*/, uint16_t* cmd_bits /*[128]*/, uint8_t* cmd_code /*[512]*/,
                                   size_t* cmd_code_numbits) {
  static const uint8_t kDefaultCommandDepths[128] = {
      0,  4,  4,  5,  6,  6,  7,  7,  7,  7,  7,  8,  8,  8,  8,  8,  0,  0,  0,  4,  4,  4,  4,  4,  5,  5,
      6,  6,  6,  6,  7,  7,  7,  7,  10, 10, 10, 10, 10, 10, 0,  4,  4,  5,  5,  5,  6,  6,  7,  8,  8,  9,
      10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 5,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
      0,  0,  6,  6,  6,  6,  6,  6,  5,  5,  5,  5,  5,  5,  4,  4,  4,  4,  4,  4,  4,  5,  5,  5,  5,  5,
      5,  6,  6,  7,  7,  7,  8,  10, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 0,  0,  0,  0};
  static const uint16_t kDefaultCommandBits[128] = {
      0,   0,   8,   9,   3,    35,   7,    71,   39,   103,  23,   47,   175,  111,  239,  31,   0,  0,  0,  4,
      12,  2,   10,  6,   13,   29,   11,   43,   27,   59,   87,   55,   15,   79,   319,  831,  191, 703, 447, 959,
      0,   14,  1,   25,  5,    21,   19,   51,   119,  159,  95,   223,  479,  991,  63,   575,  127, 639, 383, 895,
      255, 767, 511, 1023, 14,  0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,   0,   0,   0,
      27,  59,  7,   39,  23,   55,   30,   1,    17,   9,    25,   5,    0,    8,    4,    12,   2,   10,  6,   21,
      13,  29,  3,   19,  11,   15,   47,   31,   95,   63,   127,  255,  767,  2815, 1791, 3839, 511, 2559, 1535, 3583,
      1023, 3071, 2047, 4095, 0, 0,   0,    0};
  static const uint8_t kDefaultCommandCode[57] = {
      0xff, 0x77, 0xd5, 0xbf, 0xe7, 0xde, 0xea, 0x9e, 0x51, 0x5d, 0xde, 0xc6, 0x70, 0x57, 0xbc, 0x58, 0x58, 0x58, 0xd8,
      0xd8, 0x58, 0xd5, 0xcb, 0x8c, 0xea, 0xe0, 0xc3, 0x87, 0x1f, 0x83, 0xc1, 0x60, 0x1c, 0x67, 0xb2, 0xaa, 0x06, 0x83,
      0xc1, 0x60, 0x30, 0x18, 0xcc, 0xa1, 0xce, 0x88, 0x54, 0x94, 0x46, 0xe1, 0xb0, 0xd0, 0x4e, 0xb2, 0xf7, 0x04, 0x00};
  memcpy(cmd_depths, kDefaultCommandDepths, sizeof(kDefaultCommandDepths));
  memcpy(cmd_bits, kDefaultCommandBits, sizeof(kDefaultCommandBits));
  memcpy(cmd_code, kDefaultCommandCode, sizeof(kDefaultCommandCode));
  *cmd_code_numbits = 448;
}

static inline uint32_t hash0(const uint8_t* p, size_t shift) { /* :32-35 */
  return (uint32_t)(((load64(p) << 24) * (uint64_t)kHashMul32) >> shift);
}
static inline uint32_t hash0_at_offset(uint64_t v, int offset, size_t shift) { /* :448-451 */
  return (uint32_t)((((v >> (8 * offset)) << 24) * (uint64_t)kHashMul32) >> shift);
}
static inline int is_match0(const uint8_t* p1, const uint8_t* p2) { return load32(p1) == load32(p2) && p1[4] == p2[4]; }

/* :41-125 */
static size_t build_and_store_literal_prefix_code(const uint8_t* input, size_t input_size, uint8_t* depths, uint16_t* bits,
                                                  size_t* storage_ix, uint8_t* storage) {
  uint32_t histogram[256] = {0};
  size_t histogram_total;
  if (input_size < (1u << 15)) {
    for (size_t i = 0; i < input_size; ++i) histogram[input[i]]++;
    histogram_total = input_size;
    for (size_t i = 0; i < 256; ++i) {
      const uint32_t adjust = 2 * ORC_MIN(histogram[i], 11u);
      histogram[i] += adjust;
      histogram_total += adjust;
    }
  } else {
    const size_t kSampleRate = 29;
    for (size_t i = 0; i < input_size; i += kSampleRate) histogram[input[i]]++;
    histogram_total = (input_size + kSampleRate - 1) / kSampleRate;
    for (size_t i = 0; i < 256; ++i) {
      const uint32_t adjust = 1 + 2 * ORC_MIN(histogram[i], 11u);
      histogram[i] += adjust;
      histogram_total += adjust;
    }
  }
  orc_build_and_store_huffman_tree_fast(histogram, histogram_total, 8, depths, bits, storage_ix, storage);
  {
    size_t literal_ratio = 0;
    for (size_t i = 0; i < 256; ++i)
      if (histogram[i] != 0) literal_ratio += (size_t)(uint32_t)(histogram[i] * (uint32_t)depths[i]);
    return literal_ratio * 125 / histogram_total;
  }
}

static void emit_insert_len0(size_t insertlen, const uint8_t* depth, const uint16_t* bits, uint32_t* histo,
                             size_t* storage_ix, uint8_t* storage) { /* :133-213 */
  if (insertlen < 6) {
    const size_t code = insertlen + 40;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    ++histo[code];
  } else if (insertlen < 130) {
    const size_t tail = insertlen - 2;
    const uint32_t nbits = orc_log2_floor_nonzero(tail) - 1;
    const size_t prefix = tail >> nbits;
    const size_t inscode = (nbits << 1) + prefix + 42;
    orc_write_bits(depth[inscode], bits[inscode], storage_ix, storage);
    orc_write_bits(nbits, tail - (prefix << nbits), storage_ix, storage);
    ++histo[inscode];
  } else if (insertlen < 2114) {
    const size_t tail = insertlen - 66;
    const uint32_t nbits = orc_log2_floor_nonzero(tail);
    const size_t code = nbits + 50;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    orc_write_bits(nbits, tail - ((size_t)1 << nbits), storage_ix, storage);
    ++histo[code];
  } else {
    orc_write_bits(depth[61], bits[61], storage_ix, storage);
    orc_write_bits(12, insertlen - 2114, storage_ix, storage);
    ++histo[61];
  }
}
static void emit_long_insert_len0(size_t insertlen, const uint8_t* depth, const uint16_t* bits, uint32_t* histo,
                                  size_t* storage_ix, uint8_t* storage) { /* :251-286 */
  if (insertlen < 22594) {
    orc_write_bits(depth[62], bits[62], storage_ix, storage);
    orc_write_bits(14, insertlen - 6210, storage_ix, storage);
    ++histo[62];
  } else {
    orc_write_bits(depth[63], bits[63], storage_ix, storage);
    orc_write_bits(24, insertlen - 22594, storage_ix, storage);
    ++histo[63];
  }
}
static void emit_literals0(const uint8_t* input, size_t len, const uint8_t* depth, const uint16_t* bits,
                           size_t* storage_ix, uint8_t* storage) { /* :288-305 */
  for (size_t j = 0; j < len; ++j) orc_write_bits(depth[input[j]], bits[input[j]], storage_ix, storage);
}
static void emit_distance0(size_t distance, const uint8_t* depth, const uint16_t* bits, uint32_t* histo,
                           size_t* storage_ix, uint8_t* storage) { /* :307-334 */
  const uint64_t d = distance + 3;
  const uint32_t nbits = orc_log2_floor_nonzero(d) - 1;
  const uint64_t prefix = (d >> nbits) & 1;
  const uint64_t offset = (2 + prefix) << nbits;
  const uint64_t distcode = 2 * (nbits - 1) + prefix + 80;
  orc_write_bits(depth[distcode], bits[distcode], storage_ix, storage);
  orc_write_bits(nbits, d - offset, storage_ix, storage);
  ++histo[distcode];
}
static void emit_copy_len_last_distance0(size_t copylen, const uint8_t* depth, const uint16_t* bits, uint32_t* histo,
                                         size_t* storage_ix, uint8_t* storage) { /* :336-446 */
  if (copylen < 12) {
    orc_write_bits(depth[copylen - 4], bits[copylen - 4], storage_ix, storage);
    ++histo[copylen - 4];
  } else if (copylen < 72) {
    const size_t tail = copylen - 8;
    const uint32_t nbits = orc_log2_floor_nonzero(tail) - 1;
    const size_t prefix = tail >> nbits;
    const size_t code = (nbits << 1) + prefix + 4;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    orc_write_bits(nbits, tail - (prefix << nbits), storage_ix, storage);
    ++histo[code];
  } else if (copylen < 136) {
    const size_t tail = copylen - 8;
    const size_t code = (tail >> 5) + 30;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    orc_write_bits(5, tail & 31, storage_ix, storage);
    orc_write_bits(depth[64], bits[64], storage_ix, storage);
    ++histo[code];
    ++histo[64];
  } else if (copylen < 2120) {
    const size_t tail = copylen - 72;
    const uint32_t nbits = orc_log2_floor_nonzero(tail);
    const size_t code = nbits + 28;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    orc_write_bits(nbits, tail - ((size_t)1 << nbits), storage_ix, storage);
    orc_write_bits(depth[64], bits[64], storage_ix, storage);
    ++histo[code];
    ++histo[64];
  } else {
    orc_write_bits(depth[39], bits[39], storage_ix, storage);
    orc_write_bits(24, copylen - 2120, storage_ix, storage);
    orc_write_bits(depth[64], bits[64], storage_ix, storage);
    ++histo[39];
    ++histo[64];
  }
}
static void emit_copy_len0(size_t copylen, const uint8_t* depth, const uint16_t* bits, uint32_t* histo, size_t* storage_ix,
                           uint8_t* storage) { /* :453-532 */
  if (copylen < 10) {
    orc_write_bits(depth[copylen + 14], bits[copylen + 14], storage_ix, storage);
    ++histo[copylen + 14];
  } else if (copylen < 134) {
    const size_t tail = copylen - 6;
    const uint32_t nbits = orc_log2_floor_nonzero(tail) - 1;
    const size_t prefix = tail >> nbits;
    const size_t code = (nbits << 1) + prefix + 20;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    orc_write_bits(nbits, tail - (prefix << nbits), storage_ix, storage);
    ++histo[code];
  } else if (copylen < 2118) {
    const size_t tail = copylen - 70;
    const uint32_t nbits = orc_log2_floor_nonzero(tail);
    const size_t code = nbits + 28;
    orc_write_bits(depth[code], bits[code], storage_ix, storage);
    orc_write_bits(nbits, tail - ((size_t)1 << nbits), storage_ix, storage);
    ++histo[code];
  } else {
    orc_write_bits(depth[39], bits[39], storage_ix, storage);
    orc_write_bits(24, copylen - 2118, storage_ix, storage);
    ++histo[39];
  }
}

static int should_use_uncompressed_mode(ptrdiff_t delta, size_t insertlen, size_t literal_ratio) { /* :215-224 */
  const size_t compressed = (size_t)delta;
  if (compressed * 50 > insertlen) return 0;
  return literal_ratio > 980;
}
static void emit_uncompressed_meta_block0(const uint8_t* begin, size_t len, size_t storage_ix_start, size_t* storage_ix,
                                          uint8_t* storage) { /* :236-249 */
  rewind_bit_position(storage_ix_start, storage_ix, storage);
  orc_fragment_store_meta_block_header(len, 1, storage_ix, storage);
  *storage_ix = (*storage_ix + 7) & ~(size_t)7;
  memcpy(&storage[*storage_ix >> 3], begin, len);
  *storage_ix += len << 3;
  storage[*storage_ix >> 3] = 0;
}

/* :534-558 */
static int should_merge_block(const uint8_t* data, size_t len, const uint8_t* depths) {
  size_t histo[256] = {0};
  const size_t kSampleRate = 43;
  for (size_t i = 0; i < len; i += kSampleRate) ++histo[data[i]];
  const size_t total = (len + kSampleRate - 1) / kSampleRate;
  if (orc_test_c109_merge_block_double) {
    double r = ((total < 256 ? (double)orc_logs_8()[total] : log2((double)total)) + 0.5) * (double)total + 200;
    for (size_t i = 0; i < 256; ++i)
      r -= (double)histo[i] * ((double)depths[i] + (histo[i] < 256 ? (double)orc_logs_8()[histo[i]] : log2((double)histo[i])));
    return r >= 0.0;
  }
  float r = (orc_fast_log2(total) + 0.5f) * (float)total + 200.0f;
  for (size_t i = 0; i < 256; ++i) r -= (float)histo[i] * ((float)depths[i] + orc_fast_log2(histo[i]));
  return r >= 0.0f;
}

/* :560-575 */
static void update_bits(size_t n_bits, uint32_t bits, size_t pos, uint8_t* array) {
  while (n_bits > 0) {
    const size_t byte_pos = pos >> 3;
    const size_t n_unchanged_bits = pos & 7;
    const size_t n_changed_bits = ORC_MIN(n_bits, 8 - n_unchanged_bits);
    const size_t total_bits = n_unchanged_bits + n_changed_bits;
    const uint32_t mask = (~((1u << total_bits) - 1u)) | ((1u << n_unchanged_bits) - 1u);
    const uint32_t unchanged_bits = array[byte_pos] & mask;
    const uint32_t changed_bits = bits & ((1u << n_changed_bits) - 1u);
    array[byte_pos] = (uint8_t)((changed_bits << n_unchanged_bits) | unchanged_bits);
    n_bits -= n_changed_bits;
    bits >>= n_changed_bits;
    pos += n_changed_bits;
  }
}

/* :577-648 */
static void build_and_store_command_prefix_code0(const uint32_t* histogram, uint8_t* depth /*[128]*/,
                                                 uint16_t* bits /*[128]*/, size_t* storage_ix, uint8_t* storage) {
  uint8_t cmd_depth[704];
  uint16_t cmd_bits[64];
  memset(cmd_depth, 0, sizeof(cmd_depth));
  memset(cmd_bits, 0, sizeof(cmd_bits));
  orc_create_huffman_tree(histogram, 64, 15, depth);
  orc_create_huffman_tree(&histogram[64], 64, 14, &depth[64]);
  memcpy(cmd_depth, depth, 24);
  memcpy(cmd_depth + 24, depth + 40, 8);
  memcpy(cmd_depth + 32, depth + 24, 8);
  memcpy(cmd_depth + 40, depth + 48, 8);
  memcpy(cmd_depth + 48, depth + 32, 8);
  memcpy(cmd_depth + 56, depth + 56, 8);
  orc_convert_bit_depths_to_symbols(cmd_depth, 64, cmd_bits);
  memcpy(bits, cmd_bits, 24 * sizeof(uint16_t));
  memcpy(bits + 24, cmd_bits + 32, 8 * sizeof(uint16_t));
  memcpy(bits + 32, cmd_bits + 48, 8 * sizeof(uint16_t));
  memcpy(bits + 40, cmd_bits + 24, 8 * sizeof(uint16_t));
  memcpy(bits + 48, cmd_bits + 40, 8 * sizeof(uint16_t));
  memcpy(bits + 56, cmd_bits + 56, 8 * sizeof(uint16_t));
  orc_convert_bit_depths_to_symbols(&depth[64], 64, &bits[64]);
  {
    memset(cmd_depth, 0, 64);
    memcpy(cmd_depth, depth, 8);
    memcpy(cmd_depth + 64, depth + 8, 8);
    memcpy(cmd_depth + 128, depth + 16, 8);
    memcpy(cmd_depth + 192, depth + 24, 8);
    memcpy(cmd_depth + 384, depth + 32, 8);
    for (size_t i = 0; i < 8; ++i) {
      cmd_depth[128 + 8 * i] = depth[i + 40];
      cmd_depth[256 + 8 * i] = depth[i + 48];
      cmd_depth[448 + 8 * i] = depth[i + 56];
    }
    orc_store_huffman_tree(cmd_depth, 704, storage_ix, storage);
  }
  orc_store_huffman_tree(&depth[64], 64, storage_ix, storage);
}

/* :650-1045 */
static void compress_fragment_fast_impl(const uint8_t* input_ptr, size_t input_size, int is_last, int32_t* table,
                                        size_t table_bits, uint8_t* cmd_depth, uint16_t* cmd_bits,
                                        size_t* cmd_code_numbits, uint8_t* cmd_code, size_t* storage_ix,
                                        uint8_t* storage) {
  enum { EMIT_REMAINDER, EMIT_COMMANDS, NEXT_BLOCK };
  uint32_t cmd_histo[128];
  size_t ip_end = 0, next_emit = 0;
  const size_t kFirstBlockSize = 3u << 15, kMergeBlockSize = 1u << 16;
  const size_t kInputMarginBytes = 16, kMinMatchLen = 5;
  const size_t kMaxDistance = ((size_t)1 << 18) - 16;
  size_t metablock_start = 0;
  size_t block_size = ORC_MIN(input_size, kFirstBlockSize);
  size_t total_block_size = block_size;
  size_t mlen_storage_ix = *storage_ix + 3;
  uint8_t lit_depth[256] = {0};
  uint16_t lit_bits[256] = {0};
  size_t literal_ratio;
  size_t input_index = 0;
  int32_t last_distance = -1;
  const size_t shift = 64 - table_bits;
  memset(cmd_histo, 0, sizeof(cmd_histo));
  orc_fragment_store_meta_block_header(block_size, 0, storage_ix, storage);
  orc_write_bits(13, 0, storage_ix, storage);
  literal_ratio = build_and_store_literal_prefix_code(&input_ptr[input_index], block_size, lit_depth, lit_bits, storage_ix,
                                                      storage);
  {
    size_t i = 0;
    for (; i + 7 < *cmd_code_numbits; i += 8) orc_write_bits(8, cmd_code[i >> 3], storage_ix, storage);
  }
  orc_write_bits((unsigned)(*cmd_code_numbits & 7), cmd_code[*cmd_code_numbits >> 3], storage_ix, storage);
  int state = EMIT_COMMANDS;
  for (;;) {
    if (state == EMIT_COMMANDS) {
      size_t ip_index = input_index;
      memcpy(cmd_histo, kCmdHistoSeed, sizeof(kCmdHistoSeed));
      last_distance = -1;
      ip_end = input_index + block_size;
      if (block_size >= kInputMarginBytes) {
        const size_t len_limit = ORC_MIN(block_size - kMinMatchLen, input_size - kInputMarginBytes);
        const size_t ip_limit = input_index + len_limit;
        uint32_t next_hash = hash0(&input_ptr[++ip_index], shift);
        int restart_outer = 0; /* `continue 'continue_to_next_block` from inside the match loops */
        for (;;) {
          uint32_t skip = 32;
          size_t next_ip = ip_index;
          size_t candidate = 0;
          for (;;) {
            for (;;) {
              const uint32_t hash = next_hash;
              const uint32_t bytes_between_hash_lookups = skip >> 5;
              ++skip;
              ip_index = next_ip;
              next_ip = ip_index + bytes_between_hash_lookups;
              if (next_ip > ip_limit) {
                state = EMIT_REMAINDER;
                break;
              }
              next_hash = hash0(&input_ptr[next_ip], shift);
              candidate = ip_index - (size_t)(int64_t)last_distance;
              if (candidate < ip_index && is_match0(&input_ptr[ip_index], &input_ptr[candidate])) {
                table[hash] = (int32_t)ip_index;
                break;
              }
              candidate = (size_t)(int64_t)table[hash];
              table[hash] = (int32_t)ip_index;
              if (is_match0(&input_ptr[ip_index], &input_ptr[candidate])) break;
            }
            if (!(ip_index - candidate > kMaxDistance && state == EMIT_COMMANDS)) break;
          }
          if (state != EMIT_COMMANDS) break;
          {
            const size_t base = ip_index;
            const size_t matched = 5 + match_length(&input_ptr[candidate + 5], &input_ptr[ip_index + 5], ip_end - ip_index - 5);
            const int32_t distance = (int32_t)(base - candidate);
            const size_t insert = base - next_emit;
            ip_index += matched;
            if (insert < 6210) {
              emit_insert_len0(insert, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
            } else if (should_use_uncompressed_mode((ptrdiff_t)next_emit - (ptrdiff_t)metablock_start, insert, literal_ratio)) {
              emit_uncompressed_meta_block0(&input_ptr[metablock_start], base - metablock_start, mlen_storage_ix - 3,
                                            storage_ix, storage);
              input_size -= base - input_index;
              input_index = base;
              next_emit = input_index;
              state = NEXT_BLOCK;
              restart_outer = 1;
              break;
            } else {
              emit_long_insert_len0(insert, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
            }
            emit_literals0(&input_ptr[next_emit], insert, lit_depth, lit_bits, storage_ix, storage);
            if (distance == last_distance) {
              orc_write_bits(cmd_depth[64], cmd_bits[64], storage_ix, storage);
              ++cmd_histo[64];
            } else {
              emit_distance0((size_t)distance, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
              last_distance = distance;
            }
            emit_copy_len_last_distance0(matched, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
            next_emit = ip_index;
            if (ip_index >= ip_limit) {
              state = EMIT_REMAINDER;
              restart_outer = 1;
              break;
            }
            {
              const uint64_t input_bytes = load64(&input_ptr[ip_index - 3]);
              uint32_t prev_hash = hash0_at_offset(input_bytes, 0, shift);
              const uint32_t cur_hash = hash0_at_offset(input_bytes, 3, shift);
              table[prev_hash] = (int32_t)(ip_index - 3);
              prev_hash = hash0_at_offset(input_bytes, 1, shift);
              table[prev_hash] = (int32_t)(ip_index - 2);
              prev_hash = hash0_at_offset(input_bytes, 2, shift);
              table[prev_hash] = (int32_t)(ip_index - 1);
              candidate = (size_t)(int64_t)table[cur_hash];
              table[cur_hash] = (int32_t)ip_index;
            }
            while (is_match0(&input_ptr[ip_index], &input_ptr[candidate])) {
              const size_t base2 = ip_index;
              const size_t matched2 = 5 + match_length(&input_ptr[candidate + 5], &input_ptr[ip_index + 5], ip_end - ip_index - 5);
              if (ip_index - candidate > kMaxDistance) break;
              ip_index += matched2;
              last_distance = (int32_t)(base2 - candidate);
              emit_copy_len0(matched2, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
              emit_distance0((size_t)last_distance, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
              next_emit = ip_index;
              if (ip_index >= ip_limit) {
                state = EMIT_REMAINDER;
                restart_outer = 1;
                break;
              }
              {
                const uint64_t input_bytes = load64(&input_ptr[ip_index - 3]);
                uint32_t prev_hash = hash0_at_offset(input_bytes, 0, shift);
                const uint32_t cur_hash = hash0_at_offset(input_bytes, 3, shift);
                table[prev_hash] = (int32_t)(ip_index - 3);
                prev_hash = hash0_at_offset(input_bytes, 1, shift);
                table[prev_hash] = (int32_t)(ip_index - 2);
                prev_hash = hash0_at_offset(input_bytes, 2, shift);
                table[prev_hash] = (int32_t)(ip_index - 1);
                candidate = (size_t)(int64_t)table[cur_hash];
                table[cur_hash] = (int32_t)ip_index;
              }
            }
            if (restart_outer) break;
            if (state == EMIT_REMAINDER) break;
            if (state == EMIT_COMMANDS) next_hash = hash0(&input_ptr[++ip_index], shift);
          }
        }
        if (restart_outer) continue;
      }
      state = EMIT_REMAINDER;
      continue;
    } else if (state == EMIT_REMAINDER) {
      input_index += block_size;
      input_size -= block_size;
      block_size = ORC_MIN(input_size, kMergeBlockSize);
      if (input_size > 0 && total_block_size + block_size <= (1u << 20) &&
          should_merge_block(&input_ptr[input_index], block_size, lit_depth)) {
        total_block_size += block_size;
        update_bits(20, (uint32_t)(total_block_size - 1), mlen_storage_ix, storage);
        state = EMIT_COMMANDS;
        continue;
      }
      if (next_emit < ip_end) {
        const size_t insert = ip_end - next_emit;
        if (insert < 6210) {
          emit_insert_len0(insert, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
          emit_literals0(&input_ptr[next_emit], insert, lit_depth, lit_bits, storage_ix, storage);
        } else if (should_use_uncompressed_mode((ptrdiff_t)next_emit - (ptrdiff_t)metablock_start, insert, literal_ratio)) {
          emit_uncompressed_meta_block0(&input_ptr[metablock_start], ip_end - metablock_start, mlen_storage_ix - 3,
                                        storage_ix, storage);
        } else {
          emit_long_insert_len0(insert, cmd_depth, cmd_bits, cmd_histo, storage_ix, storage);
          emit_literals0(&input_ptr[next_emit], insert, lit_depth, lit_bits, storage_ix, storage);
        }
      }
      next_emit = ip_end;
      state = NEXT_BLOCK;
      continue;
    } else { /* NEXT_BLOCK */
      if (input_size > 0) {
        metablock_start = input_index;
        block_size = ORC_MIN(input_size, kFirstBlockSize);
        total_block_size = block_size;
        mlen_storage_ix = *storage_ix + 3;
        orc_fragment_store_meta_block_header(block_size, 0, storage_ix, storage);
        orc_write_bits(13, 0, storage_ix, storage);
        literal_ratio = build_and_store_literal_prefix_code(&input_ptr[input_index], block_size, lit_depth, lit_bits,
                                                            storage_ix, storage);
        build_and_store_command_prefix_code0(cmd_histo, cmd_depth, cmd_bits, storage_ix, storage);
        state = EMIT_COMMANDS;
        continue;
      }
      break;
    }
  }
  if (!is_last) {
    cmd_code[0] = 0;
    *cmd_code_numbits = 0;
    build_and_store_command_prefix_code0(cmd_histo, cmd_depth, cmd_bits, cmd_code_numbits, cmd_code);
  }
}

/* :1089-1179 */
void orc_compress_fragment_fast(const uint8_t* input, size_t input_size, int is_last, int32_t* table, size_t table_size,
                                uint8_t* cmd_depth, uint16_t* cmd_bits, size_t* cmd_code_numbits, uint8_t* cmd_code,
                                size_t* storage_ix, uint8_t* storage) {
  const size_t initial_storage_ix = *storage_ix;
  const size_t table_bits = orc_log2_floor_nonzero(table_size);
  if (input_size == 0) {
    orc_write_bits(1, 1, storage_ix, storage);
    orc_write_bits(1, 1, storage_ix, storage);
    *storage_ix = (*storage_ix + 7) & ~(size_t)7;
    return;
  }
  if (table_bits == 9 || table_bits == 11 || table_bits == 13 || table_bits == 15)
    compress_fragment_fast_impl(input, input_size, is_last, table, table_bits, cmd_depth, cmd_bits, cmd_code_numbits,
                                cmd_code, storage_ix, storage);
  if (*storage_ix - initial_storage_ix > 31 + (input_size << 3))
    emit_uncompressed_meta_block0(input, input_size, initial_storage_ix, storage_ix, storage);
  if (is_last) {
    orc_write_bits(1, 1, storage_ix, storage);
    orc_write_bits(1, 1, storage_ix, storage);
    *storage_ix = (*storage_ix + 7) & ~(size_t)7;
  }
}
