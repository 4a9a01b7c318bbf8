/* oracle/orc_fragment.c -- CPU restatement of rust-brotli's quality 1 encoder (two-pass fragment compressor).
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
