// fragment_device.h -- qualities 0 and 1 as device code: the one-pass and the two-pass fragment compressor.
// Shared by the gfx950 kernel (fragment_kernels.hip) and the host emulation of the device seam (tests/emu, test infrastructure).
//
// What it replaces, per fragment of a BrotliEncoderCompressStream call (encode.rs:2706-2861):
//   compress_fragment_two_pass (quality 1)   compress_fragment_two_pass.rs:646-703, 752-905
//     CreateCommands :157-385, ShouldCompress :387-406, StoreCommands :519-629, BuildAndStoreCommandPrefixCode :449-517
//   compress_fragment_fast (quality 0)       compress_fragment.rs:650-1045, 1089-1179
//     BuildAndStoreLiteralPrefixCode :41-125, ShouldMergeBlock :534-558, UpdateBits :560-575, the Emit* family :133-532
//   BrotliBuildAndStoreHuffmanTreeFast       brotli_bit_stream.rs:925-1121 (metablock_fast.h)
//
// A fragment is a sequential object: its hash table holds the LAST position filed under a key and every search files its own;
// the quality-0 encoder emits bits while it parses and re-opens the meta-block header it wrote to extend the block (UpdateBits).
// One wavefront walks it in order -- all lanes run the same scalar code, lane 0 stores -- with the code tables, histograms and the
// Huffman builder's nodes in workgroup memory.  Fragments of one stream hang together through the bit position and (quality 0)
// the command prefix code; different streams are independent.
#ifndef BROTLI_MI355X_FRAGMENT_DEVICE_H_
#define BROTLI_MI355X_FRAGMENT_DEVICE_H_

#include "fragment_api.h"
#include "metablock_fast.h"

namespace brotli_mi355x {

#if BR_SCALAR
#define FR_FENCE() ((void)0)
#else
#define FR_FENCE() __threadfence()
#endif

// workgroup memory of one fragment job
struct FragmentScratch {
  HuffmanScratch huff;
  uint32_t lit_histo[256];
  uint8_t lit_depth[256];
  uint16_t lit_bits[256];
  uint32_t cmd_histo[128];
  uint8_t cmd_depth[128];
  uint16_t cmd_bits[128];
  uint8_t wide_depth[704];
  uint16_t wide_bits[64];
  uint64_t words[kTreeBitsWords];  // one serialised Huffman tree on its way into the stream
  uint32_t sample_histo[256];
};

// ---- bit writer: BrotliWriteBits (brotli_bit_stream.rs:742-757) into device memory.  Whole bytes are stored as they fill up, the
// open byte lives in a register; the few operations that go back (RewindBitPosition, UpdateBits) read it from memory.
struct FragmentOut {
  uint8_t* out;
  uint64_t pos;   // storage_ix
  uint64_t acc;   // the open byte (low `nacc` bits valid)
  uint32_t nacc;  // == pos & 7
  // position of the first jump to a byte boundary that still stands (~0: none).  Fragments of a batch are compressed from bit 0 of a
  // slot of their own; what comes behind this point lands on whole bytes whatever phase the stream really has (frag_join).
  uint64_t first_align;

  BR_DEV void put(uint32_t n_bits, uint64_t bits) {
    if (n_bits == 0) return;
    acc |= bits << nacc;
    const uint32_t total = nacc + n_bits;
    uint64_t at = pos >> 3;
    uint32_t left = total;
    while (left >= 8) {
      if (BR_LANE == 0) BR_LIVE_ST8(out + at, (uint8_t)acc);
      acc >>= 8;
      ++at;
      left -= 8;
    }
    nacc = left;
    pos += n_bits;
  }
  // the byte at index b of the stream as it stands (b <= pos >> 3)
  BR_DEV uint32_t byte_at(uint64_t b) const {
    if (b == (pos >> 3)) return (uint32_t)(acc & 0xffu);
#if BR_SCALAR
    return out[b];
#else
    const uint32_t* w = (const uint32_t*)((uintptr_t)(out + b) & ~(uintptr_t)3);
    const uint32_t v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return BR_UNIFORM((v >> (8u * (uint32_t)((uintptr_t)(out + b) & 3u))) & 0xffu);
#endif
  }
  BR_DEV void set_byte(uint64_t b, uint32_t v) {
    if (b == (pos >> 3)) {
      acc = (acc & ~0xffull) | (uint64_t)(v & 0xffu);
    } else if (BR_LANE == 0) {
      BR_LIVE_ST8(out + b, (uint8_t)v);
    }
  }
  // RewindBitPosition, compress_fragment_two_pass.rs:741-750
  BR_DEV void rewind(uint64_t new_pos) {
    if (first_align != ~0ull && first_align >= new_pos) first_align = ~0ull;  // (that jump is taken back)
    const uint32_t keep = (uint32_t)(new_pos & 7u);
    const uint32_t v = byte_at(new_pos >> 3) & ((1u << keep) - 1u);
    pos = new_pos;
    acc = v;
    nacc = keep;
  }
  // UpdateBits, compress_fragment.rs:560-575
  BR_DEV void update_bits(uint32_t n_bits, uint32_t bits, uint64_t at) {
    while (n_bits > 0) {
      const uint64_t byte_pos = at >> 3;
      const uint32_t n_unchanged = (uint32_t)(at & 7u);
      const uint32_t n_changed = n_bits < 8u - n_unchanged ? n_bits : 8u - n_unchanged;
      const uint32_t total_bits = n_unchanged + n_changed;
      const uint32_t mask = (~((1u << total_bits) - 1u)) | ((1u << n_unchanged) - 1u);
      const uint32_t unchanged = byte_at(byte_pos) & mask;
      const uint32_t changed = bits & ((1u << n_changed) - 1u);
      set_byte(byte_pos, ((changed << n_unchanged) | unchanged) & 0xffu);
      n_bits -= n_changed;
      bits >>= n_changed;
      at += n_changed;
    }
  }
  BR_DEV void align() {
    if (first_align == ~0ull) first_align = pos;
    if (nacc != 0) {
      if (BR_LANE == 0) BR_LIVE_ST8(out + (pos >> 3), (uint8_t)acc);
      pos += 8u - nacc;
      acc = 0;
      nacc = 0;
    }
  }
  // the open byte goes to memory as well (the host reads the stream from there)
  BR_DEV void park() {
    if (BR_LANE == 0) BR_LIVE_ST8(out + (pos >> 3), (uint8_t)(acc & ((1u << nacc) - 1u)));
  }
};

// appends the first `nbits` bits of a word-aligned bit string
BR_DEV void fr_append(FragmentOut& o, const uint64_t* words, uint32_t nbits) {
  uint32_t i = 0;
  while (nbits >= 32) {
    o.put(32, (words[i >> 1] >> ((i & 1) * 32)) & 0xffffffffull);
    nbits -= 32;
    i++;
  }
  if (nbits) o.put(nbits, (words[i >> 1] >> ((i & 1) * 32)) & ((1ull << nbits) - 1));
}
BR_DEV BitSink fr_sink(FragmentScratch& S) {
  for (uint32_t i = 0; i < kTreeBitsWords; ++i) S.words[i] = 0;
  BitSink k;
  k.words = S.words;
  k.pos = 0;
  return k;
}

// store_meta_block_header, compress_fragment_two_pass.rs:416-437 (the same in compress_fragment.rs)
BR_DEV void fr_store_meta_block_header(uint32_t len, bool is_uncompressed, FragmentOut& o) {
  uint32_t nibbles = 6;
  o.put(1, 0);
  if (len <= (1u << 16)) {
    nibbles = 4;
  } else if (len <= (1u << 20)) {
    nibbles = 5;
  }
  o.put(2, nibbles - 4);
  o.put(nibbles * 4, len - 1);
  o.put(1, is_uncompressed ? 1 : 0);
}
// EmitUncompressedMetaBlock: header, byte boundary, the bytes (all lanes copy)
BR_DEV void fr_emit_uncompressed(const uint8_t* input, uint32_t len, FragmentOut& o) {
  fr_store_meta_block_header(len, true, o);
  o.align();
  uint8_t* dst = o.out + (o.pos >> 3);
  for (uint32_t i = BR_LANE; i < len; i += BR_NLANES) dst[i] = input[i];
  FR_FENCE();
  o.pos += (uint64_t)len << 3;
}

BR_DEV uint32_t fr_match_length(const uint8_t* s1, const uint8_t* s2, uint32_t limit) { return BR_UNIFORM(br_match_len_wide(s1, s2, limit)); }

// BuildAndStoreHuffmanTreeFast into the stream
BR_DEV void fr_huffman_fast(const uint32_t* histogram, uint32_t total, uint32_t max_bits, uint8_t* depth, uint16_t* bits, FragmentScratch& S, FragmentOut& o) {
  BitSink k = fr_sink(S);
  br_build_and_store_huffman_tree_fast(histogram, total, max_bits, &S.huff, depth, bits, k);
  fr_append(o, S.words, (uint32_t)k.pos);
}
BR_DEV void fr_store_huffman_tree(const uint8_t* depths, uint32_t num, FragmentScratch& S, FragmentOut& o) {
  BitSink k = fr_sink(S);
  br_store_huffman_tree(depths, num, &S.huff, k);
  fr_append(o, S.words, (uint32_t)k.pos);
}

// ================================================================================================== quality 1
BR_DEV uint32_t fr2_insert_len(uint32_t insertlen) {  // EmitInsertLen :19-48
  if (insertlen < 6) return insertlen;
  if (insertlen < 130) {
    const uint32_t tail = insertlen - 2;
    const uint32_t nbits = br_log2_floor_nonzero(tail) - 1;
    const uint32_t prefix = tail >> nbits;
    return ((nbits << 1) + prefix + 2) | ((tail - (prefix << nbits)) << 8);
  }
  if (insertlen < 2114) {
    const uint32_t tail = insertlen - 66;
    const uint32_t nbits = br_log2_floor_nonzero(tail);
    return (nbits + 10) | ((tail - (1u << nbits)) << 8);
  }
  if (insertlen < 6210) return 21u | ((insertlen - 2114) << 8);
  if (insertlen < 22594) return 22u | ((insertlen - 6210) << 8);
  return 23u | ((insertlen - 22594) << 8);
}
BR_DEV uint32_t fr2_distance(uint32_t distance) {  // EmitDistance :50-64
  const uint32_t d = distance + 3;
  const uint32_t nbits = br_log2_floor_nonzero(d) - 1;
  const uint32_t prefix = (d >> nbits) & 1;
  const uint32_t offset = (2 + prefix) << nbits;
  return (2 * (nbits - 1) + prefix + 80) | ((d - offset) << 8);
}
struct FragmentCmds {
  uint32_t* at;
  uint32_t n;
  BR_DEV void push(uint32_t v) {
    if (BR_LANE == 0) at[n] = v;
    ++n;
  }
};
BR_DEV void fr2_copy_len_last_distance(uint32_t copylen, FragmentCmds& c) {  // EmitCopyLenLastDistance :66-115
  if (copylen < 12) {
    c.push(copylen + 20);
  } else if (copylen < 72) {
    const uint32_t tail = copylen - 8;
    const uint32_t nbits = br_log2_floor_nonzero(tail) - 1;
    const uint32_t prefix = tail >> nbits;
    c.push(((nbits << 1) + prefix + 28) | ((tail - (prefix << nbits)) << 8));
  } else if (copylen < 136) {
    const uint32_t tail = copylen - 8;
    c.push(((tail >> 5) + 54) | ((tail & 31) << 8));
    c.push(64);
  } else if (copylen < 2120) {
    const uint32_t tail = copylen - 72;
    const uint32_t nbits = br_log2_floor_nonzero(tail);
    c.push((nbits + 52) | ((tail - (1u << nbits)) << 8));
    c.push(64);
  } else {
    c.push(63u | ((copylen - 2120) << 8));
    c.push(64);
  }
}
BR_DEV void fr2_copy_len(uint32_t copylen, FragmentCmds& c) {  // EmitCopyLen :121-144
  if (copylen < 10) {
    c.push(copylen + 38);
  } else if (copylen < 134) {
    const uint32_t tail = copylen - 6;
    const uint32_t nbits = br_log2_floor_nonzero(tail) - 1;
    const uint32_t prefix = tail >> nbits;
    c.push(((nbits << 1) + prefix + 44) | ((tail - (prefix << nbits)) << 8));
  } else if (copylen < 2118) {
    const uint32_t tail = copylen - 70;
    const uint32_t nbits = br_log2_floor_nonzero(tail);
    c.push((nbits + 52) | ((tail - (1u << nbits)) << 8));
  } else {
    c.push(63u | ((copylen - 2118) << 8));
  }
}
BR_DEV uint32_t fr2_hash_at(uint64_t v, uint32_t offset, uint32_t shift, uint32_t length) {  // HashBytesAtOffset :116-119
  const uint64_t h = ((v >> (8 * offset)) << ((8 - length) * 8)) * (uint64_t)0x1e35a7bdu;
  return (uint32_t)(h >> shift);
}
BR_DEV uint32_t fr2_hash(const uint8_t* p, uint32_t shift, uint32_t length) { return BR_UNIFORM(fr2_hash_at(br_load64(p), 0, shift, length)); }
BR_DEV bool fr2_is_match(const uint8_t* p1, const uint8_t* p2, uint32_t length) {  // IsMatch :151-155
  return br_load32(p1) == br_load32(p2) && (length == 4 || (p1[4] == p2[4] && p1[5] == p2[5]));
}
BR_DEV uint32_t fr_table_get(uint32_t* table, uint32_t h) { return BR_UNIFORM(BR_LIVE_LD32(table + h)); }
BR_DEV void fr_table_put(uint32_t* table, uint32_t h, uint32_t v) {
  if (BR_LANE == 0) BR_LIVE_ST32(table + h, v);
}
BR_DEV void fr_copy_literals(uint8_t* dst, const uint8_t* src, uint32_t n) {
  for (uint32_t i = BR_LANE; i < n; i += BR_NLANES) dst[i] = src[i];
}

// the table updates behind a copy (:262-330): the positions just in front of ip, then the look-up at ip itself
BR_DEV uint32_t fr2_after_copy(const uint8_t* base_ip, uint32_t ip_index, uint32_t* table, uint32_t shift, uint32_t min_match, bool first) {
  uint32_t cur_hash;
  if (min_match == 4) {
    const uint64_t input_bytes = br_load64(base_ip + ip_index - 3);
    cur_hash = fr2_hash_at(input_bytes, 3, shift, min_match);
    fr_table_put(table, fr2_hash_at(input_bytes, 0, shift, min_match), ip_index - 3);
    fr_table_put(table, fr2_hash_at(input_bytes, 1, shift, min_match), ip_index - 2);
    // (behind an insert-and-copy command the reference hashes offset 0 again for ip - 1, compress_fragment_two_pass.rs:279-281)
    fr_table_put(table, fr2_hash_at(input_bytes, first ? 0 : 2, shift, min_match), ip_index - 1);
  } else {
    uint64_t input_bytes = br_load64(base_ip + ip_index - 5);
    fr_table_put(table, fr2_hash_at(input_bytes, 0, shift, min_match), ip_index - 5);
    fr_table_put(table, fr2_hash_at(input_bytes, 1, shift, min_match), ip_index - 4);
    fr_table_put(table, fr2_hash_at(input_bytes, 2, shift, min_match), ip_index - 3);
    input_bytes = br_load64(base_ip + ip_index - 2);
    cur_hash = fr2_hash_at(input_bytes, 2, shift, min_match);
    fr_table_put(table, fr2_hash_at(input_bytes, 0, shift, min_match), ip_index - 2);
    fr_table_put(table, fr2_hash_at(input_bytes, 1, shift, min_match), ip_index - 1);
  }
  cur_hash = BR_UNIFORM(cur_hash);
  const uint32_t candidate = fr_table_get(table, cur_hash);
  fr_table_put(table, cur_hash, ip_index);
  return candidate;
}

// CreateCommands, :157-385.  Returns the number of literals.
BR_DEV uint32_t fr2_create_commands(uint32_t input_index, uint32_t block_size, uint32_t input_size, const uint8_t* base_ip, uint32_t* table,
                                    uint32_t table_bits, uint32_t min_match, uint8_t* literals, FragmentCmds& cmds) {
  uint32_t ip_index = input_index;
  const uint32_t shift = 64 - table_bits;
  const uint32_t ip_end = input_index + block_size;
  uint32_t next_emit = input_index;
  uint32_t n_lits = 0;
  int32_t last_distance = -1;
  const uint32_t kInputMarginBytes = 16, kMaxDistance = (1u << 18) - 16;
  if (block_size >= kInputMarginBytes) {
    const uint32_t a = block_size - min_match, b = input_size - kInputMarginBytes;
    const uint32_t ip_limit = input_index + (a < b ? a : b);
    bool remainder = false;
    uint32_t next_hash = fr2_hash(base_ip + (++ip_index), shift, min_match);
    while (!remainder) {
      uint32_t skip = 32;
      uint32_t next_ip = ip_index;
      uint32_t candidate = 0;
      for (;;) {
        for (;;) {
          const uint32_t hash = next_hash;
          const uint32_t between = skip >> 5;
          ++skip;
          ip_index = next_ip;
          next_ip = ip_index + between;
          if (next_ip > ip_limit) {
            remainder = true;
            break;
          }
          next_hash = fr2_hash(base_ip + next_ip, shift, min_match);
          // candidate = ip - last_distance: with last_distance == -1 it lies one byte ahead and is not a candidate
          if (last_distance > 0 && (uint32_t)last_distance <= ip_index && fr2_is_match(base_ip + ip_index, base_ip + (ip_index - (uint32_t)last_distance), min_match)) {
            candidate = ip_index - (uint32_t)last_distance;
            fr_table_put(table, hash, ip_index);
            break;
          }
          candidate = fr_table_get(table, hash);
          fr_table_put(table, hash, ip_index);
          if (fr2_is_match(base_ip + ip_index, base_ip + candidate, min_match)) break;
        }
        if (!(ip_index - candidate > kMaxDistance && !remainder)) break;
      }
      if (remainder) break;
      {
        const uint32_t base = ip_index;
        const uint32_t matched = min_match + fr_match_length(base_ip + candidate + min_match, base_ip + ip_index + min_match, ip_end - ip_index - min_match);
        const int32_t distance = (int32_t)(base - candidate);
        const uint32_t insert = base - next_emit;
        ip_index += matched;
        cmds.push(fr2_insert_len(insert));
        fr_copy_literals(literals + n_lits, base_ip + next_emit, insert);
        n_lits += insert;
        if (distance == last_distance) {
          cmds.push(64);
        } else {
          cmds.push(fr2_distance((uint32_t)distance));
          last_distance = distance;
        }
        fr2_copy_len_last_distance(matched, cmds);
        next_emit = ip_index;
        if (ip_index >= ip_limit) {
          remainder = true;
          break;
        }
        candidate = fr2_after_copy(base_ip, ip_index, table, shift, min_match, true);
      }
      while (ip_index - candidate <= kMaxDistance && fr2_is_match(base_ip + ip_index, base_ip + candidate, min_match)) {
        const uint32_t base_index = ip_index;
        const uint32_t matched = min_match + fr_match_length(base_ip + candidate + min_match, base_ip + ip_index + min_match, ip_end - ip_index - min_match);
        ip_index += matched;
        last_distance = (int32_t)(base_index - candidate);
        fr2_copy_len(matched, cmds);
        cmds.push(fr2_distance((uint32_t)last_distance));
        next_emit = ip_index;
        if (ip_index >= ip_limit) {
          remainder = true;
          break;
        }
        candidate = fr2_after_copy(base_ip, ip_index, table, shift, min_match, false);
      }
      if (!remainder) next_hash = fr2_hash(base_ip + (++ip_index), shift, min_match);
    }
  }
  if (next_emit < ip_end) {
    const uint32_t insert = ip_end - next_emit;
    cmds.push(fr2_insert_len(insert));
    fr_copy_literals(literals + n_lits, base_ip + next_emit, insert);
    n_lits += insert;
  }
  return n_lits;
}

// ShouldCompress, :387-406
BR_DEV bool fr2_should_compress(const EntropyTables& et, const uint8_t* input, uint32_t input_size, uint32_t num_literals, FragmentScratch& S) {
  const float corpus_size = (float)input_size;
  if ((float)num_literals < 0.98f * corpus_size) return true;
  for (uint32_t i = 0; i < 256; ++i) S.sample_histo[i] = 0;
  const float max_total_bit_cost = corpus_size * 8.0f * 0.98f / 43.0f;
  for (uint32_t i = 0; i < input_size; i += 43) S.sample_histo[input[i]]++;
  return br_bits_entropy(et, S.sample_histo, 256) < max_total_bit_cost;
}

// BuildAndStoreCommandPrefixCode: compress_fragment_two_pass.rs:449-517 (two_pass) and compress_fragment.rs:577-648 -- the same
// construction with the 64 insert-and-copy codes arranged in another order
template <typename Sink>
BR_DEV void fr_command_prefix_code(const uint32_t* histogram, uint8_t* depth, uint16_t* bits, bool two_pass, FragmentScratch& S, Sink& sink) {
  uint8_t* cmd_depth = S.wide_depth;
  uint16_t* cmd_bits = S.wide_bits;
  for (uint32_t i = 0; i < 704; ++i) cmd_depth[i] = 0;
  for (uint32_t i = 0; i < 64; ++i) cmd_bits[i] = 0;
  for (uint32_t i = 0; i < 128; ++i) depth[i] = 0;
  br_create_huffman_tree(histogram, 64, 15, S.huff.tree, depth);
  br_create_huffman_tree(histogram + 64, 64, 14, S.huff.tree, depth + 64);
  // (from, to, count) of the rearrangements
  const uint8_t in2[6][3] = {{24, 0, 24}, {0, 24, 8}, {48, 32, 8}, {8, 40, 8}, {56, 48, 8}, {16, 56, 8}};
  const uint8_t in0[6][3] = {{0, 0, 24}, {40, 24, 8}, {24, 32, 8}, {48, 40, 8}, {32, 48, 8}, {56, 56, 8}};
  for (int r = 0; r < 6; ++r) {
    const uint8_t* m = two_pass ? in2[r] : in0[r];
    for (uint32_t i = 0; i < m[2]; ++i) cmd_depth[m[1] + i] = depth[m[0] + i];
  }
  br_convert_bit_depths_to_symbols(cmd_depth, 64, cmd_bits);
  const uint8_t out2[6][3] = {{24, 0, 16}, {40, 8, 8}, {56, 16, 8}, {0, 24, 48}, {32, 48, 8}, {48, 56, 8}};
  const uint8_t out0[6][3] = {{0, 0, 24}, {32, 24, 8}, {48, 32, 8}, {24, 40, 8}, {40, 48, 8}, {56, 56, 8}};
  for (int r = 0; r < 6; ++r) {
    const uint8_t* m = two_pass ? out2[r] : out0[r];
    for (uint32_t i = 0; i < m[2]; ++i) bits[m[1] + i] = cmd_bits[m[0] + i];
  }
  br_convert_bit_depths_to_symbols(depth + 64, 64, bits + 64);
  for (uint32_t i = 0; i < 64; ++i) cmd_depth[i] = 0;
  if (two_pass) {
    for (uint32_t i = 0; i < 8; ++i) {
      cmd_depth[i] = depth[24 + i];
      cmd_depth[64 + i] = depth[32 + i];
      cmd_depth[128 + i] = depth[40 + i];
      cmd_depth[192 + i] = depth[48 + i];
      cmd_depth[384 + i] = depth[56 + i];
    }
    for (uint32_t i = 0; i < 8; ++i) {
      cmd_depth[128 + 8 * i] = depth[i];
      cmd_depth[256 + 8 * i] = depth[i + 8];
      cmd_depth[448 + 8 * i] = depth[i + 16];
    }
  } else {
    for (uint32_t i = 0; i < 8; ++i) {
      cmd_depth[i] = depth[i];
      cmd_depth[64 + i] = depth[8 + i];
      cmd_depth[128 + i] = depth[16 + i];
      cmd_depth[192 + i] = depth[24 + i];
      cmd_depth[384 + i] = depth[32 + i];
    }
    for (uint32_t i = 0; i < 8; ++i) {
      cmd_depth[128 + 8 * i] = depth[i + 40];
      cmd_depth[256 + 8 * i] = depth[i + 48];
      cmd_depth[448 + 8 * i] = depth[i + 56];
    }
  }
  sink.tree(cmd_depth, 704, S);
  sink.tree(depth + 64, 64, S);
}
struct FragmentTreeToStream {
  FragmentOut* o;
  BR_DEV void tree(const uint8_t* depths, uint32_t num, FragmentScratch& S) { fr_store_huffman_tree(depths, num, S, *o); }
};

// StoreCommands, :519-629
BR_DEV void fr2_store_commands(const uint8_t* literals, uint32_t num_literals, const uint32_t* commands, uint32_t num_commands, FragmentScratch& S, FragmentOut& o) {
  const uint8_t kNumExtraBits[128] = {
      0,  0,  0,  0,  0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  5,  5,  6,  7,  8,  9,  10, 12, 14, 24, 0,  0,
      0,  0,  0,  0,  0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  0,  0,  0,  0,  0,  0,  0,  0,  1,  1,  2,  2,
      3,  3,  4,  4,  5,  5,  6,  7,  8,  9,  10, 24, 0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
      0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  5,  5,  6,  6,  7,  7,  8,  8,  9,  9,  10, 10, 11, 11, 12, 12,
      13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 24, 24};
  const uint32_t kInsertOffset[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
  for (uint32_t i = 0; i < 256; ++i) {
    S.lit_histo[i] = 0;
    S.lit_depth[i] = 0;
    S.lit_bits[i] = 0;
  }
  for (uint32_t i = 0; i < 128; ++i) {
    S.cmd_histo[i] = 0;
    S.cmd_depth[i] = 0;
    S.cmd_bits[i] = 0;
  }
  for (uint32_t i = 0; i < num_literals; ++i) S.lit_histo[literals[i]]++;
  fr_huffman_fast(S.lit_histo, num_literals, 8, S.lit_depth, S.lit_bits, S, o);
  for (uint32_t i = 0; i < num_commands; ++i) S.cmd_histo[commands[i] & 0xff]++;
  S.cmd_histo[1] += 1;
  S.cmd_histo[2] += 1;
  S.cmd_histo[64] += 1;
  S.cmd_histo[84] += 1;
  FragmentTreeToStream sink{&o};
  fr_command_prefix_code(S.cmd_histo, S.cmd_depth, S.cmd_bits, true, S, sink);
  uint32_t lit = 0;
  for (uint32_t i = 0; i < num_commands; ++i) {
    const uint32_t cmd = commands[i];
    const uint32_t code = cmd & 0xff;
    const uint32_t extra = cmd >> 8;
    o.put(S.cmd_depth[code], S.cmd_bits[code]);
    o.put(kNumExtraBits[code], extra);
    if (code < 24) {
      const uint32_t insert = kInsertOffset[code] + extra;
      for (uint32_t j = 0; j < insert; ++j) {
        const uint32_t b = literals[lit + j];
        o.put(S.lit_depth[b], S.lit_bits[b]);
      }
      lit += insert;
    }
  }
}

// the one decision of a fragment that counts OUTPUT bits (and so sees the padding of an alignment in front of it)
struct FragmentDecision {
  uint64_t bits, align;
  uint32_t fell_back;
};

// compress_fragment_two_pass, :646-703 + 752-905
BR_DEV void fr2_compress(const EntropyTables& et, const uint8_t* input, uint32_t input_size, bool is_last, uint32_t table_bits, const FragmentBuffers& B,
                         FragmentScratch& S, FragmentOut& o, FragmentDecision& dec) {
  const uint64_t initial = o.pos;
  if (table_bits >= 8 && table_bits <= 17) {
    const uint32_t min_match = table_bits < 15 ? 4 : 6;
    uint32_t input_index = 0, remaining = input_size;
    while (remaining > 0) {
      const uint32_t block_size = remaining < (1u << 17) ? remaining : (1u << 17);
      FragmentCmds cmds;
      cmds.at = B.commands;
      cmds.n = 0;
      const uint32_t num_literals = fr2_create_commands(input_index, block_size, remaining, input, B.table, table_bits, min_match, B.literals, cmds);
      FR_FENCE();  // (what lane 0 wrote is read back by all lanes below)
      if (fr2_should_compress(et, input + input_index, block_size, num_literals, S)) {
        fr_store_meta_block_header(block_size, false, o);
        o.put(13, 0);
        fr2_store_commands(B.literals, num_literals, B.commands, cmds.n, S, o);
      } else {
        fr_emit_uncompressed(input + input_index, block_size, o);
      }
      FR_FENCE();
      input_index += block_size;
      remaining -= block_size;
    }
  }
  dec.bits = o.pos - initial;
  dec.align = o.first_align;
  dec.fell_back = 0;
  if (o.pos - initial > 31 + ((uint64_t)input_size << 3)) {
    dec.fell_back = 1;
    o.rewind(initial);
    fr_emit_uncompressed(input, input_size, o);
  }
  if (is_last) {
    o.put(1, 1);
    o.put(1, 1);
    o.align();
  }
}

// ================================================================================================== quality 0
BR_DEV uint32_t fr0_hash_at(uint64_t v, uint32_t offset, uint32_t shift) { return (uint32_t)((((v >> (8 * offset)) << 24) * (uint64_t)0x1e35a7bdu) >> shift); }
BR_DEV uint32_t fr0_hash(const uint8_t* p, uint32_t shift) { return BR_UNIFORM(fr0_hash_at(br_load64(p), 0, shift)); }
BR_DEV bool fr0_is_match(const uint8_t* p1, const uint8_t* p2) { return br_load32(p1) == br_load32(p2) && p1[4] == p2[4]; }

// the command code of the running meta-block: depths, codes and the histogram of what has been emitted with them
struct Fragment0Code {
  uint8_t* depth;
  uint16_t* bits;
  uint32_t* histo;
  BR_DEV void emit(uint32_t code, FragmentOut& o) {
    o.put(depth[code], bits[code]);
    ++histo[code];
  }
};
BR_DEV void fr0_insert_len(uint32_t insertlen, Fragment0Code& c, FragmentOut& o) {  // EmitInsertLen :133-213
  if (insertlen < 6) {
    c.emit(insertlen + 40, o);
  } else if (insertlen < 130) {
    const uint32_t tail = insertlen - 2;
    const uint32_t nbits = br_log2_floor_nonzero(tail) - 1;
    const uint32_t prefix = tail >> nbits;
    c.emit((nbits << 1) + prefix + 42, o);
    o.put(nbits, tail - (prefix << nbits));
  } else if (insertlen < 2114) {
    const uint32_t tail = insertlen - 66;
    const uint32_t nbits = br_log2_floor_nonzero(tail);
    c.emit(nbits + 50, o);
    o.put(nbits, tail - (1u << nbits));
  } else {
    c.emit(61, o);
    o.put(12, insertlen - 2114);
  }
}
BR_DEV void fr0_long_insert_len(uint32_t insertlen, Fragment0Code& c, FragmentOut& o) {  // EmitLongInsertLen :251-286
  if (insertlen < 22594) {
    c.emit(62, o);
    o.put(14, insertlen - 6210);
  } else {
    c.emit(63, o);
    o.put(24, insertlen - 22594);
  }
}
BR_DEV void fr0_literals(const uint8_t* input, uint32_t len, const FragmentScratch& S, FragmentOut& o) {  // EmitLiterals :288-305
  for (uint32_t j = 0; j < len; ++j) {
    const uint32_t b = input[j];
    o.put(S.lit_depth[b], S.lit_bits[b]);
  }
}
BR_DEV void fr0_distance(uint32_t distance, Fragment0Code& c, FragmentOut& o) {  // EmitDistance :307-334
  const uint32_t d = distance + 3;
  const uint32_t nbits = br_log2_floor_nonzero(d) - 1;
  const uint32_t prefix = (d >> nbits) & 1;
  const uint32_t offset = (2 + prefix) << nbits;
  c.emit(2 * (nbits - 1) + prefix + 80, o);
  o.put(nbits, d - offset);
}
BR_DEV void fr0_copy_len_last_distance(uint32_t copylen, Fragment0Code& c, FragmentOut& o) {  // EmitCopyLenLastDistance :336-446
  if (copylen < 12) {
    c.emit(copylen - 4, o);
  } else if (copylen < 72) {
    const uint32_t tail = copylen - 8;
    const uint32_t nbits = br_log2_floor_nonzero(tail) - 1;
    const uint32_t prefix = tail >> nbits;
    c.emit((nbits << 1) + prefix + 4, o);
    o.put(nbits, tail - (prefix << nbits));
  } else if (copylen < 136) {
    const uint32_t tail = copylen - 8;
    const uint32_t code = (tail >> 5) + 30;
    o.put(c.depth[code], c.bits[code]);
    o.put(5, tail & 31);
    o.put(c.depth[64], c.bits[64]);
    ++c.histo[code];
    ++c.histo[64];
  } else if (copylen < 2120) {
    const uint32_t tail = copylen - 72;
    const uint32_t nbits = br_log2_floor_nonzero(tail);
    const uint32_t code = nbits + 28;
    o.put(c.depth[code], c.bits[code]);
    o.put(nbits, tail - (1u << nbits));
    o.put(c.depth[64], c.bits[64]);
    ++c.histo[code];
    ++c.histo[64];
  } else {
    o.put(c.depth[39], c.bits[39]);
    o.put(24, copylen - 2120);
    o.put(c.depth[64], c.bits[64]);
    ++c.histo[39];
    ++c.histo[64];
  }
}
BR_DEV void fr0_copy_len(uint32_t copylen, Fragment0Code& c, FragmentOut& o) {  // EmitCopyLen :453-532
  if (copylen < 10) {
    c.emit(copylen + 14, o);
  } else if (copylen < 134) {
    const uint32_t tail = copylen - 6;
    const uint32_t nbits = br_log2_floor_nonzero(tail) - 1;
    const uint32_t prefix = tail >> nbits;
    c.emit((nbits << 1) + prefix + 20, o);
    o.put(nbits, tail - (prefix << nbits));
  } else if (copylen < 2118) {
    const uint32_t tail = copylen - 70;
    const uint32_t nbits = br_log2_floor_nonzero(tail);
    c.emit(nbits + 28, o);
    o.put(nbits, tail - (1u << nbits));
  } else {
    c.emit(39, o);
    o.put(24, copylen - 2118);
  }
}

// BuildAndStoreLiteralPrefixCode, :41-125.  Returns literal_ratio.
BR_DEV uint32_t fr0_literal_prefix_code(const uint8_t* input, uint32_t input_size, FragmentScratch& S, FragmentOut& o) {
  uint32_t* histogram = S.lit_histo;
  uint32_t histogram_total;
  for (uint32_t i = 0; i < 256; ++i) {
    histogram[i] = 0;
    S.lit_depth[i] = 0;
    S.lit_bits[i] = 0;
  }
  if (input_size < (1u << 15)) {
    for (uint32_t i = 0; i < input_size; ++i) histogram[input[i]]++;
    histogram_total = input_size;
    for (uint32_t i = 0; i < 256; ++i) {
      const uint32_t adjust = 2 * (histogram[i] < 11u ? histogram[i] : 11u);
      histogram[i] += adjust;
      histogram_total += adjust;
    }
  } else {
    const uint32_t kSampleRate = 29;
    for (uint32_t i = 0; i < input_size; i += kSampleRate) histogram[input[i]]++;
    histogram_total = (input_size + kSampleRate - 1) / kSampleRate;
    for (uint32_t i = 0; i < 256; ++i) {
      const uint32_t adjust = 1 + 2 * (histogram[i] < 11u ? histogram[i] : 11u);
      histogram[i] += adjust;
      histogram_total += adjust;
    }
  }
  fr_huffman_fast(histogram, histogram_total, 8, S.lit_depth, S.lit_bits, S, o);
  uint64_t literal_ratio = 0;
  for (uint32_t i = 0; i < 256; ++i)
    if (histogram[i] != 0) literal_ratio += (uint64_t)(uint32_t)(histogram[i] * (uint32_t)S.lit_depth[i]);
  return (uint32_t)(literal_ratio * 125 / histogram_total);
}

// ShouldMergeBlock, :534-558 (f32 in the reference's order)
BR_DEV bool fr0_should_merge_block(const EntropyTables& et, const uint8_t* data, uint32_t len, FragmentScratch& S) {
  uint32_t* histo = S.sample_histo;
  for (uint32_t i = 0; i < 256; ++i) histo[i] = 0;
  const uint32_t kSampleRate = 43;
  for (uint32_t i = 0; i < len; i += kSampleRate) ++histo[data[i]];
  const uint32_t total = (len + kSampleRate - 1) / kSampleRate;
  float r = (br_fast_log2(et, total) + 0.5f) * (float)total + 200.0f;
  for (uint32_t i = 0; i < 256; ++i) r -= (float)histo[i] * ((float)S.lit_depth[i] + br_fast_log2(et, histo[i]));
  return r >= 0.0f;
}
BR_DEV bool fr0_should_use_uncompressed_mode(uint32_t compressed, uint32_t insertlen, uint32_t literal_ratio) {  // :215-224
  if ((uint64_t)compressed * 50 > insertlen) return false;
  return literal_ratio > 980;
}
// EmitUncompressedMetaBlock, :236-249: back to where the meta-block began
BR_DEV void fr0_emit_uncompressed(const uint8_t* begin, uint32_t len, uint64_t storage_ix_start, FragmentOut& o) {
  o.rewind(storage_ix_start);
  fr_emit_uncompressed(begin, len, o);
}
struct FragmentTreeToCode {  // the serialised command code kept for the next fragment (cmd_code / cmd_code_numbits)
  BitSink k;
  BR_DEV void tree(const uint8_t* depths, uint32_t num, FragmentScratch& S) { br_store_huffman_tree(depths, num, &S.huff, k); }
};

// the table updates behind a copy, :868-878 and :905-915
BR_DEV uint32_t fr0_after_copy(const uint8_t* input, uint32_t ip_index, uint32_t* table, uint32_t shift) {
  const uint64_t input_bytes = br_load64(input + ip_index - 3);
  const uint32_t cur_hash = BR_UNIFORM(fr0_hash_at(input_bytes, 3, shift));
  fr_table_put(table, fr0_hash_at(input_bytes, 0, shift), ip_index - 3);
  fr_table_put(table, fr0_hash_at(input_bytes, 1, shift), ip_index - 2);
  fr_table_put(table, fr0_hash_at(input_bytes, 2, shift), ip_index - 1);
  const uint32_t candidate = fr_table_get(table, cur_hash);
  fr_table_put(table, cur_hash, ip_index);
  return candidate;
}

// compress_fragment_fast_impl, :650-1045.  cmd_code (words) / cmd_code_numbits: the command code in serialised form, in and out.
BR_DEV void fr0_compress_impl(const EntropyTables& et, const uint8_t* input_ptr, uint32_t input_size, bool is_last, uint32_t* table, uint32_t table_bits,
                              FragmentScratch& S, uint64_t* cmd_code, uint32_t* cmd_code_numbits, FragmentOut& o) {
  const uint32_t kCmdHistoSeed[128] = {
      0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
      1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
      1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
      1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0};
  enum { kEmitRemainder, kEmitCommands, kNextBlock };
  Fragment0Code code;
  code.depth = S.cmd_depth;
  code.bits = S.cmd_bits;
  code.histo = S.cmd_histo;
  uint32_t ip_end = 0, next_emit = 0;
  const uint32_t kFirstBlockSize = 3u << 15, kMergeBlockSize = 1u << 16;
  const uint32_t kInputMarginBytes = 16, kMinMatchLen = 5;
  const uint32_t kMaxDistance = (1u << 18) - 16;
  uint32_t metablock_start = 0;
  uint32_t block_size = input_size < kFirstBlockSize ? input_size : kFirstBlockSize;
  uint32_t total_block_size = block_size;
  uint64_t mlen_storage_ix = o.pos + 3;
  uint32_t literal_ratio;
  uint32_t input_index = 0;
  int32_t last_distance = -1;
  const uint32_t shift = 64 - table_bits;
  for (uint32_t i = 0; i < 128; ++i) S.cmd_histo[i] = 0;
  fr_store_meta_block_header(block_size, false, o);
  o.put(13, 0);
  literal_ratio = fr0_literal_prefix_code(input_ptr + input_index, block_size, S, o);
  fr_append(o, cmd_code, *cmd_code_numbits);
  int state = kEmitCommands;
  for (;;) {
    if (state == kEmitCommands) {
      uint32_t ip_index = input_index;
      for (uint32_t i = 0; i < 128; ++i) S.cmd_histo[i] = kCmdHistoSeed[i];
      last_distance = -1;
      ip_end = input_index + block_size;
      if (block_size >= kInputMarginBytes) {
        const uint32_t a = block_size - kMinMatchLen, b = input_size - kInputMarginBytes;
        const uint32_t ip_limit = input_index + (a < b ? a : b);
        uint32_t next_hash = fr0_hash(input_ptr + (++ip_index), shift);
        bool restart_outer = false;
        for (;;) {
          uint32_t skip = 32;
          uint32_t next_ip = ip_index;
          uint32_t candidate = 0;
          for (;;) {
            for (;;) {
              const uint32_t hash = next_hash;
              const uint32_t between = skip >> 5;
              ++skip;
              ip_index = next_ip;
              next_ip = ip_index + between;
              if (next_ip > ip_limit) {
                state = kEmitRemainder;
                break;
              }
              next_hash = fr0_hash(input_ptr + next_ip, shift);
              if (last_distance > 0 && (uint32_t)last_distance <= ip_index && fr0_is_match(input_ptr + ip_index, input_ptr + (ip_index - (uint32_t)last_distance))) {
                candidate = ip_index - (uint32_t)last_distance;
                fr_table_put(table, hash, ip_index);
                break;
              }
              candidate = fr_table_get(table, hash);
              fr_table_put(table, hash, ip_index);
              if (fr0_is_match(input_ptr + ip_index, input_ptr + candidate)) break;
            }
            if (!(ip_index - candidate > kMaxDistance && state == kEmitCommands)) break;
          }
          if (state != kEmitCommands) break;
          {
            const uint32_t base = ip_index;
            const uint32_t matched = 5 + fr_match_length(input_ptr + candidate + 5, input_ptr + ip_index + 5, ip_end - ip_index - 5);
            const int32_t distance = (int32_t)(base - candidate);
            const uint32_t insert = base - next_emit;
            ip_index += matched;
            if (insert < 6210) {
              fr0_insert_len(insert, code, o);
            } else if (fr0_should_use_uncompressed_mode(next_emit - metablock_start, insert, literal_ratio)) {
              fr0_emit_uncompressed(input_ptr + metablock_start, base - metablock_start, mlen_storage_ix - 3, o);
              input_size -= base - input_index;
              input_index = base;
              next_emit = input_index;
              state = kNextBlock;
              restart_outer = true;
              break;
            } else {
              fr0_long_insert_len(insert, code, o);
            }
            fr0_literals(input_ptr + next_emit, insert, S, o);
            if (distance == last_distance) {
              code.emit(64, o);
            } else {
              fr0_distance((uint32_t)distance, code, o);
              last_distance = distance;
            }
            fr0_copy_len_last_distance(matched, code, o);
            next_emit = ip_index;
            if (ip_index >= ip_limit) {
              state = kEmitRemainder;
              restart_outer = true;
              break;
            }
            candidate = fr0_after_copy(input_ptr, ip_index, table, shift);
            while (fr0_is_match(input_ptr + ip_index, input_ptr + candidate)) {
              const uint32_t base2 = ip_index;
              const uint32_t matched2 = 5 + fr_match_length(input_ptr + candidate + 5, input_ptr + ip_index + 5, ip_end - ip_index - 5);
              if (ip_index - candidate > kMaxDistance) break;
              ip_index += matched2;
              last_distance = (int32_t)(base2 - candidate);
              fr0_copy_len(matched2, code, o);
              fr0_distance((uint32_t)last_distance, code, o);
              next_emit = ip_index;
              if (ip_index >= ip_limit) {
                state = kEmitRemainder;
                restart_outer = true;
                break;
              }
              candidate = fr0_after_copy(input_ptr, ip_index, table, shift);
            }
            if (restart_outer) break;
            if (state == kEmitRemainder) break;
            if (state == kEmitCommands) next_hash = fr0_hash(input_ptr + (++ip_index), shift);
          }
        }
        if (restart_outer) continue;
      }
      state = kEmitRemainder;
      continue;
    } else if (state == kEmitRemainder) {
      input_index += block_size;
      input_size -= block_size;
      block_size = input_size < kMergeBlockSize ? input_size : kMergeBlockSize;
      if (input_size > 0 && total_block_size + block_size <= (1u << 20) && fr0_should_merge_block(et, input_ptr + input_index, block_size, S)) {
        total_block_size += block_size;
        o.update_bits(20, total_block_size - 1, mlen_storage_ix);
        state = kEmitCommands;
        continue;
      }
      if (next_emit < ip_end) {
        const uint32_t insert = ip_end - next_emit;
        if (insert < 6210) {
          fr0_insert_len(insert, code, o);
          fr0_literals(input_ptr + next_emit, insert, S, o);
        } else if (fr0_should_use_uncompressed_mode(next_emit - metablock_start, insert, literal_ratio)) {
          fr0_emit_uncompressed(input_ptr + metablock_start, ip_end - metablock_start, mlen_storage_ix - 3, o);
        } else {
          fr0_long_insert_len(insert, code, o);
          fr0_literals(input_ptr + next_emit, insert, S, o);
        }
      }
      next_emit = ip_end;
      state = kNextBlock;
      continue;
    } else {
      if (input_size > 0) {
        metablock_start = input_index;
        block_size = input_size < kFirstBlockSize ? input_size : kFirstBlockSize;
        total_block_size = block_size;
        mlen_storage_ix = o.pos + 3;
        fr_store_meta_block_header(block_size, false, o);
        o.put(13, 0);
        literal_ratio = fr0_literal_prefix_code(input_ptr + input_index, block_size, S, o);
        FragmentTreeToStream sink{&o};
        fr_command_prefix_code(S.cmd_histo, S.cmd_depth, S.cmd_bits, false, S, sink);
        state = kEmitCommands;
        continue;
      }
      break;
    }
  }
  if (!is_last) {
    // the code for the next fragment, from what this one's last meta-block emitted (:1033-1044)
    FragmentTreeToCode sink;
    for (uint32_t i = 0; i < kTreeBitsWords; ++i) cmd_code[i] = 0;
    sink.k.words = cmd_code;
    sink.k.pos = 0;
    fr_command_prefix_code(S.cmd_histo, S.cmd_depth, S.cmd_bits, false, S, sink);
    *cmd_code_numbits = (uint32_t)sink.k.pos;
  }
}

// compress_fragment_fast, :1089-1179
BR_DEV void fr0_compress(const EntropyTables& et, const uint8_t* input, uint32_t input_size, bool is_last, uint32_t table_bits, const FragmentBuffers& B,
                         FragmentScratch& S, uint64_t* cmd_code, uint32_t* cmd_code_numbits, FragmentOut& o, FragmentDecision& dec) {
  const uint64_t initial = o.pos;
  dec.bits = 0;
  dec.align = ~0ull;
  dec.fell_back = 0;
  if (input_size == 0) {
    o.put(1, 1);
    o.put(1, 1);
    o.align();
    return;
  }
  if (table_bits == 9 || table_bits == 11 || table_bits == 13 || table_bits == 15)
    fr0_compress_impl(et, input, input_size, is_last, B.table, table_bits, S, cmd_code, cmd_code_numbits, o);
  dec.bits = o.pos - initial;
  dec.align = o.first_align;
  if (o.pos - initial > 31 + ((uint64_t)input_size << 3)) {
    dec.fell_back = 1;
    fr0_emit_uncompressed(input, input_size, initial, o);
  }
  if (is_last) {
    o.put(1, 1);
    o.put(1, 1);
    o.align();
  }
}

// One fragment of a batch (the seam's frag_compress_batch): job j on slab j of B, bits into its own slot of `out`.
// cmd_code_words: kTreeBitsWords words of workgroup memory.
BR_DEV void br_fragment_job(int quality, const EntropyTables& et, const uint8_t* input_base, const FragmentJob& job, uint32_t j, const FragmentBuffers& slabs,
                            const FragmentState* states_in, FragmentState* states_out, FragmentResult* results, uint8_t* out_base, FragmentScratch& S,
                            uint64_t* cmd_code_words) {
  FragmentBuffers B;
  B.table = slabs.table + (size_t)j * slabs.table_stride;
  B.commands = slabs.commands ? slabs.commands + (size_t)j * slabs.cmd_stride : nullptr;
  B.literals = slabs.literals ? slabs.literals + (size_t)j * slabs.lit_stride : nullptr;
  const uint8_t* input = input_base + job.in_offset;
  uint8_t* out = out_base + job.out_offset;
  FragmentOut o;
  o.out = out;
  o.pos = job.start_bits;
  o.nacc = job.start_bits & 7u;
  o.acc = 0;  // (the slot starts with a zero byte: the bits in front of start_bits belong to whoever joins the slots)
  o.first_align = ~0ull;
  FragmentDecision dec;
  dec.bits = 0;
  dec.align = ~0ull;
  dec.fell_back = 0;
  uint32_t numbits = 0;
  if (quality == 0) {
    const FragmentState* st = states_in + job.state_in;
    numbits = st->cmd_code_numbits;
    for (uint32_t i = 0; i < 128; ++i) {
      S.cmd_depth[i] = st->cmd_depths[i];
      S.cmd_bits[i] = st->cmd_bits[i];
    }
    for (uint32_t i = 0; i < kTreeBitsWords; ++i) {
      uint64_t w = 0;
      for (uint32_t b = 0; b < 8; ++b) w |= (uint64_t)st->cmd_code[8 * i + b] << (8 * b);
      cmd_code_words[i] = w;
    }
    fr0_compress(et, input, job.in_size, job.is_last != 0, job.table_bits, B, S, cmd_code_words, &numbits, o, dec);
  } else {
    fr2_compress(et, input, job.in_size, job.is_last != 0, job.table_bits, B, S, o, dec);
  }
  o.park();
  FR_FENCE();
  if (BR_LANE == 0) {
    FragmentResult r;
    r.end_bits = o.pos;
    r.first_align = o.first_align;
    r.decision_bits = dec.bits;
    r.decision_align = dec.align;
    r.fell_back = dec.fell_back;
    r.bad = 0;
    results[j] = r;
    if (quality == 0 && states_out != nullptr) {
      FragmentState* st = states_out + j;
      st->cmd_code_numbits = numbits;
      st->pad = 0;
      for (uint32_t i = 0; i < 128; ++i) {
        st->cmd_depths[i] = S.cmd_depth[i];
        st->cmd_bits[i] = S.cmd_bits[i];
      }
      for (uint32_t i = 0; i < 512; ++i) st->cmd_code[i] = (uint8_t)(cmd_code_words[i >> 3] >> (8 * (i & 7)));
    }
  }
}

}  // namespace brotli_mi355x
#endif
