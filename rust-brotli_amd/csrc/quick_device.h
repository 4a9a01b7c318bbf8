// quick_device.h -- qualities 2, 3 and 4 of the backward-reference stage as device code: the BasicHasher family
// under the greedy / lazy parse.  Shared by the gfx950 kernel (quick_kernels.hip) and the host emulation of the device seam
// (tests/emu, test infrastructure).
//
// What it replaces, per input block (encode.rs:2417-2453):
//   StitchToPreviousBlock                          mod.rs:210-222
//   extend_last_command                            encode.rs:360-400
//   BrotliCreateBackwardReferences, quality < 5    mod.rs:2376-2552
// built from
//   BasicHasher::HashBytes / Store / StoreRange    mod.rs:437-441 (482-486, 502-506, 527-531), 322-327, 299-321
//   BasicHasher::FindLongestMatch                  mod.rs:359-473
//   SearchInStaticDictionary (shallow)             mod.rs:1942-1988, 1896-1940
//
// Why one wavefront per stream: a BasicHasher slot holds the LAST position filed under it, and which positions are filed
// hangs on the parse (a search files its own position, a copy its range, a literal spree every second or fourth one).  The
// state of the reference is a function of everything parsed before; the chain below walks it in order, on the reference's own
// table (in device memory, one per stream), and keeps every rule that shows in the output:
//   * slots are key + ((ix >> 3) % sweep); StoreRange files runs of >= 16 positions four at a time under the offset of the
//     quad's first position and as RING-BUFFER indices (ix & mask) -- entries that no search can reach once the stream has
//     passed one ring-buffer size;
//   * the last-distance candidate is taken unconditionally, a single-slot hasher (H2) does not even consult its slot after it;
//   * at quality < 5 the lazy search at position + 1 starts from best_len = len - 1 of the match it tries to beat
//     (mod.rs:2450-2454), i.e. its quick-reject byte sits where that match ends;
//   * the byte behind the end of the block (reachable by that quick reject) is what the reference's ring buffer holds there:
//     0 during the first lap, the byte of one lap earlier afterwards (br_unwritten_byte).
// Parallelism of such a call is the number of streams (BrotliEncoderCompressMulti shards, concurrent encoder instances).
#ifndef BROTLI_MI355X_QUICK_DEVICE_H_
#define BROTLI_MI355X_QUICK_DEVICE_H_

#include "lz77_chain.h"
#include "quick_api.h"

namespace brotli_mi355x {

// read-only tables (device memory; the host arrays in the emulation)
struct QuickTables {
  const uint16_t* dict_hash;               // kStaticDictionaryHash [32768]
  const uint8_t* dict_data;
  const uint32_t* dict_offsets_by_length;  // [32]
};

static constexpr uint64_t kQuickHashMul64 = 0x1e35a7bd1e35a7bdull;  // kHashMul64, mod.rs:56
static constexpr uint32_t kQuickMinScore = 30 * 8 * 8 + 100;        // mod.rs:2397
static constexpr uint32_t kQuickHtl = 8;                            // HashTypeLength == StoreLookahead of every BasicHasher

// the chain reads back what it wrote a moment ago: device-scope accesses, as for the live chains' rings (lz77_live.h)
BR_DEV uint32_t q_get(const QuickJob& J, uint32_t slot) { return BR_UNIFORM(BR_LIVE_LD32(J.table + slot)); }
BR_DEV void q_put(const QuickJob& J, uint32_t slot, uint32_t value) {
  if (BR_LANE == 0) BR_LIVE_ST32(J.table + slot, value);
}

// HashBytes, mod.rs:437-441: the low hash_len bytes of a 64-bit load, multiplied, top bucket_bits bits
BR_DEV uint32_t q_key(const QuickJob& J, const uint8_t* p) {
  const uint64_t v = (br_load64(p) << (64u - 8u * J.hash_len)) * kQuickHashMul64;
  return BR_UNIFORM((uint32_t)(v >> (64u - J.bucket_bits)));
}

// Store, mod.rs:322-327
BR_DEV void q_store(const QuickJob& J, const uint8_t* text, uint32_t ix) { q_put(J, q_key(J, text + ix) + ((ix >> 3) & (J.sweep - 1u)), ix); }

// StoreRange, mod.rs:299-321: runs of 16 and more go four at a time -- slot offset of the quad's first position, and what is
// filed is the position's ring-buffer index
BR_DEV void q_store_range(const QuickJob& J, const Lz77Params& P, const uint8_t* text, uint32_t ix_start, uint32_t ix_end) {
  uint32_t i = ix_start;
  if (ix_end >= ix_start + 16u) {
    const uint32_t chunk_count = (ix_end - ix_start) / 4u;
    for (uint32_t c = 0; c < chunk_count; ++c) {
      const uint32_t at = ix_start + c * 4u;
      const uint32_t p = at & P.ring_mask;
      const uint32_t off = (p >> 3) & (J.sweep - 1u);
      for (uint32_t b = 0; b < 4; ++b) q_put(J, q_key(J, text + at + b) + off, p + b);
    }
    i = ix_start + chunk_count * 4u;
  }
  for (; i < ix_end; ++i) q_store(J, text, i);
}

// BackwardReferenceScore / ...UsingLastDistance, mod.rs:1871-1889
BR_DEV uint32_t q_score(const Lz77Params& P, uint32_t len, uint32_t backward) {
  return 30u * 8u * 8u + (P.literal_byte_score >> 2) * len - 30u * br_log2_floor_nonzero(backward);
}
BR_DEV uint32_t q_score_last(const Lz77Params& P, uint32_t len) { return (P.literal_byte_score >> 2) * len + 30u * 8u * 8u + 15u; }

struct QuickResult {
  uint32_t len, len_x_code, distance, score;
};
struct QuickBooks {
  uint32_t lookups, matches;  // dict_num_lookups / dict_num_matches of the hasher
  uint32_t searches;
};

// the byte at text position `at` as the reference's ring buffer holds it while the block [.., blk_end) is being searched
BR_DEV uint8_t q_byte(const Lz77Params& P, const uint8_t* text, uint32_t at, uint32_t blk_end) {
  if (at < blk_end) return text[at];
  return at <= P.ring_mask ? (uint8_t)0 : text[at - (P.ring_mask + 1u)];
}
// FindMatchLengthWithLimitMin4, static_dict.rs:134-147 (limit > 4 here): the common prefix, 0 if shorter than 4
BR_DEV uint32_t q_match_min4(const uint8_t* text, uint32_t prev, uint32_t cur, uint32_t limit) {
  if (br_load32(text + prev) != br_load32(text + cur)) return 0;
  const uint32_t n = BR_UNIFORM(br_match_len_wide(text + prev, text + cur, limit));
  return n;
}
// fix_unbroken_len, mod.rs:42-54, on the ring-buffer index of the candidate
BR_DEV uint32_t q_fix_len(const Lz77Params& P, uint32_t unbroken, uint32_t prev) {
  const uint32_t brk = P.dict_break, pm = prev & P.ring_mask;
  if (brk != 0 && pm < brk && pm + unbroken > brk) return brk - pm;
  return unbroken;
}

// SearchInStaticDictionary with shallow = true + TestStaticDictionaryItem, mod.rs:1942-1988, 1896-1940
BR_DEV bool q_search_dictionary(const Lz77Params& P, const QuickTables& T, const uint8_t* text, uint32_t cur, uint32_t max_length,
                                uint32_t max_backward, QuickBooks& books, QuickResult& out) {
  if (books.matches < (books.lookups >> 7)) return false;
  const uint32_t key = ((br_load32(text + cur) * 0x1e35a7bdu) >> (32 - 14)) << 1;  // Hash14 << 1
  const uint32_t item = T.dict_hash[key];
  books.lookups++;
  if (item == 0) return false;
  const uint32_t len = item & 0x1f, dist = item >> 5;
  if (len > max_length) return false;
  const uint32_t matchlen = br_match_len(text + cur, T.dict_data + T.dict_offsets_by_length[len] + len * dist, len);
  if (matchlen + 10 <= len || matchlen == 0) return false;  // kCutoffTransformsCount
  const uint32_t cut = len - matchlen;
  const uint32_t transform_id = (cut << 2) + (uint32_t)((0x071b520ada2d3200ull >> (cut * 6)) & 0x3f);
  const uint32_t backward = max_backward + dist + 1 + (transform_id << br_dict_size_bits(len));
  if (backward > P.dist_max_distance) return false;
  const uint32_t score = q_score(P, matchlen, backward);
  if (score < out.score) return false;
  out.len = matchlen;
  out.len_x_code = len ^ matchlen;
  out.distance = backward;
  out.score = score;
  books.matches++;
  return true;
}

// BasicHasher::FindLongestMatch, mod.rs:359-473.  out.len on entry = best_len the search starts from; the text is flat, so a
// slot's entry is looked at only after it has proved to lie inside the window (the reference tests the byte first and the
// distance second; neither has a side effect).
BR_DEV bool q_find_longest_match(const QuickJob& J, const Lz77Params& P, const QuickTables& T, const uint8_t* text, int32_t dc0, uint32_t cur,
                                 uint32_t max_length, uint32_t max_backward, uint32_t blk_end, QuickBooks& books, QuickResult& out) {
  const uint32_t best_len_in = out.len;
  const uint32_t key = q_key(J, text + cur);
  uint8_t compare_char = q_byte(P, text, cur + best_len_in, blk_end);
  uint32_t best_score = out.score, best_len = best_len_in;
  bool found = false;
  books.searches++;
  out.len_x_code = 0;
  if (dc0 > 0 && (uint32_t)dc0 <= cur) {  // prev_ix < cur_ix in the reference's wrapping arithmetic
    const uint32_t prev = cur - (uint32_t)dc0;
    if (compare_char == text[prev + best_len]) {
      const uint32_t unbroken = q_match_min4(text, prev, cur, max_length);
      if (unbroken != 0) {
        const uint32_t len = q_fix_len(P, unbroken, prev);
        best_score = q_score_last(P, len);
        best_len = len;
        out.len = len;
        out.distance = (uint32_t)dc0;
        out.score = best_score;
        compare_char = q_byte(P, text, cur + best_len, blk_end);
        if (J.sweep == 1) {
          q_put(J, key, cur);
          return true;
        }
        found = true;
      }
    }
  }
  if (J.sweep == 1) {
    const uint32_t prev = q_get(J, key);
    q_put(J, key, cur);
    const uint32_t backward = cur - prev;
    if (backward == 0 || backward > max_backward) return false;
    if (compare_char != text[prev + best_len_in]) return false;
    const uint32_t unbroken = q_match_min4(text, prev, cur, max_length);
    if (unbroken != 0) {
      const uint32_t len = q_fix_len(P, unbroken, prev);
      out.len = len;
      out.distance = backward;
      out.score = q_score(P, len, backward);
      return true;
    }
  } else {
    for (uint32_t j = 0; j < J.sweep; ++j) {
      const uint32_t prev = q_get(J, key + j);
      const uint32_t backward = cur - prev;
      if (backward == 0 || backward > max_backward) continue;
      if (compare_char != text[prev + best_len]) continue;
      const uint32_t unbroken = q_match_min4(text, prev, cur, max_length);
      if (unbroken != 0) {
        const uint32_t len = q_fix_len(P, unbroken, prev);
        const uint32_t score = q_score(P, len, backward);
        if (best_score < score) {
          best_score = score;
          best_len = len;
          out.len = len;
          out.distance = backward;
          out.score = score;
          compare_char = q_byte(P, text, cur + best_len, blk_end);
          found = true;
        }
      }
    }
  }
  if (J.use_dictionary && !found) found = q_search_dictionary(P, T, text, cur, max_length, max_backward, books, out);
  q_put(J, key + ((cur >> 3) & (J.sweep - 1u)), cur);
  return found;
}

// HasherPrependCustomDictionary, encode.rs:1163-1194 + mod.rs:224-229 (StoreLookaheadThenStore): every dictionary position
// but the last StoreLookahead - 1
BR_DEV void br_quick_prepend(const QuickJob& J, const uint8_t* text, uint32_t dict_bytes) {
  const uint32_t overlap = kQuickHtl - 1u;
  if (dict_bytes > overlap)
    for (uint32_t i = 0; i < dict_bytes - overlap; ++i) q_store(J, text, i);
}

// One input block.  The commands go to the slab as the gather pass wants them (br_raw_command; the first one with its LOCAL
// literals only: the resolver adds what was pending at the entry of the block).
BR_DEV void br_quick_block(const QuickJob& J, const Lz77Params& P, const QuickTables& T, const uint8_t* text, const Segment& seg,
                           const SegEntry& entry, Command* slab, SegExit* exit_out) {
  uint32_t position = seg.blk_start;
  uint32_t num_bytes = seg.blk_end - seg.blk_start;
  const uint32_t pos_end = seg.blk_end;
  QuickBooks books;
  books.lookups = BR_UNIFORM(BR_LIVE_LD32(J.table + quick_books_at(J)));
  books.matches = BR_UNIFORM(BR_LIVE_LD32(J.table + quick_books_at(J) + 1));
  books.searches = 0;
  // StitchToPreviousBlock, mod.rs:210-222
  if (num_bytes >= kQuickHtl - 1u && position >= 3) {
    q_store(J, text, position - 3);
    q_store(J, text, position - 2);
    q_store(J, text, position - 1);
  }
  // extend_last_command, encode.rs:360-400 (the resolver has checked everything but the bytes)
  uint32_t ext_len = 0;
  if (entry.ext_allowed) {
    const uint32_t d = (uint32_t)entry.cache[0];
    while (num_bytes != 0 && text[position] == text[position - d]) {
      ++ext_len;
      ++position;
      --num_bytes;
    }
  }
  int32_t dist_cache[4];
  for (int i = 0; i < 4; ++i) dist_cache[i] = entry.cache[i];
  // CreateBackwardReferences, mod.rs:2376-2552 (prepare_distance_cache is a no-op for a BasicHasher, mod.rs:297-298)
  const uint32_t store_end = num_bytes >= kQuickHtl ? position + num_bytes - kQuickHtl + 1u : position;
  const uint32_t window = P.spree_window;
  uint32_t apply_random_heuristics = position + window;
  uint32_t insert_length = 0;  // local literals (the reference starts from *last_insert_len)
  uint32_t n_cmds = 0, n_lits = 0, n_bad = 0, n_pushes = 0, last_dist_code = 0xffffffffu, last_copy_len = 0;
  while (position + kQuickHtl < pos_end) {
    uint32_t max_length = pos_end - position;
    uint32_t max_distance = position < P.max_backward_limit ? position : P.max_backward_limit;
    QuickResult sr;
    sr.len = 0;
    sr.len_x_code = 0;
    sr.distance = 0;
    sr.score = kQuickMinScore;
    if (q_find_longest_match(J, P, T, text, dist_cache[0], position, max_length, max_distance, pos_end, books, sr)) {
      uint32_t delayed = 0;
      max_length--;
      for (;;) {
        QuickResult sr2;
        sr2.len = sr.len - 1u < max_length ? sr.len - 1u : max_length;  // quality < 5, mod.rs:2450-2454
        sr2.len_x_code = 0;
        sr2.distance = 0;
        sr2.score = kQuickMinScore;
        max_distance = position + 1u < P.max_backward_limit ? position + 1u : P.max_backward_limit;
        const bool is_match_found = q_find_longest_match(J, P, T, text, dist_cache[0], position + 1u, max_length, max_distance, pos_end, books, sr2);
        if (is_match_found && sr2.score >= sr.score + 175u) {  // cost_diff_lazy
          position++;
          insert_length++;
          sr = sr2;
          if (++delayed < 4 && position + kQuickHtl < pos_end) {
            max_length--;
            continue;
          }
        }
        break;
      }
      apply_random_heuristics = position + 2u * sr.len + window;
      max_distance = position < P.max_backward_limit ? position : P.max_backward_limit;
      const uint32_t distance_code = br_compute_distance_code(sr.distance, max_distance, dist_cache);
      if (sr.distance <= max_distance && distance_code > 0) {
        dist_cache[3] = dist_cache[2];
        dist_cache[2] = dist_cache[1];
        dist_cache[1] = dist_cache[0];
        dist_cache[0] = (int32_t)sr.distance;
        n_pushes++;
      }
      if (BR_LANE == 0) slab[n_cmds] = br_raw_command(insert_length, sr.len, sr.len ^ sr.len_x_code, distance_code);
      if (sr.len < 2) n_bad++;
      ++n_cmds;
      n_lits += insert_length;
      last_dist_code = distance_code;
      last_copy_len = sr.len;
      insert_length = 0;
      {
        const uint32_t a = position + 2u, b = position + sr.len < store_end ? position + sr.len : store_end;
        q_store_range(J, P, text, a, b);
      }
      position += sr.len;
    } else {
      insert_length++;
      position++;
      if (position > apply_random_heuristics) {
        const uint32_t margin = kQuickHtl - 1u;  // max(StoreLookahead - 1, 4)
        if (position + 16u >= pos_end - margin) {
          insert_length += pos_end - position;
          position = pos_end;
        } else if (position > apply_random_heuristics + 4u * window) {
          for (uint32_t i = 0; i < 4; ++i) q_store(J, text, position + i * 4u);
          insert_length += 16u;
          position += 16u;
        } else {
          for (uint32_t i = 0; i < 4; ++i) q_store(J, text, position + i * 2u);
          insert_length += 8u;
          position += 8u;
        }
      }
    }
  }
  insert_length += pos_end - position;
  if (BR_LANE == 0) {
    BR_LIVE_ST32(J.table + quick_books_at(J), books.lookups);
    BR_LIVE_ST32(J.table + quick_books_at(J) + 1, books.matches);
    SegExit x;
    x.pos = pos_end;
    x.apply = 0;
    for (int i = 0; i < 4; ++i) x.cache[i] = dist_cache[i];
    x.insert_len = insert_length;
    x.n_cmds = n_cmds;
    x.n_lits = n_lits;
    x.ext_len = ext_len;
    x.dict_lookups = books.lookups;
    x.dict_matches = books.matches;
    x.last_dist_code = last_dist_code;
    x.bad_commands = n_bad;
    x.n_searches = books.searches;
    x.last_copy_len = last_copy_len;
    x.dict_mode = 0;
    x.dict_maxdef = 0;
    x.n_pushes = n_pushes < 4 ? n_pushes : 4;
    x.tail_kind = kHeadNone;
    x.tail_base = x.tail_p1 = 0;
    x.n_pushes_all = n_pushes;
    x.dict_entry_lookups = x.dict_entry_matches = 0;
    *exit_out = x;
  }
}

}  // namespace brotli_mi355x
#endif
