// lz77_lanes.h -- the plain quality-5 parse with ONE CHAIN PER LANE (round 5): round 0 and the warm-up.
//
// lz77_chain.h gives a chain a whole wavefront: the candidates of a search sit in the lanes, the parse state in scalar
// registers, and the greedy / lazy control flow of CreateBackwardReferences (backward_references/mod.rs:2376-2552) runs
// wave-uniform.  That is the right shape for the re-parse launches, which hold a few (thousand) chains and last as long as
// ONE chain does.  Round 0 and the warm-up are the opposite case: every segment of the input at once, 32 768 chains for
// 64 MiB -- four generations of waves whose ~300 wave-instructions per search (a third of them moves of spilled scalars) keep
// the vector pipes busy for 4 ms.  Here a lane IS a chain: 64 segments to a wavefront, the whole parse state in vector
// registers (no scalar pressure at all), every search candidate by candidate in the reference's order.  The loop is written
// as a state machine whose every step is exactly ONE FindLongestMatch (AdvHasher, mod.rs:1684-1812) followed by the transition
// it causes -- "fresh" search at the loop top, or the lazy probe of position + 1 (mod.rs:2440-2475) -- so that the 64 chains
// of a wave stay converged on the one piece of code that is expensive; what a search leads to (a command, a literal, a spree
// jump) is short predicated code behind it.  An earlier attempt (round 3, commit 838e49b) compiled the wave-uniform parse loop
// per lane as it stood -- nested loops, every lane somewhere else -- and lost a factor 3.5; the difference is this structure.
//
// Same inputs, same outputs as br_parse_segment<false, true> without a splice: commands, flags, exit record, checkpoints.
// Covered: the plain configuration (plain_q5_config: four cache candidates, 16-entry candidate rows, no custom-dictionary break,
// no hasher reset, no masked entries) without a run table.  Everything else keeps the wave-per-chain kernels.
// The host emulation compiles the same functions and calls them segment by segment (tests/emu/device_emu.cpp).
#ifndef BROTLI_MI355X_LZ77_LANES_H_
#define BROTLI_MI355X_LZ77_LANES_H_

#include "lz77_chain.h"

namespace brotli_mi355x {

#if defined(BROTLI_HOST_EMU)
#define LN_ANY(x) (x)
#define LN_UNROLL
#else
#define LN_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#define LN_UNROLL _Pragma("unroll")
#endif

struct Ln16 {
  uint32_t w[4];
};
BR_DEV Ln16 ln_load16(const uint8_t* p) {
  Ln16 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
BR_DEV void ln_store8(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
BR_DEV void ln_store4(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// common prefix of two 16-byte heads, 0..16
BR_DEV uint32_t ln_common16(const Ln16& a, const Ln16& b) {
  const uint32_t x0 = a.w[0] ^ b.w[0], x1 = a.w[1] ^ b.w[1], x2 = a.w[2] ^ b.w[2], x3 = a.w[3] ^ b.w[3];
  // (from the back: the first non-zero word wins)
  uint32_t n = 16;
  if (x3 != 0) n = 12 + ((uint32_t)__builtin_ctz(x3) >> 3);
  if (x2 != 0) n = 8 + ((uint32_t)__builtin_ctz(x2) >> 3);
  if (x1 != 0) n = 4 + ((uint32_t)__builtin_ctz(x1) >> 3);
  if (x0 != 0) n = (uint32_t)__builtin_ctz(x0) >> 3;
  return n;
}

// the tail of br_match_len_wide: both sides agree in their first 32 bytes
BR_DEV uint32_t ln_match_beyond32(const uint8_t* a, const uint8_t* b, uint32_t limit) {
  uint32_t i = 32;
  while (i + 32 <= limit) {
    const Ln16 c0 = ln_load16(a + i), c1 = ln_load16(a + i + 16), d0 = ln_load16(b + i), d1 = ln_load16(b + i + 16);
    if (ln_common16(c0, d0) != 16 || ln_common16(c1, d1) != 16) break;
    i += 32;
  }
  return i + br_match_len(a + i, b + i, limit - i);
}

// the parse state of one chain (vector registers on the device)
struct LaneChain {
  // geometry
  uint32_t seg_start, seg_end, pos_end, seg_flags, cmd_cap, store_end;
  Command* cmds;
  // CreateBackwardReferences
  uint32_t position, apply, insert_length;
  int32_t dc[4];
  DictState ds;
  uint32_t no_dict;
  // the lazy loop (mod.rs:2440-2475): the match in hand while position + 1 is probed
  uint32_t lazy, delayed;
  uint32_t sr_len, sr_len_x_code, sr_distance, sr_score;
  // exit bookkeeping
  uint32_t n_cmds, n_lits, n_searches, n_pushes, n_bad, ext_len, last_dist_code, last_copy_len;
  uint32_t tail_kind, tail_base, tail_p1;
  // flags of this round (FlagWriter of lz77_chain.h, per lane)
  uint8_t* flags_next;
  uint32_t fw_enabled, fw_hi, tail_lo, tail_value;
  // checkpoints
  uint32_t cp_on, next_cp;
  uint32_t walked_from;
  uint32_t active;
};

BR_DEV uint8_t ln_unstored(const LaneChain& c, uint32_t q) { return q >= c.tail_lo ? (uint8_t)c.tail_value : (uint8_t)0; }
BR_DEV void ln_flag(const LaneChain& c, uint32_t q, uint8_t v) {
  if (c.fw_enabled && q < c.fw_hi) c.flags_next[q] = v;
}
// [a, b) := "not stored by the main loop" (FlagWriter::range with split 0)
BR_DEV void ln_flag_unstored_range(const LaneChain& c, uint32_t a, uint32_t b) {
  if (!c.fw_enabled) return;
  if (b > c.fw_hi) b = c.fw_hi;
  for (uint32_t q = a; q < b; ++q) c.flags_next[q] = ln_unstored(c, q);
}
// the StoreRange part [first, copy_end) of a copy (FlagWriter::copy_range without masked entries): stored up to store_end
BR_DEV void ln_flag_copy_range(const LaneChain& c, uint32_t first, uint32_t copy_end) {
  if (!c.fw_enabled) return;
  uint32_t b = copy_end > c.fw_hi ? c.fw_hi : copy_end;
  if (first >= b) return;
  uint32_t q = first;
  const uint32_t ones_end = b < c.store_end ? b : c.store_end;  // [q, ones_end) := 1
  if (ones_end > q) {
    const uint32_t n = ones_end - q;
    if (n >= 8) {
      // whole words, the last one overlapping
      for (; q + 8 <= ones_end; q += 8) ln_store8(c.flags_next + q, 0x0101010101010101ull);
      if (q < ones_end) ln_store8(c.flags_next + ones_end - 8, 0x0101010101010101ull);
    } else if (n >= 4) {
      ln_store4(c.flags_next + q, 0x01010101u);
      ln_store4(c.flags_next + ones_end - 4, 0x01010101u);
    } else {
      for (; q < ones_end; ++q) c.flags_next[q] = 1;
    }
    q = ones_end;
  }
  for (; q < b; ++q) c.flags_next[q] = ln_unstored(c, q);
}
// FlagWriter::head: the part [a, b) of the previous chain's last step (kind, base, p1) that lies in this segment
BR_DEV void ln_flag_head(const LaneChain& c, uint32_t kind, uint32_t base, uint32_t p1, uint32_t a, uint32_t b) {
  if (!c.fw_enabled || kind == kHeadNone) return;
  const uint32_t step_end = b;
  if (b > c.fw_hi) b = c.fw_hi;
  for (uint32_t q = a; q < b; ++q) {
    uint8_t v;
    if (kind == kHeadCopy) {
      if (q <= base) v = (uint8_t)(kFlagStored | kFlagSearched);
      else if (q == base + 1) v = (p1 & 1u) ? (uint8_t)(kFlagStored | kFlagSearched) : ln_unstored(c, q);
      else v = q < c.store_end ? (uint8_t)1 : ln_unstored(c, q);  // (copy_value: q < step_end at every call)
    } else if (kind == kHeadUnstored) {
      v = ln_unstored(c, q);
    } else if (kind == kHeadVec4) {
      v = ((q - base) & 3) == 0;
    } else {
      v = ((q - base) & 1) == 0;
    }
    c.flags_next[q] = v;
  }
  (void)step_end;
}

// ---- AdvHasher::FindLongestMatch (mod.rs:1684-1812) for the position `cur` of one chain: the four distance-cache candidates and
// the candidate row of the position, folded in the reference's order, then the static dictionary.  The fold is the general form
// of br_fold_probe (lz77_chain.h): a later candidate replaces the best one if it passes the quick reject at best_len -- which,
// knowing every candidate's unbroken length, reads "longer than best_len, or as long when the best match already reaches the
// block end and the byte behind it agrees" -- and scores strictly higher; ring-buffer wraps cut the walk as they do there.
template <uint32_t kHtl>
BR_DEV SearchResult ln_search(const Lz77Params& P, const ChainTables& t, LaneChain& c, uint32_t cur) {
  constexpr uint32_t kCache = 4, kCand = kCache + kRowEntries;
  const uint32_t pos_end = c.pos_end;
  const uint32_t max_length = pos_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  const uint8_t* cur_data = t.text + cur;
  // ---- where the candidates are
  uint32_t prev[kCand];
LN_UNROLL
  for (uint32_t i = 0; i < kCache; ++i) {
    const int64_t b = (int64_t)c.dc[i];
    prev[i] = (b > 0 && b <= (int64_t)max_backward) ? cur - (uint32_t)b : 0xffffffffu;
  }
  {
    const uint8_t* row = (const uint8_t*)(t.rows + (size_t)cur * kRowEntries);
LN_UNROLL
    for (uint32_t j = 0; j < kRowEntries / 4; ++j) {
      const Ln16 v = ln_load16(row + 16 * j);
      prev[kCache + 4 * j + 0] = v.w[0];
      prev[kCache + 4 * j + 1] = v.w[1];
      prev[kCache + 4 * j + 2] = v.w[2];
      prev[kCache + 4 * j + 3] = v.w[3];
    }
  }
  // ---- their text: 32 bytes of each, requested together (text is padded: reading past the end of the block is harmless)
  const Ln16 a0 = ln_load16(cur_data), a1 = ln_load16(cur_data + 16);
  Ln16 b0[kCand], b1[kCand];
LN_UNROLL
  for (uint32_t i = 0; i < kCand; ++i) {
    const uint8_t* src = t.text + (prev[i] != 0xffffffffu ? prev[i] : cur);
    b0[i] = ln_load16(src);
    b1[i] = ln_load16(src + 16);
  }
  // ---- unbroken match lengths (br_match_len_wide)
  uint32_t unbroken[kCand];
LN_UNROLL
  for (uint32_t i = 0; i < kCand; ++i) {
    uint32_t n = ln_common16(a0, b0[i]);
    if (LN_ANY(n == 16 && prev[i] != 0xffffffffu)) {
      if (n == 16) n = 16 + ln_common16(a1, b1[i]);
      if (LN_ANY(n == 32 && max_length > 32 && prev[i] != 0xffffffffu)) {
        if (n == 32 && max_length > 32 && prev[i] != 0xffffffffu) n = ln_match_beyond32(t.text + prev[i], cur_data, max_length);
      }
    }
    unbroken[i] = n < max_length ? n : max_length;
  }
  // ---- the fold
  SearchResult out;
  out.len = 0;
  out.len_x_code = 0;
  out.distance = 0;
  out.score = kMinScore;
  out.found = false;
  out.stored = true;
  uint32_t best_len = 0, best_score = kMinScore;
  const uint32_t mask = P.ring_mask;
  const uint32_t cur_ring = cur & mask;
  bool open = true;  // false: the walk is over (a ring-buffer wrap at the searched position, or the end of the row)
LN_UNROLL
  for (uint32_t i = 0; i < kCand; ++i) {
    const bool is_cache = i < kCache;
    const uint32_t q = prev[i];
    const bool has = q != 0xffffffffu;
    if (!is_cache && !has) open = false;  // (kRowEnd: entries are packed from the front)
    if (cur_ring + best_len > mask) open = false;
    const uint32_t u = unbroken[i];
    const bool type_ok = is_cache ? (u >= 3 || (u == 2 && i < 2)) : u >= 4;
    const uint32_t backward = cur - q;
    const uint32_t score = is_cache ? br_score_cache<false>(P, u, i) : br_score_ring<false>(P, u, has ? backward : 1u);
    bool longer = u > best_len;
    if (u == best_len && best_len == max_length && has && type_ok && open) {
      // a match that runs to the end of the block: the byte behind it decides (ring-buffer semantics, br_unwritten_byte)
      longer = br_unwritten_byte(P, t, cur + max_length) == t.text[q + max_length];
    }
    const bool pass = open && has && type_ok && !((q & mask) + best_len > mask) && longer && score > best_score;
    if (pass) {
      best_len = u;
      best_score = score;
      out.len = u;
      out.distance = backward;
      out.score = score;
      out.found = true;
    }
  }
  // ---- the static dictionary (SearchInStaticDictionary, mod.rs:1942-1988)
  if (!out.found && P.use_dictionary) {
    const uint32_t first4 = a0.w[0];
    const uint32_t packed = (t.dict_items != nullptr && !c.no_dict) ? t.dict_items[cur] : 0u;
    br_dictionary_stage(P, t, c.ds, c.no_dict != 0, max_length, max_backward, out, [&](uint32_t i, uint32_t* item_out, uint32_t* matchlen_out) {
      const uint32_t item = t.dict_items != nullptr ? ((packed >> (16u * i)) & 0xffffu)
                                                    : (uint32_t)t.dict_hash[(((first4 * 0x1e35a7bdu) >> (32 - 14)) << 1) + i];
      uint32_t matchlen = 0;
      if (item != 0) {
        const uint32_t wlen = item & 0x1f;
        if (wlen <= max_length) matchlen = br_match_len(t.dict_data + t.dict_offsets_by_length[wlen] + wlen * (item >> 5), cur_data, wlen);
      }
      *item_out = item;
      *matchlen_out = matchlen;
    });
  }
  return out;
}

// ---- setting a chain up: br_parse_segment up to its loop
template <uint32_t kHtl>
BR_DEV void ln_begin(const Lz77Params& P, const ChainTables& t, const Segment& seg, const SegEntry& entry, LaneChain& c) {
  const uint32_t window = P.spree_window;
  c.seg_start = seg.start;
  c.seg_end = seg.end;
  c.pos_end = seg.blk_end;
  c.seg_flags = seg.flags;
  c.cmd_cap = seg.cmd_cap;
  c.cmds = t.cmds + (size_t)seg.cmd_base;
  c.position = entry.pos;
  c.apply = entry.apply;
  c.insert_length = 0;
  for (int i = 0; i < 4; ++i) c.dc[i] = entry.cache[i];
  c.ds.lookups = c.ds.lookups0 = entry.dict_lookups;
  c.ds.matches = c.ds.matches0 = entry.dict_matches;
  c.ds.mode = 0;
  c.ds.maxdef = -(1 << 30);
  c.ds.vlookups = 0;
  c.ds.vwould = 0;
  c.ds.vmaxdef = -(1 << 30);
  c.no_dict = (P.use_dictionary && entry.dict_exact && c.ds.matches < (c.ds.lookups >> 7)) ? 1u : 0u;
  c.lazy = 0;
  c.delayed = 0;
  c.sr_len = c.sr_len_x_code = c.sr_distance = c.sr_score = 0;
  c.n_cmds = c.n_lits = c.n_searches = c.n_pushes = c.n_bad = c.ext_len = 0;
  c.last_dist_code = 0xffffffffu;
  c.last_copy_len = 0;
  c.tail_kind = kHeadNone;
  c.tail_base = 0;
  c.tail_p1 = 0;
  c.flags_next = t.flags_next;
  c.fw_enabled = (seg.flags & kSegWarmup) ? 0u : 1u;
  c.fw_hi = seg.end;
  c.tail_lo = c.pos_end - 3;
  c.tail_value = (seg.flags & kSegTailStitched) ? 1u : 0u;
  c.store_end = c.pos_end >= kHtl ? c.pos_end - kHtl + 1 : 0;
  c.walked_from = entry.pos;
  if (seg.flags & kSegFirstInBlock) {
    c.position = seg.blk_start;
    if (entry.ext_allowed) {
      // extend_last_command, encode.rs:360-400: the previous copy continues while bytes keep matching
      const uint32_t d = (uint32_t)c.dc[0];
      const uint32_t limit = c.pos_end - c.position;
      const uint32_t n = br_match_len(t.text + c.position, t.text + c.position - d, limit);
      c.ext_len = n;
      ln_flag_unstored_range(c, c.position, c.position + n);
      c.tail_kind = kHeadUnstored;
      c.tail_base = c.position;
      c.position += n;
    }
    c.apply = c.position + window;
  } else {
    // the part of the previous chain's last step that lies in this segment
    c.tail_kind = entry.head_kind;
    c.tail_base = entry.head_base;
    c.tail_p1 = entry.head_p1;
    if (c.position > seg.start) ln_flag_head(c, c.tail_kind, c.tail_base, c.tail_p1, seg.start, c.position);
  }
  c.cp_on = (t.checkpoints != nullptr && c.fw_enabled) ? 1u : 0u;
  c.next_cp = c.cp_on ? (seg.start / kCheckpointStride + 1u) * kCheckpointStride : 0xffffffffu;
  c.active = 1;
}

BR_DEV void ln_write_checkpoint(const ChainTables& t, const LaneChain& c) {
  Checkpoint r;
  r.pos = c.position;
  r.insert_len = c.insert_length;
  r.apply = c.apply;
  for (int i = 0; i < 4; ++i) r.dc[i] = c.dc[i];
  r.n_cmds = c.n_cmds;
  r.n_lits = c.n_lits;
  r.n_searches = c.n_searches;
  r.n_pushes = c.n_pushes;
  r.n_bad = c.n_bad;
  r.last_dist_code = c.last_dist_code;
  r.last_copy_len = c.last_copy_len;
  r.ext_len = c.ext_len;
  r.tail_kind = c.tail_kind;
  r.tail_base = c.tail_base;
  r.tail_p1 = c.tail_p1;
  r.d_lookups = c.ds.lookups;
  r.d_matches = c.ds.matches;
  r.d_mode = c.ds.mode;
  r.d_maxdef = c.ds.maxdef;
  r.d_vlookups = c.ds.vlookups;
  r.d_vwould = c.ds.vwould;
  r.d_vmaxdef = c.ds.vmaxdef;
  r.entry_lookups = c.ds.lookups0;
  r.entry_matches = c.ds.matches0;
  r.no_dict = c.no_dict;
  r.valid = kCheckpointValid;
  r.pad[0] = r.pad[1] = r.pad[2] = 0;
  t.checkpoints[c.next_cp / kCheckpointStride] = r;
}

// ---- one step: one search and what follows from it
template <uint32_t kHtl>
BR_DEV void ln_step(const Lz77Params& P, const ChainTables& t, LaneChain& c) {
  const uint32_t pos_end = c.pos_end;
  const uint32_t window = P.spree_window;
  if (!c.lazy) {
    // the loop top of CreateBackwardReferences
    if (!(c.position + kHtl < pos_end && c.position < c.seg_end)) {
      c.active = 0;
      return;
    }
    // checkpoints: the first loop-top position at or behind every boundary (br_parse_segment)
    while (c.next_cp <= c.position && c.next_cp < c.seg_end) {
      if (c.cp_on) ln_write_checkpoint(t, c);
      c.next_cp += kCheckpointStride;
    }
  }
  const uint32_t cur = c.position + (c.lazy ? 1u : 0u);
  const SearchResult sr = ln_search<kHtl>(P, t, c, cur);
  c.n_searches++;
  bool emit = false;
  uint32_t next_probed = 0;
  if (!c.lazy) {
    if (sr.found) {
      // a match: look at position + 1 before taking it (the next step)
      c.lazy = 1;
      c.delayed = 0;
      c.sr_len = sr.len;
      c.sr_len_x_code = sr.len_x_code;
      c.sr_distance = sr.distance;
      c.sr_score = sr.score;
    } else {
      ln_flag(c, c.position, (uint8_t)(kFlagStored | kFlagSearched));
      c.insert_length++;
      c.position++;
      if (c.position > c.apply) {
        const uint32_t margin = kHtl - 1 > 4 ? kHtl - 1 : 4;
        if (c.position + 16 >= pos_end - margin) {
          c.tail_kind = kHeadUnstored;
          c.tail_base = c.position;
          ln_flag_unstored_range(c, c.position, pos_end);
          c.insert_length += pos_end - c.position;
          c.position = pos_end;
        } else if (c.position > c.apply + 4 * window) {
          // Store4Vec4: position, +4, +8, +12
          c.tail_kind = kHeadVec4;
          c.tail_base = c.position;
          if (c.fw_enabled)
            for (uint32_t q = c.position; q < c.position + 16; ++q)
              if (q < c.fw_hi) c.flags_next[q] = ((q - c.position) & 3) == 0;
          c.insert_length += 16;
          c.position += 16;
        } else {
          // StoreEvenVec4: position, +2, +4, +6
          c.tail_kind = kHeadEven4;
          c.tail_base = c.position;
          if (c.fw_enabled)
            for (uint32_t q = c.position; q < c.position + 8; ++q)
              if (q < c.fw_hi) c.flags_next[q] = ((q - c.position) & 1) == 0;
          c.insert_length += 8;
          c.position += 8;
        }
      }
    }
  } else {
    next_probed = 1;
    emit = true;
    if (sr.found && sr.score >= c.sr_score + 175) {
      ln_flag(c, c.position, (uint8_t)(kFlagStored | kFlagSearched));
      c.position++;
      c.insert_length++;
      c.sr_len = sr.len;
      c.sr_len_x_code = sr.len_x_code;
      c.sr_distance = sr.distance;
      c.sr_score = sr.score;
      next_probed = 0;
      if (++c.delayed < 4 && c.position + kHtl < pos_end) emit = false;  // probe the next position as well
    }
  }
  if (emit) {
    c.lazy = 0;
    const uint32_t len = c.sr_len;
    c.apply = c.position + 2 * len + window;
    const uint32_t max_distance = c.position < P.max_backward_limit ? c.position : P.max_backward_limit;
    const uint32_t distance_code = br_compute_distance_code(c.sr_distance, max_distance, c.dc);
    if (c.sr_distance <= max_distance && distance_code > 0) {
      c.dc[3] = c.dc[2];
      c.dc[2] = c.dc[1];
      c.dc[1] = c.dc[0];
      c.dc[0] = (int32_t)c.sr_distance;
      c.n_pushes++;
    }
    if (c.n_cmds < c.cmd_cap && c.fw_enabled) c.cmds[c.n_cmds] = br_raw_command(c.insert_length, len, len ^ c.sr_len_x_code, distance_code);
    c.n_cmds++;
    if (len < 2) c.n_bad++;
    c.n_lits += c.insert_length;
    c.insert_length = 0;
    c.last_dist_code = distance_code;
    c.last_copy_len = len;
    // hash-table side effects: position searched, position + 1 only if probed, then StoreRange
    c.tail_kind = kHeadCopy;
    c.tail_base = c.position;
    c.tail_p1 = next_probed;
    ln_flag(c, c.position, (uint8_t)(kFlagStored | kFlagSearched));
    if (len > 1) ln_flag(c, c.position + 1, next_probed ? (uint8_t)(kFlagStored | kFlagSearched) : ln_unstored(c, c.position + 1));
    if (len > 2) ln_flag_copy_range(c, c.position + 2, c.position + len);
    c.position += len;
  }
}

// ---- the end of br_parse_segment: what is left of the block, the exit record
BR_DEV void ln_end(const ChainTables& t, LaneChain& c, SegExit& exit_out) {
  if (c.cp_on) {
    // boundaries this parse never reached at a loop top: whatever record sits there belongs to an older parse
    for (; c.next_cp < c.seg_end; c.next_cp += kCheckpointStride) t.checkpoints[c.next_cp / kCheckpointStride].valid = 0;
  }
  if (c.seg_flags & kSegLastInBlock) {
    if (c.position < c.pos_end) ln_flag_unstored_range(c, c.position, c.pos_end);
    c.insert_length += c.pos_end - c.position;
    c.position = c.pos_end;
  }
  const DictState& ds = c.ds;
  SegExit x;
  x.pos = c.position;
  x.apply = c.apply;
  for (int i = 0; i < 4; ++i) x.cache[i] = c.dc[i];
  x.insert_len = c.insert_length;
  x.n_cmds = c.n_cmds;
  x.n_lits = c.n_lits;
  x.ext_len = c.ext_len;
  x.dict_lookups = ds.mode == 2 ? ds.lookups + ds.vlookups : ds.lookups;
  x.dict_matches = ds.mode == 2 ? ds.matches + ds.vwould : ds.matches;
  x.last_dist_code = c.last_dist_code;
  x.bad_commands = c.n_bad;
  x.n_searches = c.n_searches;
  x.last_copy_len = c.last_copy_len;
  x.dict_mode = ds.mode;
  x.dict_maxdef = ds.mode == 2 ? ds.vmaxdef : ds.maxdef;
  x.n_pushes = c.n_pushes < 4 ? c.n_pushes : 4u;
  x.tail_kind = c.position > c.seg_end ? c.tail_kind : (uint32_t)kHeadNone;
  x.tail_base = c.position > c.seg_end ? c.tail_base : 0u;
  x.tail_p1 = c.position > c.seg_end ? c.tail_p1 : 0u;
  x.n_pushes_all = c.n_pushes;
  x.dict_entry_lookups = ds.lookups0;
  x.dict_entry_matches = ds.matches0;
  exit_out = x;
}

// One chain from its entry to its exit (the host emulation calls this per segment; the kernel runs it in every lane).
template <uint32_t kHtl>
BR_DEV void br_lane_parse(const Lz77Params& P, const ChainTables& t, const Segment& seg, const SegEntry& entry, SegExit& exit_out, uint32_t* walked,
                          uint32_t* searches, uint32_t* commands) {
  LaneChain c;
  ln_begin<kHtl>(P, t, seg, entry, c);
  while (c.active) ln_step<kHtl>(P, t, c);
  ln_end(t, c, exit_out);
  *walked = c.position - c.walked_from;
  *searches = c.n_searches;
  *commands = c.fw_enabled ? c.n_cmds : 0u;
}

}  // namespace brotli_mi355x
#endif
