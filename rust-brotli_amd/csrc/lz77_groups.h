// lz77_groups.h -- the plain quality-5 parse with FOUR CHAINS PER WAVEFRONT (round 6): round 0 and the warm-up.
//
// lz77_chain.h gives a chain a whole wavefront: the candidates of a search sit in the lanes, the parse state in scalar
// registers, and the greedy / lazy control flow of CreateBackwardReferences (backward_references/mod.rs:2376-2552) runs
// wave-uniform.  Measured over three rounds: ~125 scalar + ~128 vector wave-instructions per search, of which the lanes do
// useful work in a dozen (a candidate row of text holds one to three candidates) -- the launch is bound by instruction ISSUE,
// and the instructions are per-search overhead, not per-candidate work.  Round 5 tried the other extreme, one chain per LANE
// (64 to a wavefront, lz77_lanes.h in the history): 2.4 x slower, because 32 768 chains are then half a wavefront per SIMD and
// every candidate of a search is evaluated one after the other.  This file is the point in between: a chain is a GROUP of 16
// lanes (one DPP row), four chains to a wavefront.
//   * Lane layout of a search (i = lane & 15), two candidate slots per lane: i = 0..3 the four distance-cache candidates;
//     i = 4..11 ring entries i - 4 (slot A) and i + 4 (slot B) of the position's candidate row (lz77_rows.h) -- all sixteen, in ring
//     order A then B; i = 12, 13 the two static-dictionary probes, compared in the same memory round trip.  One text comparison
//     serves all candidates of four searches.
//   * The parse state lives in VECTOR registers, replicated in the 16 lanes of the group: no scalar state, no scalar spills, no
//     wave-uniform control flow per chain.
//   * The loop is the state machine of the lane experiment: every step is exactly ONE FindLongestMatch (AdvHasher,
//     mod.rs:1684-1812) followed by the transition it causes, written with selects, so that the four chains of a wave stay
//     converged; the rare paths (a literal spree's jump, a checkpoint, a match beyond 16 bytes) sit behind wave-level tests.
//   * The "first strictly better candidate in ring order" fold runs for the four groups side by side: a ballot, the group's
//     16 bits of it, find-first-set, two lane permutes per improvement step.
//   * The candidate row and the dictionary items of the position searched next are requested a step ahead, so a step is one
//     memory round trip (candidate text) in the common case.
//   * Memory is written by the group's leader lane (commands, records) or by the group's lanes side by side (flags).
// Same inputs, same outputs as br_parse_segment<false, true> without a splice: commands, flags, exit record, checkpoints.
// Covered: the plain configuration (plain_q5_config: four cache candidates, 16-entry candidate rows, no custom-dictionary break,
// no hasher reset, no masked entries).  Everything else, and every list launch, keeps the wave-per-chain kernels.
// The host emulation compiles the same state machine with one lane per group and the sequential search (tests/emu/device_emu.cpp).
#ifndef BROTLI_MI355X_LZ77_GROUPS_H_
#define BROTLI_MI355X_LZ77_GROUPS_H_

#include "lz77_chain.h"

namespace brotli_mi355x {

#if defined(BROTLI_HOST_EMU)
#define LG_WIDTH 1u
#define LG_IDX 0u
#define LG_LEADER true
#define LG_ANY(x) (x)
#define LG_NOUNROLL
#else
#define LG_WIDTH 16u
#define LG_IDX ((uint32_t)threadIdx.x & 15u)
#define LG_LEADER (((uint32_t)threadIdx.x & 15u) == 0u)
#define LG_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#define LG_NOUNROLL _Pragma("nounroll")
#endif

// ---- cold state: what a step rarely touches (counters, the books of the dictionary throttle, the last step), behind accessors
// (Tried on the device: one field per LANE of a single register, read with a DPP row broadcast and written with a lane-masked select --
// 3 registers instead of 20.  It compiled to what was meant, as far as the disassembly was read, and gave other exit records than the
// plain form on the MI355X; not resolved in the time there was, so the plain form it is: every field in a register of its own.)
struct Cold {
  uint32_t v[16];
};
template <int J>
BR_DEV uint32_t cold_get(const Cold& c) {
  return c.v[J];
}
template <int J>
BR_DEV void cold_set(Cold& c, uint32_t value, bool cond = true) {
  if (cond) c.v[J] = value;
}
template <int J>
BR_DEV void cold_add(Cold& c, uint32_t delta, bool cond = true) {
  if (cond) c.v[J] += delta;
}
BR_DEV void cold_clear(Cold& c) {
  for (int j = 0; j < 16; ++j) c.v[j] = 0;
}

// fields of GroupChain::a
enum : int { kA_n_cmds = 0, kA_n_lits, kA_n_pushes, kA_n_bad, kA_ext_len, kA_last_dist_code, kA_last_copy_len, kA_tail_kind, kA_tail_base,
             kA_tail_p1, kA_walked_from, kA_cmd_cap, kA_cmd_base };
// fields of GroupChain::b: the books of the static-dictionary throttle (DictState) besides the two counters
enum : int { kB_lookups0 = 0, kB_matches0, kB_mode, kB_maxdef, kB_vlookups, kB_vwould, kB_vmaxdef };
// bits of GroupChain::st
enum : uint32_t { kStActive = 1u, kStLazy = 2u, kStDelayedShift = 2u, kStDelayedMask = 3u << 2, kStFlags = 16u, kStCheckpoints = 32u, kStNoDict = 64u,
                  kStTailValue = 128u, kStLastInBlock = 256u };

// the parse state of one chain
struct GroupChain {
  // hot: replicated in the lanes of the group
  uint32_t seg_end, pos_end;
  uint32_t position, apply, insert_length;
  int32_t dc0, dc1, dc2, dc3;  // (four scalars, not an array: a lane-indexed array would pin the state in scratch memory)
  uint32_t st;
  uint32_t sr_len, sr_len_x_code, sr_distance, sr_score;  // the lazy loop (mod.rs:2440-2475): the match in hand while position + 1 is probed
  uint32_t n_searches, next_cp;
  uint32_t lookups, matches;  // static-dictionary throttle counters (mod.rs:1957-1960)
  // device: the lane's entry of the candidate row and the dictionary items of the two positions most likely searched next
  uint32_t pfa_pos, pfa_row, pfa_items, pfb_pos, pfb_row, pfb_items;
  Cold a, b;
};

BR_DEV uint8_t lg_unstored(const GroupChain& c, uint32_t q) { return (q >= c.pos_end - 3u && (c.st & kStTailValue)) ? (uint8_t)1 : (uint8_t)0; }
// [a, b) := "not stored by the main loop" (FlagWriter::range with split 0)
BR_DEV void lg_flag_unstored_range(const ChainTables& t, const GroupChain& c, uint32_t a, uint32_t b) {
  if (!(c.st & kStFlags)) return;
  if (b > c.seg_end) b = c.seg_end;
  LG_NOUNROLL
  for (uint32_t q = a + LG_IDX; q < b; q += LG_WIDTH) t.flags_next[q] = lg_unstored(c, q);
}
// FlagWriter::head: the part [a, b) of the previous chain's last step (kind, base, p1) that lies in this segment
BR_DEV void lg_flag_head(const ChainTables& t, const GroupChain& c, uint32_t store_end, uint32_t kind, uint32_t base, uint32_t p1, uint32_t a, uint32_t b) {
  if (!(c.st & kStFlags) || kind == kHeadNone) return;
  if (b > c.seg_end) b = c.seg_end;
  LG_NOUNROLL
  for (uint32_t q = a + LG_IDX; q < b; q += LG_WIDTH) {
    uint8_t v;
    if (kind == kHeadCopy) {
      if (q <= base) v = (uint8_t)(kFlagStored | kFlagSearched);
      else if (q == base + 1) v = (p1 & 1u) ? (uint8_t)(kFlagStored | kFlagSearched) : lg_unstored(c, q);
      else v = q < store_end ? (uint8_t)1 : lg_unstored(c, q);  // (copy_value: q < the end of the step at every call)
    } else if (kind == kHeadUnstored) {
      v = lg_unstored(c, q);
    } else if (kind == kHeadVec4) {
      v = ((q - base) & 3) == 0;
    } else {
      v = ((q - base) & 1) == 0;
    }
    t.flags_next[q] = v;
  }
}

// one of four values by a small index (written with selects on laundered values: the device compiler otherwise turns the chain of
// selects over adjacent fields into an indexed load, which pins the whole chain state in scratch memory)
BR_DEV int32_t lg_pick4(int32_t a, int32_t b, int32_t c, int32_t d, uint32_t i) {
#if !defined(BROTLI_HOST_EMU)
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#endif
  int32_t r = d;
  r = i == 2 ? c : r;
  r = i == 1 ? b : r;
  r = i == 0 ? a : r;
  return r;
}

// ---- AdvHasher::FindLongestMatch (mod.rs:1684-1812) for the position `cur` of one chain, candidate by candidate in the
// reference's order: the four distance-cache candidates and the candidate row of the position, then (not here) the static
// dictionary.  The general form of br_fold_probe (lz77_chain.h): a later candidate replaces the best one if it passes the quick
// reject at best_len -- which, knowing the candidate's unbroken length, reads "longer than best_len, or as long when the best
// match already reaches the block end and the byte behind it agrees" -- and scores strictly higher; ring-buffer wraps cut the walk
// as they do there.  The host emulation runs every search through it; on the device it is the fall-back of a group whose
// search has a candidate next to a ring-buffer wrap or the block end (one lane's worth of work done by all sixteen: rare).
BR_DEV SearchResult lg_search_sequential(const Lz77Params& P, const ChainTables& t, const GroupChain& c, uint32_t cur) {
  constexpr uint32_t kCache = 4, kCand = kCache + kRowEntries;
  const uint32_t pos_end = c.pos_end;
  const uint32_t max_length = pos_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
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
  LG_NOUNROLL
  for (uint32_t i = 0; i < kCand; ++i) {
    const bool is_cache = i < kCache;
    uint32_t q;
    if (is_cache) {
      const int64_t b = (int64_t)lg_pick4(c.dc0, c.dc1, c.dc2, c.dc3, i);
      q = (b > 0 && b <= (int64_t)max_backward) ? cur - (uint32_t)b : 0xffffffffu;
    } else {
      q = open ? t.rows[(size_t)cur * kRowEntries + (i - kCache)] : 0xffffffffu;
    }
    const bool has = q != 0xffffffffu;
    if (!is_cache && !has) open = false;  // (kRowEnd: entries are packed from the front)
    if (cur_ring + best_len > mask) open = false;
    if (!open && !is_cache) break;
    if (!has || !open) continue;
    const uint32_t u = br_match_len(t.text + q, t.text + cur, max_length);
    const bool type_ok = is_cache ? (u >= 3 || (u == 2 && i < 2)) : u >= 4;
    const uint32_t backward = cur - q;
    const uint32_t score = is_cache ? br_score_cache<false>(P, u, i) : br_score_ring<false>(P, u, backward);
    bool longer = u > best_len;
    if (u == best_len && best_len == max_length && type_ok) {
      // a match that runs to the end of the block: the byte behind it decides (ring-buffer semantics, br_unwritten_byte)
      longer = br_unwritten_byte(P, t, cur + max_length) == t.text[q + max_length];
    }
    const bool pass = type_ok && !((q & mask) + best_len > mask) && longer && score > best_score;
    if (pass) {
      best_len = u;
      best_score = score;
      out.len = u;
      out.distance = backward;
      out.score = score;
      out.found = true;
    }
  }
  return out;
}

// ---- the static dictionary stage of FindLongestMatch (SearchInStaticDictionary + TestStaticDictionaryItem, mod.rs:1891-1988, and
// the bookkeeping of br_dictionary_stage in lz77_chain.h) written with selects: `run` says whether the lane's chain consults the
// dictionary in this step (a search that found nothing); item / matchlen of the two probes are handed in.
BR_DEV void lg_dictionary_stage(const Lz77Params& P, GroupChain& c, bool run, uint32_t max_length, uint32_t max_backward, uint32_t item0,
                                uint32_t matchlen0, uint32_t item1, uint32_t matchlen1, SearchResult& out) {
  const bool no_dict = (c.st & kStNoDict) != 0;
  const bool dead = c.matches < (c.lookups >> 7);
  const uint32_t seen = dead ? 2u : 1u;
  const uint32_t mode = cold_get<kB_mode>(c.b);
  // switched off for good under the exact counters of this round: no probes, no virtual bookkeeping ("ran blind", mode 4)
  cold_set<kB_mode>(c.b, no_dict ? 4u : ((mode == 0 || mode == seen) ? seen : 3u), run);
  const uint32_t vwould = cold_get<kB_vwould>(c.b), vlookups = cold_get<kB_vlookups>(c.b);
  const bool go = run && !no_dict && !(dead && vwould);
  {
    const int32_t vmaxdef = (int32_t)cold_get<kB_vmaxdef>(c.b), maxdef = (int32_t)cold_get<kB_maxdef>(c.b);
    const int32_t def = (int32_t)(c.lookups - cold_get<kB_lookups0>(c.b)) - 128 * (int32_t)(c.matches - cold_get<kB_matches0>(c.b));
    cold_set<kB_vmaxdef>(c.b, (uint32_t)((int32_t)vlookups > vmaxdef ? (int32_t)vlookups : vmaxdef), go && dead);
    cold_set<kB_maxdef>(c.b, (uint32_t)(def > maxdef ? def : maxdef), go && !dead);
  }
  uint32_t threshold = out.score;
  bool would = false;
#if !defined(BROTLI_HOST_EMU)
#pragma unroll
#endif
  for (uint32_t k = 0; k < 2; ++k) {
    const uint32_t item = k == 0 ? item0 : item1, matchlen = k == 0 ? matchlen0 : matchlen1;
    const uint32_t len = item & 0x1f, dist = item >> 5;
    bool ok = go && item != 0 && len <= max_length && !(matchlen + 10 <= len || matchlen == 0);
    const uint32_t cut = (len - matchlen) & 15u;  // (< 10 whenever ok)
    const uint32_t transform_id = (cut << 2) + (uint32_t)((0x071b520ada2d3200ull >> (cut * 6 > 63 ? 63 : cut * 6)) & 0x3f);
    const uint32_t backward = max_backward + dist + 1 + (transform_id << br_dict_size_bits(len));
    ok = ok && backward <= P.dist_max_distance;
    const uint32_t score = 30 * 8 * 8 + P.score_per_byte * matchlen - 30 * br_log2_floor_nonzero(backward | 1u);
    ok = ok && score >= threshold;
    threshold = ok ? score : threshold;
    would = would || (ok && dead);
    const bool take = ok && !dead;
    out.len = take ? matchlen : out.len;
    out.len_x_code = take ? (len ^ matchlen) : out.len_x_code;
    out.distance = take ? backward : out.distance;
    out.score = take ? score : out.score;
    out.found = out.found || take;
    c.matches += take ? 1u : 0u;
  }
  c.lookups += (go && !dead) ? 2u : 0u;
  cold_add<kB_vlookups>(c.b, 2u, go && dead);
  cold_set<kB_vwould>(c.b, 1u, would);
}

#if !defined(BROTLI_HOST_EMU)
BR_DEV uint32_t lg_shfl(uint32_t v, uint32_t lane) { return (uint32_t)__shfl((int)v, (int)lane, 64); }
// DPP moves inside a row of 16 lanes (= a group): lane i reads lane i - 4 / lane i + 4 of its row (0 where there is none)
BR_DEV uint32_t lg_from_lane_minus4(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); }  // row_shr:4
BR_DEV uint32_t lg_from_lane_plus4(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true); }   // row_shl:4
// bit number of the lowest set bit, 0xffffffff for 0 (v_ffbl_b32 as the hardware defines it; __ffs costs a compare and a select more)
BR_DEV uint32_t lg_ffbl(uint32_t x) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// common prefix of two 16-byte vectors, 0..16 (br_common16v with the bare instruction)
BR_DEV uint32_t lg_common16v(const br_u32x4 a, const br_u32x4 b) {
  const uint32_t t0 = lg_ffbl(a.x ^ b.x), t1 = lg_ffbl(a.y ^ b.y) | 32u, t2 = lg_ffbl(a.z ^ b.z) | 64u, t3 = lg_ffbl(a.w ^ b.w) | 96u;
  const uint32_t bits = min(min(min(t0, t1), t2), t3);
  return min(bits >> 3, 16u);
}
#endif

// The search of one step.  `on`: the lane's chain takes part (a chain that has finished idles while the others of its wave go on).
// dict_off: kBrotliDictionaryOffsetsByLength in workgroup memory (device).
template <uint32_t kHtl>
BR_DEV SearchResult lg_search(const Lz77Params& P, const ChainTables& t, GroupChain& c, uint32_t cur, bool on, const uint32_t* dict_off) {
  SearchResult out;
  out.len = 0;
  out.len_x_code = 0;
  out.distance = 0;
  out.score = kMinScore;
  out.found = false;
  out.stored = true;
#if defined(BROTLI_HOST_EMU)
  (void)dict_off;
  if (!on) return out;
  const uint32_t max_length = c.pos_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  out = lg_search_sequential(P, t, c, cur);
  if (P.use_dictionary) {
    const bool probe = !out.found && !(c.st & kStNoDict);
    uint32_t item[2] = {0, 0}, matchlen[2] = {0, 0};
    if (probe) {
      const uint32_t first4 = br_load32(t.text + cur);
      for (uint32_t i = 0; i < 2; ++i) {
        item[i] = t.dict_items != nullptr ? ((t.dict_items[cur] >> (16u * i)) & 0xffffu)
                                          : (uint32_t)t.dict_hash[(((first4 * 0x1e35a7bdu) >> (32 - 14)) << 1) + i];
        const uint32_t wlen = item[i] & 0x1f;
        if (item[i] != 0 && wlen <= max_length)
          matchlen[i] = br_match_len(t.dict_data + t.dict_offsets_by_length[wlen] + wlen * (item[i] >> 5), t.text + cur, wlen);
      }
    }
    lg_dictionary_stage(P, c, !out.found, max_length, max_backward, item[0], matchlen[0], item[1], matchlen[1], out);
  }
  return out;
#else
  const uint32_t lane = (uint32_t)threadIdx.x, i = lane & 15u, gbase = lane & 48u;
  if (!on) cur = 0;  // (addresses stay inside the text; nothing of an idle group's lanes is used)
  const uint32_t max_length = on ? c.pos_end - cur : 0u;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  const bool is_cache = i < 4u, is_ring = i - 4u < 8u, is_dict = i - 12u < 2u;
  const bool items_ready = t.dict_items != nullptr;
  const bool use_dict = P.use_dictionary && !(c.st & kStNoDict);
  // ---- the candidate row (lane i holds entry i) and the dictionary items of the position: requested a step ahead in the common case
  uint32_t row_e, items;
  {
    const bool hit_a = c.pfa_pos == cur, hit_b = c.pfb_pos == cur;
    row_e = hit_a ? c.pfa_row : c.pfb_row;
    items = hit_a ? c.pfa_items : c.pfb_items;
    const bool need = on && !(hit_a || hit_b);
    if (LG_ANY(need)) {
      if (need) {
        row_e = t.rows[(size_t)cur * kRowEntries + i];
        items = items_ready ? t.dict_items[cur] : 0u;
      }
    }
    // what the next step most likely searches: cur + 1 (the lazy probe behind a match, the position behind a literal); behind a
    // lazy probe that does not improve on the match in hand, the position behind that match
    const uint32_t last = P.total_bytes - 1u;
    uint32_t na = cur + 1u, nb = (c.st & kStLazy) ? c.position + c.sr_len : cur + 1u;
    na = na < last ? na : last;
    nb = nb < last ? nb : last;
    c.pfa_pos = na;
    c.pfb_pos = nb;
    c.pfa_row = t.rows[(size_t)na * kRowEntries + i];
    c.pfb_row = t.rows[(size_t)nb * kRowEntries + i];
    c.pfa_items = items_ready ? t.dict_items[na] : 0u;
    c.pfb_items = items_ready ? t.dict_items[nb] : 0u;
  }
  // ---- where the candidates are
  const uint32_t e_a = lg_from_lane_minus4(row_e);  // lanes 4..11: entries 0..7
  const uint32_t e_b = lg_from_lane_plus4(row_e);   // lanes 4..11: entries 8..15
  const int32_t cb = lg_pick4(c.dc0, c.dc1, c.dc2, c.dc3, i);
  const uint32_t cache_prev = (cb > 0 && (uint32_t)cb <= max_backward) ? cur - (uint32_t)cb : 0xffffffffu;
  uint32_t item = 0;
  if (items_ready) {
    item = (is_dict && use_dict) ? ((items >> (16u * (i & 1u))) & 0xffffu) : 0u;
  } else if (LG_ANY(on && use_dict)) {
    if (is_dict && use_dict) item = (uint32_t)t.dict_hash[(((br_load32(t.text + cur) * 0x1e35a7bdu) >> (32 - 14)) << 1) + (i & 1u)];
  }
  const uint32_t wlen = item & 0x1fu;
  const bool dict_word = on && item != 0 && wlen <= max_length;
  const uint32_t prev_a = is_cache ? cache_prev : (is_ring ? e_a : 0xffffffffu);  // kRowEnd == "no candidate"
  const uint32_t prev_b = is_ring ? e_b : 0xffffffffu;
  const bool valid_a = on && prev_a != 0xffffffffu, valid_b = on && prev_b != 0xffffffffu;
  const uint8_t* cur_data = t.text + cur;
  const uint32_t word_off = dict_off[wlen] + wlen * (item >> 5);
  const uint8_t* src_a = (dict_word ? t.dict_data : t.text) + (dict_word ? word_off : (valid_a ? prev_a : cur));
  const uint8_t* src_b = t.text + (valid_b ? prev_b : cur);
  const uint32_t limit_a = dict_word ? wlen : max_length;
  // ---- their text
  uint32_t n_a, n_b;
  {
    const br_u32x4 x = *(const br_u32x4*)cur_data, ya = *(const br_u32x4*)src_a, yb = *(const br_u32x4*)src_b;
    n_a = lg_common16v(ya, x);
    n_b = lg_common16v(yb, x);
  }
  const bool long_a = (valid_a || dict_word) && n_a >= 16u && limit_a > 16u, long_b = valid_b && n_b >= 16u && max_length > 16u;
  if (LG_ANY(long_a || long_b)) {
    if (long_a) {
      const uint32_t n2 = br_common16(src_a + 16, cur_data + 16);
      n_a = 16u + n2;
      if (n2 >= 16u && limit_a > 32u) n_a = br_match_len_wide(src_a, cur_data, limit_a, t.run_end, prev_a, cur);
    }
    if (long_b) {
      const uint32_t n2 = br_common16(src_b + 16, cur_data + 16);
      n_b = 16u + n2;
      if (n2 >= 16u && max_length > 32u) n_b = br_match_len_wide(src_b, cur_data, max_length, t.run_end, prev_b, cur);
    }
  }
  const uint32_t len_a = (valid_a || dict_word) ? (n_a < limit_a ? n_a : limit_a) : 0u;  // (dictionary lanes: the common prefix with the word)
  const uint32_t len_b = valid_b ? (n_b < max_length ? n_b : max_length) : 0u;
  // ---- groups that need the sequential form: a candidate next to a ring-buffer wrap or reaching the block end, or the searched
  // position next to a wrap
  const bool special_a = valid_a && (((prev_a & P.ring_mask) + len_a > P.ring_mask) || len_a == max_length);
  const bool special_b = valid_b && (((prev_b & P.ring_mask) + len_b > P.ring_mask) || len_b == max_length);
  const bool cur_near_wrap = on && (cur & P.ring_mask) + max_length > P.ring_mask;
  const unsigned long long special_mask = __ballot(special_a || special_b || cur_near_wrap);
  const bool slow = ((uint32_t)(special_mask >> gbase) & 0xffffu) != 0u;
  // ---- the fold (br_fold_probe): the first lane of the group that is longer than the best so far and scores strictly higher
  // takes over, until none does; the A slots first, then the B slots
  const uint32_t back_a = cur - prev_a, back_b = cur - prev_b;
  const uint32_t penalty = i == 0 ? 0u : 39u + ((0x1ca10u >> (i & 0xeu)) & 0xeu);
  const uint32_t bias_a = is_cache ? (30u * 8u * 8u + 15u) - penalty : 30u * 8u * 8u - 30u * br_log2_floor_nonzero(valid_a ? back_a : 1u);
  const uint32_t score_a = P.score_per_byte * len_a + bias_a;
  const uint32_t score_b = P.score_per_byte * len_b + (30u * 8u * 8u - 30u * br_log2_floor_nonzero(valid_b ? back_b : 1u));
  const bool alive_a = valid_a && !slow && len_a >= (is_cache ? (i < 2u ? 2u : 3u) : 4u);
  const bool alive_b = valid_b && !slow && len_b >= 4u;
  uint32_t best_len = 0, best_score = kMinScore, best_from = 64u, best_slot = 0;
  for (;;) {
    const unsigned long long m = __ballot(alive_a && len_a > best_len && score_a > best_score);
    if (m == 0) break;
    const uint32_t gm = (uint32_t)(m >> gbase) & 0xffffu;
    const uint32_t from = gbase | (lg_ffbl(gm) & 15u);
    const uint32_t nl = lg_shfl(len_a, from), ns = lg_shfl(score_a, from);
    best_len = gm != 0 ? nl : best_len;
    best_score = gm != 0 ? ns : best_score;
    best_from = gm != 0 ? from : best_from;
  }
  if (LG_ANY(alive_b)) {
    for (;;) {
      const unsigned long long m = __ballot(alive_b && len_b > best_len && score_b > best_score);
      if (m == 0) break;
      const uint32_t gm = (uint32_t)(m >> gbase) & 0xffffu;
      const uint32_t from = gbase | (lg_ffbl(gm) & 15u);
      const uint32_t nl = lg_shfl(len_b, from), ns = lg_shfl(score_b, from);
      best_len = gm != 0 ? nl : best_len;
      best_score = gm != 0 ? ns : best_score;
      best_from = gm != 0 ? from : best_from;
      best_slot = gm != 0 ? 1u : best_slot;
    }
  }
  {
    const uint32_t d = lg_shfl(best_slot ? back_b : back_a, best_from & 63u);  // (every lane of a group asks for the same slot)
    const bool found = best_from != 64u;
    out.len = best_len;
    out.distance = found ? d : 0u;
    out.score = best_score;
    out.found = found;
  }
  if (special_mask != 0) {
    if (slow) out = lg_search_sequential(P, t, c, cur);
  }
  // ---- the static dictionary: the two probes were compared in lanes 12 and 13 of the group
  if (P.use_dictionary) {
    const bool run = on && !out.found;
    if (LG_ANY(run)) {
      const uint32_t matchlen = dict_word ? len_a : 0u;
      const uint32_t item0 = lg_shfl(item, gbase | 12u), item1 = lg_shfl(item, gbase | 13u);
      const uint32_t len0 = lg_shfl(matchlen, gbase | 12u), len1 = lg_shfl(matchlen, gbase | 13u);
      if (run) lg_dictionary_stage(P, c, true, max_length, max_backward, item0, len0, item1, len1, out);
    }
  }
  return out;
#endif
}

// ---- setting a chain up: br_parse_segment up to its loop
template <uint32_t kHtl>
BR_DEV void lg_begin(const Lz77Params& P, const ChainTables& t, const Segment& seg, const SegEntry& entry, GroupChain& c, bool have) {
  const uint32_t window = P.spree_window;
  c.seg_end = seg.end;
  c.pos_end = seg.blk_end;
  c.position = entry.pos;
  c.apply = entry.apply;
  c.insert_length = 0;
  c.dc0 = entry.cache[0];
  c.dc1 = entry.cache[1];
  c.dc2 = entry.cache[2];
  c.dc3 = entry.cache[3];
  c.lookups = entry.dict_lookups;
  c.matches = entry.dict_matches;
  cold_clear(c.a);
  cold_clear(c.b);
  cold_set<kB_lookups0>(c.b, entry.dict_lookups);
  cold_set<kB_matches0>(c.b, entry.dict_matches);
  cold_set<kB_maxdef>(c.b, (uint32_t)-(1 << 30));
  cold_set<kB_vmaxdef>(c.b, (uint32_t)-(1 << 30));
  cold_set<kA_last_dist_code>(c.a, 0xffffffffu);
  cold_set<kA_tail_kind>(c.a, (uint32_t)kHeadNone);
  cold_set<kA_walked_from>(c.a, entry.pos);
  cold_set<kA_cmd_cap>(c.a, seg.cmd_cap);
  cold_set<kA_cmd_base>(c.a, seg.cmd_base);
  const bool writes = !(seg.flags & kSegWarmup) && have;  // (a group without a segment of its own writes nothing)
  c.st = kStActive | (writes ? kStFlags : 0u) | ((seg.flags & kSegTailStitched) ? kStTailValue : 0u) | ((seg.flags & kSegLastInBlock) ? kStLastInBlock : 0u) |
         ((P.use_dictionary && entry.dict_exact && c.matches < (c.lookups >> 7)) ? kStNoDict : 0u);
  c.sr_len = c.sr_len_x_code = c.sr_distance = c.sr_score = 0;
  c.n_searches = 0;
  c.pfa_pos = c.pfb_pos = 0xffffffffu;
  c.pfa_row = c.pfb_row = c.pfa_items = c.pfb_items = 0;
  const uint32_t store_end = c.pos_end >= kHtl ? c.pos_end - kHtl + 1 : 0;
  if (seg.flags & kSegFirstInBlock) {
    c.position = seg.blk_start;
    if (entry.ext_allowed) {
      // extend_last_command, encode.rs:360-400: the previous copy continues while bytes keep matching
      const uint32_t d = (uint32_t)c.dc0;
      const uint32_t limit = c.pos_end - c.position;
      const uint32_t n = br_match_len(t.text + c.position, t.text + c.position - d, limit);
      cold_set<kA_ext_len>(c.a, n);
      lg_flag_unstored_range(t, c, c.position, c.position + n);
      cold_set<kA_tail_kind>(c.a, (uint32_t)kHeadUnstored);
      cold_set<kA_tail_base>(c.a, c.position);
      c.position += n;
    }
    c.apply = c.position + window;
  } else {
    // the part of the previous chain's last step that lies in this segment
    cold_set<kA_tail_kind>(c.a, entry.head_kind);
    cold_set<kA_tail_base>(c.a, entry.head_base);
    cold_set<kA_tail_p1>(c.a, entry.head_p1);
    if (c.position > seg.start) lg_flag_head(t, c, store_end, entry.head_kind, entry.head_base, entry.head_p1, seg.start, c.position);
  }
  const bool cp_on = t.checkpoints != nullptr && writes;
  c.st |= cp_on ? kStCheckpoints : 0u;
  c.next_cp = cp_on ? (seg.start / kCheckpointStride + 1u) * kCheckpointStride : 0xffffffffu;
}

// the record of the loop-top state at the boundary c.next_cp (see Checkpoint): gathered by every lane of the group, stored by its leader
BR_DEV void lg_write_checkpoint(const ChainTables& t, const GroupChain& c) {
  const uint32_t n_cmds = cold_get<kA_n_cmds>(c.a), n_lits = cold_get<kA_n_lits>(c.a), n_pushes = cold_get<kA_n_pushes>(c.a), n_bad = cold_get<kA_n_bad>(c.a);
  const uint32_t last_dist_code = cold_get<kA_last_dist_code>(c.a), last_copy_len = cold_get<kA_last_copy_len>(c.a), ext_len = cold_get<kA_ext_len>(c.a);
  const uint32_t tail_kind = cold_get<kA_tail_kind>(c.a), tail_base = cold_get<kA_tail_base>(c.a), tail_p1 = cold_get<kA_tail_p1>(c.a);
  const uint32_t mode = cold_get<kB_mode>(c.b), maxdef = cold_get<kB_maxdef>(c.b), vlookups = cold_get<kB_vlookups>(c.b), vwould = cold_get<kB_vwould>(c.b);
  const uint32_t vmaxdef = cold_get<kB_vmaxdef>(c.b), lookups0 = cold_get<kB_lookups0>(c.b), matches0 = cold_get<kB_matches0>(c.b);
  if (!LG_LEADER) return;
  // (field by field into memory: a record built on the stack first stays in scratch memory on the device)
  Checkpoint* r = t.checkpoints + c.next_cp / kCheckpointStride;
  r->pos = c.position;
  r->insert_len = c.insert_length;
  r->apply = c.apply;
  r->dc[0] = c.dc0;
  r->dc[1] = c.dc1;
  r->dc[2] = c.dc2;
  r->dc[3] = c.dc3;
  r->n_cmds = n_cmds;
  r->n_lits = n_lits;
  r->n_searches = c.n_searches;
  r->n_pushes = n_pushes;
  r->n_bad = n_bad;
  r->last_dist_code = last_dist_code;
  r->last_copy_len = last_copy_len;
  r->ext_len = ext_len;
  r->tail_kind = tail_kind;
  r->tail_base = tail_base;
  r->tail_p1 = tail_p1;
  r->d_lookups = c.lookups;
  r->d_matches = c.matches;
  r->d_mode = mode;
  r->d_maxdef = (int32_t)maxdef;
  r->d_vlookups = vlookups;
  r->d_vwould = vwould;
  r->d_vmaxdef = (int32_t)vmaxdef;
  r->entry_lookups = lookups0;
  r->entry_matches = matches0;
  r->no_dict = (c.st & kStNoDict) ? 1u : 0u;
  r->valid = kCheckpointValid;
  r->pad[0] = r->pad[1] = r->pad[2] = 0;
}

// ---- one step: one search and what follows from it
template <uint32_t kHtl>
BR_DEV void lg_step(const Lz77Params& P, const ChainTables& t, GroupChain& c, const uint32_t* dict_off) {
  const uint32_t pos_end = c.pos_end;
  const uint32_t window = P.spree_window;
  const bool lazy = (c.st & kStLazy) != 0;
  {
    // the loop top of CreateBackwardReferences
    const bool top = (c.st & kStActive) && !lazy;
    const bool done = top && !(c.position + kHtl < pos_end && c.position < c.seg_end);
    c.st = done ? (c.st & ~(uint32_t)kStActive) : c.st;
    // checkpoints: the first loop-top position at or behind every boundary (br_parse_segment)
    const bool cp_due = top && !done && c.next_cp <= c.position && c.next_cp < c.seg_end;
    if (LG_ANY(cp_due)) {
      if (cp_due) {
        while (c.next_cp <= c.position && c.next_cp < c.seg_end) {
          if (c.st & kStCheckpoints) lg_write_checkpoint(t, c);
          c.next_cp += kCheckpointStride;
        }
      }
    }
  }
  const bool on = (c.st & kStActive) != 0;
  const uint32_t cur = c.position + (lazy ? 1u : 0u);
  const SearchResult sr = lg_search<kHtl>(P, t, c, cur, on, dict_off);
  c.n_searches += on ? 1u : 0u;
  const bool fresh = on && !lazy, lz = on && lazy;
  const bool take = fresh && sr.found;                                     // a match: look at position + 1 before taking it (the next step)
  const bool lit = fresh && !sr.found;                                     // a literal
  const bool better = lz && sr.found && sr.score >= c.sr_score + 175;      // the lazy probe wins: the position in hand becomes a literal
  const bool flags_on = (c.st & kStFlags) != 0;
  // the position left behind by a literal or a lazy step was searched and is stored
  if (LG_ANY((lit || better) && flags_on)) {
    if ((lit || better) && flags_on && LG_LEADER && c.position < c.seg_end) t.flags_next[c.position] = (uint8_t)(kFlagStored | kFlagSearched);
  }
  {
    const bool nm = take || better;
    c.sr_len = nm ? sr.len : c.sr_len;
    c.sr_len_x_code = nm ? sr.len_x_code : c.sr_len_x_code;
    c.sr_distance = nm ? sr.distance : c.sr_distance;
    c.sr_score = nm ? sr.score : c.sr_score;
  }
  const uint32_t adv = (lit || better) ? 1u : 0u;
  c.position += adv;
  c.insert_length += adv;
  const uint32_t delayed = ((c.st & kStDelayedMask) >> kStDelayedShift) + (better ? 1u : 0u);  // (1..4 behind a lazy step that won)
  const bool again = better && delayed < 4 && c.position + kHtl < pos_end;  // probe the next position as well
  const bool emit = lz && !again;
  const uint32_t next_probed = (lz && !better) ? 1u : 0u;
  {
    uint32_t st = c.st & ~(uint32_t)(kStLazy | kStDelayedMask);
    st |= (take || again) ? kStLazy : 0u;
    st |= again ? ((delayed & 3u) << kStDelayedShift) : 0u;
    c.st = on ? st : c.st;
  }
  // ---- a literal spree (mod.rs:2529-2546): rare in compressible data
  const bool spree = lit && c.position > c.apply;
  if (LG_ANY(spree)) {
    if (spree) {
      const uint32_t margin = kHtl - 1 > 4 ? kHtl - 1 : 4;
      if (c.position + 16 >= pos_end - margin) {
        cold_set<kA_tail_kind>(c.a, (uint32_t)kHeadUnstored);
        cold_set<kA_tail_base>(c.a, c.position);
        lg_flag_unstored_range(t, c, c.position, pos_end);
        c.insert_length += pos_end - c.position;
        c.position = pos_end;
      } else {
        // Store4Vec4: position, +4, +8, +12 / StoreEvenVec4: position, +2, +4, +6
        const bool vec4 = c.position > c.apply + 4 * window;
        const uint32_t span = vec4 ? 16u : 8u, mask = vec4 ? 3u : 1u;
        cold_set<kA_tail_kind>(c.a, vec4 ? (uint32_t)kHeadVec4 : (uint32_t)kHeadEven4);
        cold_set<kA_tail_base>(c.a, c.position);
        if (flags_on) {
          LG_NOUNROLL
          for (uint32_t q = c.position + LG_IDX; q < c.position + span; q += LG_WIDTH)
            if (q < c.seg_end) t.flags_next[q] = ((q - c.position) & mask) == 0;
        }
        c.insert_length += span;
        c.position += span;
      }
    }
  }
  // ---- a command
  if (LG_ANY(emit)) {
    if (emit) {
      const uint32_t len = c.sr_len;
      c.apply = c.position + 2 * len + window;
      const uint32_t max_distance = c.position < P.max_backward_limit ? c.position : P.max_backward_limit;
      const int32_t dc_now[4] = {c.dc0, c.dc1, c.dc2, c.dc3};
      const uint32_t distance_code = br_compute_distance_code(c.sr_distance, max_distance, dc_now);
      const bool push = c.sr_distance <= max_distance && distance_code > 0;
      c.dc3 = push ? c.dc2 : c.dc3;
      c.dc2 = push ? c.dc1 : c.dc2;
      c.dc1 = push ? c.dc0 : c.dc1;
      c.dc0 = push ? (int32_t)c.sr_distance : c.dc0;
      const uint32_t n_cmds = cold_get<kA_n_cmds>(c.a), cmd_cap = cold_get<kA_cmd_cap>(c.a), cmd_base = cold_get<kA_cmd_base>(c.a);
      if (LG_LEADER && n_cmds < cmd_cap && flags_on)
        t.cmds[(size_t)cmd_base + n_cmds] = br_raw_command(c.insert_length, len, len ^ c.sr_len_x_code, distance_code);
      cold_add<kA_n_cmds>(c.a, 1u);
      cold_add<kA_n_pushes>(c.a, 1u, push);
      cold_add<kA_n_bad>(c.a, 1u, len < 2);
      cold_add<kA_n_lits>(c.a, c.insert_length);
      cold_set<kA_last_dist_code>(c.a, distance_code);
      cold_set<kA_last_copy_len>(c.a, len);
      // hash-table side effects: position searched, position + 1 only if probed, then StoreRange (the copy_range of FlagWriter
      // without masked entries: stored up to store_end) -- the group's lanes write the flags of the command side by side
      cold_set<kA_tail_kind>(c.a, (uint32_t)kHeadCopy);
      cold_set<kA_tail_base>(c.a, c.position);
      cold_set<kA_tail_p1>(c.a, next_probed);
      c.insert_length = 0;
      if (flags_on) {
        const uint32_t store_end = pos_end >= kHtl ? pos_end - kHtl + 1 : 0;
        uint32_t b = c.position + (len > 1 ? len : 1u);
        b = b > c.seg_end ? c.seg_end : b;
        LG_NOUNROLL
        for (uint32_t q = c.position + LG_IDX; q < b; q += LG_WIDTH) {
          const uint32_t k = q - c.position;
          const uint8_t rest = q < store_end ? (uint8_t)1 : lg_unstored(c, q);
          t.flags_next[q] = k == 0 ? (uint8_t)(kFlagStored | kFlagSearched) : (k == 1 ? (next_probed ? (uint8_t)(kFlagStored | kFlagSearched) : lg_unstored(c, q)) : rest);
        }
      }
      c.position += len;
    }
  }
}

// ---- the end of br_parse_segment: what is left of the block, the exit record
BR_DEV void lg_end(const ChainTables& t, GroupChain& c, SegExit& exit_out) {
  if (c.st & kStCheckpoints) {
    // boundaries this parse never reached at a loop top: whatever record sits there belongs to an older parse
    for (; c.next_cp < c.seg_end; c.next_cp += kCheckpointStride)
      if (LG_LEADER) t.checkpoints[c.next_cp / kCheckpointStride].valid = 0;
  }
  if (c.st & kStLastInBlock) {
    if (c.position < c.pos_end) lg_flag_unstored_range(t, c, c.position, c.pos_end);
    c.insert_length += c.pos_end - c.position;
    c.position = c.pos_end;
  }
  const uint32_t n_cmds = cold_get<kA_n_cmds>(c.a), n_lits = cold_get<kA_n_lits>(c.a), n_pushes = cold_get<kA_n_pushes>(c.a), n_bad = cold_get<kA_n_bad>(c.a);
  const uint32_t last_dist_code = cold_get<kA_last_dist_code>(c.a), last_copy_len = cold_get<kA_last_copy_len>(c.a), ext_len = cold_get<kA_ext_len>(c.a);
  const uint32_t tail_kind = cold_get<kA_tail_kind>(c.a), tail_base = cold_get<kA_tail_base>(c.a), tail_p1 = cold_get<kA_tail_p1>(c.a);
  const uint32_t mode = cold_get<kB_mode>(c.b), maxdef = cold_get<kB_maxdef>(c.b), vlookups = cold_get<kB_vlookups>(c.b), vwould = cold_get<kB_vwould>(c.b);
  const uint32_t vmaxdef = cold_get<kB_vmaxdef>(c.b), lookups0 = cold_get<kB_lookups0>(c.b), matches0 = cold_get<kB_matches0>(c.b);
  if (!LG_LEADER) return;
  SegExit& x = exit_out;  // (field by field into memory, like the checkpoints)
  x.pos = c.position;
  x.apply = c.apply;
  x.cache[0] = c.dc0;
  x.cache[1] = c.dc1;
  x.cache[2] = c.dc2;
  x.cache[3] = c.dc3;
  x.insert_len = c.insert_length;
  x.n_cmds = n_cmds;
  x.n_lits = n_lits;
  x.ext_len = ext_len;
  x.dict_lookups = mode == 2 ? c.lookups + vlookups : c.lookups;
  x.dict_matches = mode == 2 ? c.matches + vwould : c.matches;
  x.last_dist_code = last_dist_code;
  x.bad_commands = n_bad;
  x.n_searches = c.n_searches;
  x.last_copy_len = last_copy_len;
  x.dict_mode = mode;
  x.dict_maxdef = (int32_t)(mode == 2 ? vmaxdef : maxdef);
  x.n_pushes = n_pushes < 4 ? n_pushes : 4u;
  x.tail_kind = c.position > c.seg_end ? tail_kind : (uint32_t)kHeadNone;
  x.tail_base = c.position > c.seg_end ? tail_base : 0u;
  x.tail_p1 = c.position > c.seg_end ? tail_p1 : 0u;
  x.n_pushes_all = n_pushes;
  x.dict_entry_lookups = lookups0;
  x.dict_entry_matches = matches0;
}

// One chain from its entry to its exit.  The host emulation calls this per segment; on the device the 16 lanes of a group run it
// together (`have`: the group has a segment of its own) and the four groups of a wavefront side by side.
template <uint32_t kHtl>
BR_DEV void br_group_parse(const Lz77Params& P, const ChainTables& t, const Segment& seg, const SegEntry& entry, SegExit& exit_out, bool have,
                           const uint32_t* dict_off, uint32_t* walked, uint32_t* searches, uint32_t* commands) {
  // (a group without a segment -- in the last wavefront of a launch -- is set up on the launch's first segment with its writes off,
  // and idles)
  GroupChain c;
  lg_begin<kHtl>(P, t, seg, entry, c, have);
  if (!have) c.st &= ~(uint32_t)kStActive;
  while (LG_ANY((c.st & kStActive) != 0)) lg_step<kHtl>(P, t, c, dict_off);
  const uint32_t walked_from = cold_get<kA_walked_from>(c.a);
  if (have) lg_end(t, c, exit_out);
  const uint32_t n_cmds = cold_get<kA_n_cmds>(c.a);
  *walked = have ? c.position - walked_from : 0u;
  *searches = have ? c.n_searches : 0u;
  *commands = (have && (c.st & kStFlags)) ? n_cmds : 0u;
}

}  // namespace brotli_mi355x
#endif
