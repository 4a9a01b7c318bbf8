// quick_spec.h -- qualities 2, 3 and 4 on the speculative path (round 6): the segments of a block side by side.
// Shared by the gfx950 kernels (quick_spec_kernels.hip) and the host emulation of the device seam (tests/emu, test infrastructure).
//
// quick_device.h walks the reference's own BasicHasher table, one wavefront per stream.  Here the table is replaced by what it
// WOULD hold: a BasicHasher slot is overwritten by every filing into it (mod.rs:322-327), filings happen in position order (a search
// files its own position after it has looked, mod.rs:359-473; a copy files its range before the next search, mod.rs:2516-2521;
// a literal spree every second or fourth position, mod.rs:2538-2545), so the content of slot s when position p is searched is the
// value filed by the LATEST position q < p that was filed under s.  Which positions are filed, and for the quads of StoreRange
// (mod.rs:299-321) under which slot offset and as which value, is written down per position by the chains themselves (one flag
// byte); the candidates of every position follow from the flags through one sort of the potential filings (see quick_api.h).
//
// The chain below is CreateBackwardReferences (mod.rs:2376-2552) over one SEGMENT of a block from a guessed entry state, in the
// frame of lz77_chain.h's br_parse_segment: the same SegEntry / SegExit records, the same rule for who writes which flag (every
// position's flag is written by the chain of the segment it lies in; the part of a chain's last step that reaches into the next
// segment is described by the exit and flagged by the next chain), the same static-dictionary books (DictState), so that the host
// resolver (Lz77Stage::Resolve) chains exits into entries exactly as for qualities 5-9.
#ifndef BROTLI_MI355X_QUICK_SPEC_H_
#define BROTLI_MI355X_QUICK_SPEC_H_

#include "quick_device.h"

namespace brotli_mi355x {

// ---- filings ------------------------------------------------------------------------------------------------------------------
// HashBytes (mod.rs:437-441) for any thread (q_key of quick_device.h is the wave-uniform form)
BR_DEV uint32_t qs_hash(const QuickJob& J, const uint8_t* p) {
  const uint64_t v = (br_load64(p) << (64u - 8u * J.hash_len)) * kQuickHashMul64;
  return (uint32_t)(v >> (64u - J.bucket_bits));
}
// what position q filed, and under which slot offset, given its flag byte: Store (mod.rs:322-327) files q under (q >> 3) % sweep;
// a quad of StoreRange (mod.rs:299-321) that starts at `at` = q - b files (at & mask) + b under ((at & mask) >> 3) % sweep
BR_DEV void qs_filing(const QuickJob& J, const Lz77Params& P, uint32_t q, uint8_t f, uint32_t* value, uint32_t* off) {
  if (f & kQsQuad) {
    const uint32_t b = (f >> 3) & 3u;
    const uint32_t pm = (q - b) & P.ring_mask;
    *value = pm + b;
    *off = (pm >> 3) & (J.sweep - 1u);
  } else {
    *value = q;
    *off = (q >> 3) & (J.sweep - 1u);
  }
}
BR_DEV uint32_t qs_event_position(const QuickJob& J, uint32_t id) { return J.sweep == 1 ? id : id >> 1; }
// slot of potential filing `id`: position q under its own offset, or (odd ids, sweep > 1) under the offset of the 8-byte group in front
BR_DEV uint32_t qs_event_slot(const QuickJob& J, const uint8_t* text, uint32_t id) {
  const uint32_t q = qs_event_position(J, id);
  const uint32_t own = (q >> 3) & (J.sweep - 1u);
  const uint32_t off = (J.sweep != 1 && (id & 1u)) ? ((own + J.sweep - 1u) & (J.sweep - 1u)) : own;
  return qs_hash(J, text + q) + off;
}
// first event of slot [lo, hi) whose position is >= p
BR_DEV uint32_t qs_rank_in_slot(const QuickJob& J, const uint32_t* ev_id, uint32_t lo, uint32_t hi, uint32_t p) {
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (qs_event_position(J, ev_id[mid]) < p) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// the same, started where p would sit if the slot's positions were spread evenly over the text of n bytes (they nearly are: a
// handful of neighbouring probes instead of log2 of the slot's length, all in one or two cache lines; index build, 256 M searches at
// 64 MiB and four slots per key)
BR_DEV uint32_t qs_rank_in_slot_guess(const QuickJob& J, const uint32_t* ev_id, uint32_t lo, uint32_t hi, uint32_t p, uint32_t n) {
  if (hi - lo <= 8u) {
    while (lo < hi && qs_event_position(J, ev_id[lo]) < p) ++lo;
    return lo;
  }
  uint32_t g = lo + (uint32_t)(((uint64_t)(hi - lo) * p) / n);
  if (g >= hi) g = hi - 1u;
  uint32_t step = 4u;
  if (qs_event_position(J, ev_id[g]) < p) {
    // the answer lies behind g: gallop forward to an event at or behind p
    uint32_t a = g + 1u;
    for (;;) {
      const uint32_t b = a + step < hi ? a + step : hi;
      if (b == hi || qs_event_position(J, ev_id[b]) >= p) return qs_rank_in_slot(J, ev_id, a, b, p);
      a = b + 1u;
      step <<= 1;
    }
  }
  // the answer lies at or in front of g: gallop backward to an event in front of p
  uint32_t b = g;
  for (;;) {
    if (b - lo <= step) return qs_rank_in_slot(J, ev_id, lo, b, p);
    const uint32_t a = b - step;
    if (qs_event_position(J, ev_id[a]) < p) return qs_rank_in_slot(J, ev_id, a + 1u, b, p);
    b = a;
    step <<= 1;
  }
}
// event index of the filing (q, off): the position's own-offset event or (sweep > 1) the displaced one
BR_DEV uint32_t qs_event_of(const QuickJob& J, const QuickSpec& S, uint32_t q, uint32_t off) {
  if (J.sweep == 1) return S.ev_of[q];
  return S.ev_of[2u * q + (off != ((q >> 3) & (J.sweep - 1u)) ? 1u : 0u)];
}
// the pass over everything, position q: its potential filings switched on / off
BR_DEV void qs_item_activate(const QuickJob& J, const Lz77Params& P, const QuickSpec& S, uint32_t q) {
  const uint8_t f = S.flags[q];
  S.flags_prev[q] = f;
  uint32_t value, off;
  qs_filing(J, P, q, f, &value, &off);
  const bool stored = (f & kQsStored) != 0;
  if (J.sweep == 1) {
    const uint32_t e = S.ev_of[q];
    S.actraw[e] = stored ? e + 1u : 0u;
    S.val[e] = value;
    return;
  }
  const bool displaced = off != ((q >> 3) & (J.sweep - 1u));
  const uint32_t e0 = S.ev_of[2u * q], e1 = S.ev_of[2u * q + 1u];
  S.actraw[e0] = (stored && !displaced) ? e0 + 1u : 0u;
  S.val[e0] = value;
  S.actraw[e1] = (stored && displaced) ? e1 + 1u : 0u;
  S.val[e1] = value;
}
// candidate j of position p: the value of the latest active event in front of p's rank in slot key(p) + j (act: inclusive max-scan
// of actraw, over everything or over the slot alone); none = the zeroed table (encode.rs:1147), which IS a candidate: text position 0
BR_DEV uint32_t qs_candidate(const QuickJob& J, const QuickSpec& S, uint32_t slot, uint32_t rank) {
  const uint32_t e = rank == 0 ? 0u : S.act[rank - 1u];
  if (e == 0 || e - 1u < S.slot_first[slot]) return S.base ? S.base[slot] : 0u;  // (a later piece of a stream: what the table held in front of the text)
  return S.val[e - 1u];
}
// incremental update, position q of a segment that was parsed again: if its filing changed the events are switched and handed to `mark`
template <typename Mark>
BR_DEV void qs_item_diff(const QuickJob& J, const Lz77Params& P, const QuickSpec& S, uint32_t q, Mark mark) {
  const uint8_t f = S.flags[q], g = S.flags_prev[q];
  if (f == g) return;
  S.flags_prev[q] = f;
  const uint8_t filing_bits = (uint8_t)(kQsStored | kQsQuad | 0x18u);
  if (((f ^ g) & filing_bits) == 0) return;
  uint32_t v_new, off_new, v_old, off_old;
  qs_filing(J, P, q, f, &v_new, &off_new);
  qs_filing(J, P, q, g, &v_old, &off_old);
  const bool s_new = (f & kQsStored) != 0, s_old = (g & kQsStored) != 0;
  if (!s_new && !s_old) return;
  const uint32_t e_new = qs_event_of(J, S, q, off_new), e_old = qs_event_of(J, S, q, off_old);
  if (s_old && s_new && e_old == e_new && v_old == v_new) return;
  if (s_old) {
    S.actraw[e_old] = 0u;
    if (!(s_new && e_new == e_old)) mark(e_old);
  }
  if (s_new) {
    S.actraw[e_new] = e_new + 1u;
    S.val[e_new] = v_new;
    mark(e_new);
  }
}
// incremental update, listed event n: the scan entries from it up to the next active event of its slot, from actraw alone; the slot
// and the positions (lo, hi] whose candidate in that slot reads a rewritten entry go to chg_range.  false: a walk gave up.
BR_DEV bool qs_item_repair(const QuickJob& J, const QuickSpec& S, uint32_t n) {
  const uint32_t e = S.chg_list[n];
  const uint32_t slot = S.ev_slot[e];
  const uint32_t lo = S.slot_first[slot], hi = S.slot_first[slot + 1];
  uint32_t v = S.actraw[e];
  uint32_t steps = 0;
  if (v == 0) {
    for (uint32_t i = e; i > lo;) {
      --i;
      const uint32_t a = S.actraw[i];
      if (a != 0) {
        v = a;
        break;
      }
      if (++steps > kQsWalkCap) return false;
    }
  }
  uint32_t i = e;
  S.act[i] = v;
  ++i;
  while (i < hi && S.actraw[i] == 0) {
    S.act[i] = v;
    ++i;
    if (++steps > kQsWalkCap) return false;
  }
  S.chg_range[3u * n] = slot;
  S.chg_range[3u * n + 1u] = qs_event_position(J, S.ev_id[e]);
  S.chg_range[3u * n + 2u] = i < hi ? qs_event_position(J, S.ev_id[i]) : 0xffffffffu;
  return true;
}
// incremental update, listed event n and slot `t` of the 2 sweep - 1 around its own: the positions in (lo, hi] whose own-offset event
// lies in t and that look into the event's slot get that candidate derived again; `changed(p, was, now)` for those that differ
template <typename Changed>
BR_DEV bool qs_item_recand(const QuickJob& J, const QuickSpec& S, uint32_t n, uint32_t t, Changed changed) {
  const uint32_t slot = S.chg_range[3u * n], p_lo = S.chg_range[3u * n + 1u], p_hi = S.chg_range[3u * n + 2u];
  const uint32_t hi = S.slot_first[t + 1];
  uint32_t i = p_lo == 0xffffffffu ? hi : qs_rank_in_slot(J, S.ev_id, S.slot_first[t], hi, p_lo + 1u);
  uint32_t steps = 0;
  for (; i < hi; ++i) {
    const uint32_t id = S.ev_id[i];
    const uint32_t p = qs_event_position(J, id);
    if (p > p_hi) break;
    // (a stretch without a filing in one slot can face millions of positions of a neighbouring one -- runs of one byte, whose
    // positions an extended copy leaves unfiled: the pass over everything is the cheaper one then)
    if (++steps > kQsWalkCap) return false;
    if (J.sweep != 1 && (id & 1u)) continue;
    const uint32_t key = t - ((p >> 3) & (J.sweep - 1u));
    const uint32_t j = slot - key;  // (wraps when the key lies behind the slot)
    if (j >= J.sweep) continue;
    const size_t item = (size_t)p * J.sweep + j;
    const uint32_t c = qs_candidate(J, S, slot, S.qrank[item]);
    const uint32_t was = S.cand[item];
    if (c == was) continue;
    S.cand[item] = c;
    changed(p, was, c);
  }
  return true;
}

// A candidate of position p changed.  The chains that looked at it: the chain of p's segment if p carries the searched flag, and for
// the first 8 positions of a segment the chain in front whatever the flag says (a lazy probe runs up to five positions ahead of
// the loop-top position, mod.rs:2455-2480, and the flag of such a position is written by the NEXT chain, possibly a round later).
BR_DEV void qs_note_changed(const QuickSpec& S, uint32_t p, const SegGeometry& geo, uint8_t* dirty) {
  if (p < geo.first_block_start) return;
  const uint32_t blk = (p - geo.prefix_bytes) / geo.block_bytes;
  const uint32_t bs = blk == 0 ? geo.first_block_start : geo.prefix_bytes + blk * geo.block_bytes;
  const uint32_t off = p - bs;
  const uint32_t seg_bytes = geo.block_segment_bytes[blk];
  const uint32_t first = geo.block_first_segment[blk];
  uint32_t k = first + off / seg_bytes;
  if (k >= geo.block_first_segment[blk + 1]) k = geo.block_first_segment[blk + 1] - 1;
  if (S.flags[p] & kQsSearched) dirty[k] = 1;
  if (k > first && off - (k - first) * seg_bytes < 8) dirty[k - 1] = 1;
}

// Can it matter to the search at position p which of two values a slot holds?  A hasher with several slots per key (H3, H4, H54)
// skips a candidate that fails a test (mod.rs:391-440) and consults the static dictionary only when nothing was found, so a candidate
// takes part only if it lies inside the window and starts with the four bytes at p (FindMatchLengthWithLimitMin4,
// static_dict.rs:134-147): a change between two values that both fail that is no change.  The single-slot hasher (H2) RETURNS on a
// failed distance or quick-reject test without consulting the dictionary (mod.rs:359-390), so there the value shows in the books of
// the throttle even when it matches nothing -- unless the dictionary is not consulted at all.
BR_DEV bool qs_candidate_plausible(const Lz77Params& P, const uint8_t* text, uint32_t p, uint32_t v) {
  const uint32_t backward = p - v;
  const uint32_t max_backward = p < P.max_backward_limit ? p : P.max_backward_limit;
  if (backward == 0 || backward > max_backward) return false;
  return br_load32(text + v) == br_load32(text + p);
}
BR_DEV bool qs_change_matters(const QuickJob& J, const Lz77Params& P, const uint8_t* text, uint32_t p, uint32_t old_value, uint32_t new_value) {
  if (J.sweep == 1 && J.use_dictionary) return true;
  return qs_candidate_plausible(P, text, p, old_value) || qs_candidate_plausible(P, text, p, new_value);
}

// First guess of the flag of position q (lz77_qspec_init_flags): a custom dictionary is filed position by position but for its
// last 7 bytes (encode.rs:1163-1194, mod.rs:224-229); the last three positions in front of a block of >= 7 bytes are filed when
// that block starts (StitchToPreviousBlock, mod.rs:210-222), the four in front of them never (store_end, mod.rs:2397-2404; a
// search needs 8 bytes); everything else is assumed filed.
BR_DEV uint8_t qs_initial_flag(const Lz77Params& P, uint32_t q, uint32_t first_block_start, bool prefix_is_dictionary) {
  const uint32_t total = P.total_bytes, pre = P.prefix_bytes, bb = P.block_bytes;
  if (q < first_block_start) {
    bool stored = prefix_is_dictionary && pre > kQuickHtl - 1u && q < pre - (kQuickHtl - 1u);
    const uint32_t be = total < pre + bb ? total : pre + bb;
    if (be - first_block_start >= kQuickHtl - 1u && first_block_start >= 3 && q + 3 >= first_block_start) stored = true;
    return stored ? kQsStored : (uint8_t)0;
  }
  const uint32_t b = (q - pre) / bb;
  const uint64_t be64 = (uint64_t)pre + ((uint64_t)b + 1u) * bb;
  const uint32_t be = be64 < total ? (uint32_t)be64 : total;
  if (q + (kQuickHtl - 1u) < be) return kQsStored;
  if (be >= total) return 0;
  const uint32_t nbytes = (total - be) < bb ? total - be : bb;
  const bool stitched = nbytes >= kQuickHtl - 1u && be >= 3;
  return (stitched && q + 3 >= be) ? kQsStored : (uint8_t)0;
}

// ---- the chain ----------------------------------------------------------------------------------------------------------------
struct QsTables {
  const uint8_t* text;
  const uint32_t* cand;  // [n * sweep]
  uint8_t* flags;
  QuickTables dict;
  // One chain per block on a table of its own (lz77_qspec_block_tables: what the slots hold at the block's start under the flags of
  // everything in front of it): the chain reads the slots and files into them as the reference does, so that what it finds inside its
  // own block is exact in the round it is parsed -- the candidates of a position stand for the flags of the round BEFORE, the chain's
  // own block included, and a chain of 16 KiB on them settled a few hundred bytes per launch.  Null: the candidates.
  uint32_t* own = nullptr;
  uint32_t own_stride = 0;  // words from one chain's table to the next (the launch sets `own` to the chain's table)
};
// the slot `slot` of a chain's own table / the filing of position ix into it (Store, mod.rs:322-327)
BR_DEV uint32_t qs_own_get(const QsTables& T, uint32_t slot) { return BR_LIVE_LD32(T.own + slot); }
BR_DEV void qs_own_store(const QuickJob& J, const QsTables& T, uint32_t ix) {
  const uint32_t slot = q_key(J, T.text + ix) + ((ix >> 3) & (J.sweep - 1u));
  if (BR_LANE == 0) BR_LIVE_ST32(T.own + slot, ix);
}

struct QsFlagWriter {
  uint8_t* flags;
  uint32_t lo, hi;   // the chain's own segment: writes outside are dropped
  uint32_t tail_lo;  // positions >= tail_lo of the block are filed by the NEXT block's StitchToPreviousBlock (tail_value) or never
  uint8_t tail_value;
  BR_DEV void put(uint32_t q, uint8_t v) {
    if (q >= lo && q < hi) flags[q] = v;
  }
  BR_DEV void one(uint32_t q, uint8_t v) {
    if (BR_LANE == 0) put(q, v);
  }
  BR_DEV uint8_t unstored(uint32_t q) const { return q >= tail_lo ? tail_value : (uint8_t)0; }
  BR_DEV void unstored_range(uint32_t a, uint32_t b) {
    if (b > hi) b = hi;
    for (uint32_t q = (a > lo ? a : lo) + BR_LANE; q < b; q += BR_NLANES) flags[q] = unstored(q);
  }
  // StoreRange(first, min(copy_end, store_end)), mod.rs:299-321: runs of 16 and more go four at a time
  BR_DEV uint8_t copy_value(uint32_t q, uint32_t first, uint32_t copy_end, uint32_t store_end) const {
    const uint32_t last = copy_end < store_end ? copy_end : store_end;
    if (q >= last) return unstored(q);
    if (last >= first + 16u && q < first + ((last - first) & ~3u)) return (uint8_t)(kQsStored | kQsQuad | (((q - first) & 3u) << 3));
    return kQsStored;
  }
  BR_DEV void copy_range(uint32_t first, uint32_t copy_end, uint32_t store_end) {
    const uint32_t b = copy_end > hi ? hi : copy_end;
    for (uint32_t q = (first > lo ? first : lo) + BR_LANE; q < b; q += BR_NLANES) flags[q] = copy_value(q, first, copy_end, store_end);
  }
  // the part [a, b) of the step (kind, base, p1) -- HeadKind of lz77_types.h; for a copy b is where it ends
  BR_DEV void head(uint32_t kind, uint32_t base, uint32_t p1, uint32_t a, uint32_t b, uint32_t store_end) {
    if (kind == kHeadNone) return;
    const uint32_t step_end = b;
    if (b > hi) b = hi;
    for (uint32_t q = (a > lo ? a : lo) + BR_LANE; q < b; q += BR_NLANES) {
      uint8_t v;
      if (kind == kHeadCopy) {
        if (q <= base) v = (uint8_t)(kQsStored | kQsSearched);  // lazily delayed literals and the start of the match: searched
        else if (q == base + 1) v = (p1 & 1u) ? (uint8_t)(kQsStored | kQsSearched) : unstored(q);
        else v = copy_value(q, base + 2, step_end, store_end);
      } else if (kind == kHeadUnstored) {
        v = unstored(q);
      } else if (kind == kHeadVec4) {
        v = ((q - base) & 3) == 0 ? kQsStored : (uint8_t)0;
      } else {
        v = ((q - base) & 1) == 0 ? kQsStored : (uint8_t)0;
      }
      flags[q] = v;
    }
  }
};

// SearchInStaticDictionary (shallow) under the books of lz77_chain.h's DictState: the same modes and deficits as
// br_dictionary_stage, one hash item per search (mod.rs:1942-1988, 1896-1940)
BR_DEV bool qs_search_dictionary(const Lz77Params& P, const QuickTables& T, DictState& ds, bool no_dict, const uint8_t* text, uint32_t cur, uint32_t max_length,
                                 uint32_t max_backward, QuickResult& out) {
  const bool dead = ds.matches < (ds.lookups >> 7);
  const uint32_t seen = dead ? 2u : 1u;
  ds.mode = (ds.mode == 0 || ds.mode == seen) ? seen : 3u;
  if (no_dict) {
    ds.mode = 4;
    return false;
  }
  if (dead && ds.vwould) return false;
  if (dead) {
    if ((int32_t)ds.vlookups > ds.vmaxdef) ds.vmaxdef = (int32_t)ds.vlookups;
  } else {
    const int32_t def = (int32_t)(ds.lookups - ds.lookups0) - 128 * (int32_t)(ds.matches - ds.matches0);
    if (def > ds.maxdef) ds.maxdef = def;
  }
  const uint32_t key = ((br_load32(text + cur) * 0x1e35a7bdu) >> (32 - 14)) << 1;  // Hash14 << 1
  const uint32_t item = T.dict_hash[key];
  if (dead) ds.vlookups++; else ds.lookups++;
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
  if (dead) {
    ds.vwould = 1;
    return false;
  }
  out.len = matchlen;
  out.len_x_code = len ^ matchlen;
  out.distance = backward;
  out.score = score;
  ds.matches++;
  return true;
}

// BasicHasher::FindLongestMatch, mod.rs:359-473, with the slots read from the candidates of `cur` (q_find_longest_match of
// quick_device.h on the table; the position is filed by the flag the caller writes)
#if !BR_SCALAR
BR_DEV bool qs_find_longest_match_lanes(const QuickJob& J, const Lz77Params& P, const QsTables& T, DictState& ds, bool no_dict, int32_t dc0, uint32_t cur,
                                        uint32_t max_length, uint32_t max_backward, uint32_t blk_end, QuickResult& out);
#endif
BR_DEV bool qs_find_longest_match_slots(const QuickJob& J, const Lz77Params& P, const QsTables& T, DictState& ds, bool no_dict, int32_t dc0, uint32_t cur,
                                        uint32_t max_length, uint32_t max_backward, uint32_t blk_end, QuickResult& out);
BR_DEV bool qs_find_longest_match(const QuickJob& J, const Lz77Params& P, const QsTables& T, DictState& ds, bool no_dict, int32_t dc0, uint32_t cur,
                                  uint32_t max_length, uint32_t max_backward, uint32_t blk_end, QuickResult& out) {
  const bool found = qs_find_longest_match_slots(J, P, T, ds, no_dict, dc0, cur, max_length, max_backward, blk_end, out);
  // (a search files its own position, whatever it found: mod.rs:389, 400, 471 -- behind the look at the slots, one of which is its own)
  if (T.own) qs_own_store(J, T, cur);
  return found;
}
BR_DEV bool qs_find_longest_match_slots(const QuickJob& J, const Lz77Params& P, const QsTables& T, DictState& ds, bool no_dict, int32_t dc0, uint32_t cur,
                                        uint32_t max_length, uint32_t max_backward, uint32_t blk_end, QuickResult& out) {
#if !BR_SCALAR && !defined(BR_QS_NO_LANES)
  if (J.sweep != 1) return qs_find_longest_match_lanes(J, P, T, ds, no_dict, dc0, cur, max_length, max_backward, blk_end, out);
#endif
  const uint8_t* text = T.text;
  const uint32_t best_len_in = out.len;
  uint8_t compare_char = q_byte(P, text, cur + best_len_in, blk_end);
  uint32_t best_score = out.score, best_len = best_len_in;
  bool found = false;
  out.len_x_code = 0;
  if (dc0 > 0 && (uint32_t)dc0 <= cur) {
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
        if (J.sweep == 1) return true;
        found = true;
      }
    }
  }
  const uint32_t* cand = T.cand + (size_t)cur * J.sweep;
  const uint32_t own_key = T.own ? q_key(J, text + cur) : 0u;
  if (J.sweep == 1) {
    const uint32_t prev = BR_UNIFORM(T.own ? qs_own_get(T, own_key) : cand[0]);
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
      const uint32_t prev = BR_UNIFORM(T.own ? qs_own_get(T, own_key + j) : cand[j]);
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
  if (J.use_dictionary && !found) found = qs_search_dictionary(P, T.dict, ds, no_dict, text, cur, max_length, max_backward, out);
  return found;
}

#if !BR_SCALAR
// The same search for the hashers with several slots per key, the candidates side by side (device; the emulation runs the loop
// above): lane 0 takes the last distance, lanes 1 .. sweep the slots -- window test, first four bytes, common prefix, all of their
// loads in flight together -- and the fold then walks the candidates in the reference's order on the lanes' results.  The
// quick-reject byte of the loop above (the candidate's byte at the best length so far) follows from the prefix: equal while the best
// length is shorter than the candidate's prefix, different when it equals it inside the block; only a candidate SHORTER than the best so
// far (or a prefix that runs to the end of the block) needs its byte looked at.
BR_DEV bool qs_find_longest_match_lanes(const QuickJob& J, const Lz77Params& P, const QsTables& T, DictState& ds, bool no_dict, int32_t dc0, uint32_t cur,
                                        uint32_t max_length, uint32_t max_backward, uint32_t blk_end, QuickResult& out) {
  const uint8_t* text = T.text;
  const uint32_t lane = (uint32_t)BR_LANE;
  const uint32_t best_len_in = out.len;
  uint32_t prev = 0;
  bool valid = false;
  if (lane == 0) {
    if (dc0 > 0 && (uint32_t)dc0 <= cur) {
      prev = cur - (uint32_t)dc0;
      valid = true;
    }
  } else if (lane <= J.sweep) {
    prev = T.own ? qs_own_get(T, q_key(J, text + cur) + (lane - 1u)) : T.cand[(size_t)cur * J.sweep + (lane - 1u)];
    const uint32_t backward = cur - prev;
    valid = backward != 0 && backward <= max_backward;
  }
  uint32_t raw = 0;
  if (valid && br_load32(text + prev) == br_load32(text + cur)) raw = br_match_len_wide(text + prev, text + cur, max_length);
  if (raw < 4) raw = 0;  // FindMatchLengthWithLimitMin4
  const uint32_t len = raw ? q_fix_len(P, raw, prev) : 0u;
  uint8_t compare_char = q_byte(P, text, cur + best_len_in, blk_end);
  uint32_t best_score = out.score, best_len = best_len_in;
  bool found = false;
  out.len_x_code = 0;
  for (uint32_t c = 0; c <= J.sweep; ++c) {
    const uint32_t raw_c = BR_READLANE(raw, c);
    if (raw_c == 0) continue;
    const uint32_t prev_c = BR_READLANE(prev, c), len_c = BR_READLANE(len, c);
    bool pass;
    if (best_len < raw_c) pass = true;
    else if (best_len == raw_c && raw_c < max_length) pass = false;
    else pass = compare_char == text[prev_c + best_len];
    if (!pass) continue;
    if (c == 0) {
      best_score = q_score_last(P, len_c);
      best_len = len_c;
      out.len = len_c;
      out.distance = (uint32_t)dc0;
      out.score = best_score;
      compare_char = q_byte(P, text, cur + best_len, blk_end);
      found = true;
    } else {
      const uint32_t backward = cur - prev_c;
      const uint32_t score = q_score(P, len_c, backward);
      if (best_score < score) {
        best_score = score;
        best_len = len_c;
        out.len = len_c;
        out.distance = backward;
        out.score = score;
        compare_char = q_byte(P, text, cur + best_len, blk_end);
        found = true;
      }
    }
  }
  if (J.use_dictionary && !found) found = qs_search_dictionary(P, T.dict, ds, no_dict, text, cur, max_length, max_backward, out);
  return found;
}
#endif

// One chain: segment `seg_in` from `entry`; commands into its slab, flags of its own positions, `exit_out`.
BR_DEV void br_quick_segment(const QuickJob& J, const Lz77Params& P, const QsTables& T, const Segment& seg_in, const SegEntry& entry, Command* slab,
                             SegExit* exit_out) {
  const uint8_t* text = T.text;
  const uint32_t pos_end = BR_UNIFORM(seg_in.blk_end);
  const uint32_t seg_start = BR_UNIFORM(seg_in.start), seg_end = BR_UNIFORM(seg_in.end), seg_flags = BR_UNIFORM(seg_in.flags);
  const uint32_t blk_start = BR_UNIFORM(seg_in.blk_start), cmd_cap = BR_UNIFORM(seg_in.cmd_cap);
  const uint32_t window = P.spree_window;
  uint32_t position = BR_UNIFORM(entry.pos);
  uint32_t apply = BR_UNIFORM(entry.apply);
  uint32_t insert_length = 0;  // local literals (what was pending at the entry is added by the host fix-up, CmdPatch kind 2)
  int32_t dc[4];
  for (int i = 0; i < 4; ++i) dc[i] = (int32_t)BR_UNIFORM(entry.cache[i]);
  DictState ds;
  ds.lookups = ds.lookups0 = BR_UNIFORM(entry.dict_lookups);
  ds.matches = ds.matches0 = BR_UNIFORM(entry.dict_matches);
  ds.mode = 0;
  ds.maxdef = -(1 << 30);
  ds.vlookups = 0;
  ds.vwould = 0;
  ds.vmaxdef = -(1 << 30);
  // (once the throttle has tripped under exact counters it stays tripped: no lookups, no virtual books -- "ran blind", mode 4)
  const bool no_dict = J.use_dictionary && BR_UNIFORM(entry.dict_exact) && ds.matches < (ds.lookups >> 7);
  const bool dry = (seg_flags & kSegWarmup) != 0;  // a dry run over the tail of a segment: only the exit matters, nothing is written
  QuickJob Jo = J;  // (the chain's own table as the table of q_store_range)
  Jo.table = T.own;
  QsFlagWriter fw;
  fw.flags = T.flags;
  fw.lo = dry ? 0u : seg_start;
  fw.hi = dry ? 0u : seg_end;
  fw.tail_lo = pos_end - 3;
  fw.tail_value = (seg_flags & kSegTailStitched) ? kQsStored : (uint8_t)0;
  uint32_t tail_kind = kHeadNone, tail_base = 0, tail_p1 = 0;
  uint32_t n_cmds = 0, n_lits = 0, n_searches = 0, ext_len = 0, n_pushes = 0, n_bad = 0;
  uint32_t last_dist_code = 0xffffffffu, last_copy_len = 0;
  if (seg_flags & kSegFirstInBlock) {
    position = blk_start;
    if (BR_UNIFORM(entry.ext_allowed)) {
      // extend_last_command, encode.rs:360-400 (the resolver has checked everything but the bytes)
      const uint32_t d = (uint32_t)dc[0];
      const uint32_t n = BR_UNIFORM(br_match_len_wide(text + position, text + position - d, pos_end - position));
      ext_len = n;
      fw.unstored_range(position, position + n);
      tail_kind = kHeadUnstored;
      tail_base = position;
      position += n;
    }
    apply = position + window;
  }
  const uint32_t store_end = pos_end >= kQuickHtl ? pos_end - kQuickHtl + 1u : 0u;  // mod.rs:2397-2404
  if (!(seg_flags & kSegFirstInBlock)) {
    // the part of the previous chain's last step that lies in this segment
    tail_kind = BR_UNIFORM(entry.head_kind);
    tail_base = BR_UNIFORM(entry.head_base);
    tail_p1 = BR_UNIFORM(entry.head_p1);
    if (position > seg_start) fw.head(tail_kind, tail_base, tail_p1, seg_start, position, store_end);
  }
  while (position + kQuickHtl < pos_end && position < seg_end) {
    uint32_t max_length = pos_end - position;
    uint32_t max_distance = position < P.max_backward_limit ? position : P.max_backward_limit;
    QuickResult sr;
    sr.len = 0;
    sr.len_x_code = 0;
    sr.distance = 0;
    sr.score = kQuickMinScore;
    n_searches++;
    if (qs_find_longest_match(J, P, T, ds, no_dict, dc[0], position, max_length, max_distance, pos_end, sr)) {
      uint32_t delayed = 0;
      bool next_probed;
      max_length--;
      for (;;) {
        QuickResult sr2;
        sr2.len = sr.len - 1u < max_length ? sr.len - 1u : max_length;  // quality < 5, mod.rs:2450-2454
        sr2.len_x_code = 0;
        sr2.distance = 0;
        sr2.score = kQuickMinScore;
        max_distance = position + 1u < P.max_backward_limit ? position + 1u : P.max_backward_limit;
        n_searches++;
        next_probed = true;
        const bool is_match_found = qs_find_longest_match(J, P, T, ds, no_dict, dc[0], position + 1u, max_length, max_distance, pos_end, sr2);
        if (is_match_found && sr2.score >= sr.score + 175u) {  // cost_diff_lazy
          fw.one(position, (uint8_t)(kQsStored | kQsSearched));
          position++;
          insert_length++;
          sr = sr2;
          next_probed = false;
          if (++delayed < 4 && position + kQuickHtl < pos_end) {
            max_length--;
            continue;
          }
        }
        break;
      }
      apply = position + 2u * sr.len + window;
      max_distance = position < P.max_backward_limit ? position : P.max_backward_limit;
      const uint32_t distance_code = br_compute_distance_code(sr.distance, max_distance, dc);
      if (sr.distance <= max_distance && distance_code > 0) {
        dc[3] = dc[2];
        dc[2] = dc[1];
        dc[1] = dc[0];
        dc[0] = (int32_t)sr.distance;
        n_pushes++;
      }
      if (BR_LANE == 0 && n_cmds < cmd_cap && !dry) slab[n_cmds] = br_raw_command(insert_length, sr.len, sr.len ^ sr.len_x_code, distance_code);
      if (sr.len < 2) n_bad++;
      ++n_cmds;
      n_lits += insert_length;
      last_dist_code = distance_code;
      last_copy_len = sr.len;
      insert_length = 0;
      tail_kind = kHeadCopy;
      tail_base = position;
      tail_p1 = next_probed ? 1u : 0u;
      fw.one(position, (uint8_t)(kQsStored | kQsSearched));
      if (sr.len > 1) fw.one(position + 1, next_probed ? (uint8_t)(kQsStored | kQsSearched) : fw.unstored(position + 1));
      if (sr.len > 2) fw.copy_range(position + 2u, position + sr.len, store_end);
      if (T.own) q_store_range(Jo, P, text, position + 2u, position + sr.len < store_end ? position + sr.len : store_end);
      position += sr.len;
    } else {
      fw.one(position, (uint8_t)(kQsStored | kQsSearched));
      insert_length++;
      position++;
      if (position > apply) {
        const uint32_t margin = kQuickHtl - 1u;  // max(StoreLookahead - 1, 4)
        if (position + 16u >= pos_end - margin) {
          tail_kind = kHeadUnstored;
          tail_base = position;
          fw.unstored_range(position, pos_end);
          insert_length += pos_end - position;
          position = pos_end;
        } else if (position > apply + 4u * window) {
          tail_kind = kHeadVec4;
          tail_base = position;
          for (uint32_t q = position + BR_LANE; q < position + 16u; q += BR_NLANES) fw.put(q, ((q - position) & 3u) == 0 ? kQsStored : (uint8_t)0);
          if (T.own)
            for (uint32_t i = 0; i < 4; ++i) qs_own_store(J, T, position + i * 4u);
          insert_length += 16u;
          position += 16u;
        } else {
          tail_kind = kHeadEven4;
          tail_base = position;
          for (uint32_t q = position + BR_LANE; q < position + 8u; q += BR_NLANES) fw.put(q, ((q - position) & 1u) == 0 ? kQsStored : (uint8_t)0);
          if (T.own)
            for (uint32_t i = 0; i < 4; ++i) qs_own_store(J, T, position + i * 2u);
          insert_length += 8u;
          position += 8u;
        }
      }
    }
  }
  if (seg_flags & kSegLastInBlock) {
    if (position < pos_end) fw.unstored_range(position, pos_end);
    insert_length += pos_end - position;
    position = pos_end;
  } else if (position < seg_end) {
    // the loop ended for want of bytes in the block (its last segment is shorter than a hash): what is left is never filed
    fw.unstored_range(position, seg_end);
  }
  if (BR_LANE == 0) {
    SegExit x;
    x.pos = position;
    x.apply = apply;
    for (int i = 0; i < 4; ++i) x.cache[i] = dc[i];
    x.insert_len = insert_length;
    x.n_cmds = n_cmds;
    x.n_lits = n_lits;
    x.ext_len = ext_len;
    x.dict_lookups = ds.mode == 2 ? ds.lookups + ds.vlookups : ds.lookups;
    x.dict_matches = ds.mode == 2 ? ds.matches + ds.vwould : ds.matches;
    x.last_dist_code = last_dist_code;
    x.bad_commands = n_bad;
    x.n_searches = n_searches;
    x.last_copy_len = last_copy_len;
    x.dict_mode = ds.mode;
    x.dict_maxdef = ds.mode == 2 ? ds.vmaxdef : ds.maxdef;
    x.n_pushes = n_pushes < 4 ? n_pushes : 4u;
    x.tail_kind = position > seg_end ? tail_kind : (uint32_t)kHeadNone;
    x.tail_base = position > seg_end ? tail_base : 0u;
    x.tail_p1 = position > seg_end ? tail_p1 : 0u;
    x.n_pushes_all = n_pushes;
    x.dict_entry_lookups = ds.lookups0;
    x.dict_entry_matches = ds.matches0;
    *exit_out = x;
  }
}

}  // namespace brotli_mi355x
#endif
