// lz77_rows.h -- the candidate row of one position (see kRowEntries in lz77_types.h), shared by the gfx950 kernels
// (lz77_kernels.hip) and the serial host emulation of the device seam (tests/emu/device_emu.cpp, test infrastructure).
#ifndef BROTLI_MI355X_LZ77_ROWS_H_
#define BROTLI_MI355X_LZ77_ROWS_H_

#include "lz77_chain.h"

namespace brotli_mi355x {

// slots = positions in (key, position) order; per slot: position, flag byte (bit 0 stored), 16-bit tag.
// prev_stored(j): 1 + index of the nearest stored slot below j, 0 if there is none -- in O(1) on the device, from a
// bit mask of the stored bits per 64 slots (smask) and, per group of 64, 1 + the last stored slot in front of the group
// (gprev, a prefix maximum): the flags of whole input blocks can be 0 (a copy that extend_last_command carries through
// megabytes of zero fill is not stored anywhere), so the lookback must not step through unstored slots one by one.
struct SlotsInMemory {
  const uint32_t* by_key;
  const uint8_t* fbits;
  const uint16_t* stag;
  const unsigned long long* smask;  // null in the host emulation (linear scan)
  const uint32_t* gprev;
  BR_DEV uint32_t pos(uint32_t i) const { return by_key[i]; }
  BR_DEV uint32_t fb(uint32_t i) const { return fbits[i]; }
  BR_DEV uint32_t tag(uint32_t i) const { return stag[i]; }
  BR_DEV uint32_t prev_stored(uint32_t j) const {
    if (smask) {
      const uint32_t g = j >> 6;
      const unsigned long long m = smask[g] & ((1ull << (j & 63u)) - 1ull);
      if (m) return (g << 6) + 64u - (uint32_t)__builtin_clzll(m);
      return gprev[g];
    }
    while (j > 0 && !(fbits[j - 1] & 1u)) --j;
    return j;
  }
};

// per-slot byte (fbits): bit 0 = the position is stored in the hash table; bit 1 = bit 0 changed in the current round;
// bit 2 = the number of stored slots of the same key in front of this slot is a positive multiple of 65 536.
// The reference counts the insertions per key in a u16 (num[key], mod.rs:932-941) and looks at min(depth, num) ring
// entries (mod.rs:1752-1760): where the counter has just wrapped to 0 it sees nothing, after that only the insertions
// since the wrap.  So a slot with bit 2 has an empty row, and walking back from any other slot the candidates end
// WITH the nearest stored slot that has bit 2 (the first insertion after the wrap).
// bit 3 = the slot's ring entry is a masked position (kFlagMasked): it is never a candidate, and the lookback of every
// later slot ends when it reaches it -- for the reference it lies further back than max_backward (mod.rs:1763-1775).
static constexpr uint32_t kSlotStored = 1, kSlotChanged = 2, kSlotWrap = 4, kSlotMasked = 8;

// The row of slot i, whose key owns the slots from kf on: the (up to) `depth` nearest stored slots in front of it -- what
// the bucket ring of the reference holds when the position is searched (AdvHasher::FindLongestMatch, mod.rs:1744-1793)
// -- cut where the bucket walk breaks (backward > max_backward, :1769-1776), without the entries whose tag differs
// (FindMatchLengthWithLimitMin4 returns 0 for them, static_dict.rs:134-147); the rest of the row is kRowEnd.
template <typename Slots>
BR_DEV void br_collect_row(const Slots& sl, uint32_t max_backward_limit, uint32_t i, uint32_t kf, uint32_t depth, uint32_t* out,
                           uint32_t reset_pos = 0, uint32_t reset_vis = 0) {
  const uint32_t p = sl.pos(i), tag = sl.tag(i);
  const uint32_t oldest = p >= reset_pos ? reset_vis : 0u;  // nothing inserted before a hasher reset is seen behind it
  const uint32_t max_backward = p < max_backward_limit ? p : max_backward_limit;
  uint32_t n = 0, seen = 0;
  if (sl.fb(i) & kSlotWrap) depth = 0;
  for (uint32_t j = i; seen < depth;) {
    const uint32_t nj = sl.prev_stored(j);
    if (nj <= kf) break;  // (nj - 1 < kf: the nearest stored slot belongs to another key, or there is none)
    j = nj - 1;
    const uint32_t fb = sl.fb(j);
    const uint32_t q = sl.pos(j);
    if (p - q > max_backward || q < oldest || (fb & kSlotMasked)) break;
    ++seen;
    if (sl.tag(j) == tag) out[n++] = q;
    if (fb & kSlotWrap) break;
  }
  for (; n < kRowEntries; ++n) out[n] = kRowEnd;
}

// Writes the row of slot i to rows[].  compare: only differing words are written; returns whether the row in memory changed.
template <typename Slots>
BR_DEV bool br_build_row(const Slots& sl, uint32_t* rows, uint32_t max_backward_limit, uint32_t i, uint32_t kf, uint32_t depth,
                         bool compare, uint32_t reset_pos = 0, uint32_t reset_vis = 0) {
  uint32_t fresh[kRowEntries];
  br_collect_row(sl, max_backward_limit, i, kf, depth, fresh, reset_pos, reset_vis);
  uint32_t* row = rows + (size_t)sl.pos(i) * kRowEntries;
  bool changed = false;
  for (uint32_t n = 0; n < kRowEntries; ++n) {
    if (!compare || row[n] != fresh[n]) {
      changed = true;
      row[n] = fresh[n];
    }
  }
  return changed;
}

}  // namespace brotli_mi355x
#endif
