// lz77_live.h -- "live" chains: the reference's bucket rings kept as they are, one private copy per chain.
//
// Where the candidates of a search hang on the EXACT parse of the kilobytes in front of it, a surrogate of the hash table
// built from last round's flags never settles: AdvHasher::StoreRangeOptBatch (H5 family, backward_references/mod.rs:1163-1232)
// files the positions a copy covers, four at a time, as MASKED positions (ix & ring_buffer_mask); once the stream has
// passed one ring-buffer size FindLongestMatch takes such an entry for "further away than max_backward" and ends its
// walk through the bucket (mod.rs:1763-1775) -- the entries a search still sees are the ones at copy boundaries, i.e.
// exactly the parse-sensitive ones.  The sequential parse itself is robust (a perturbation of the input heals within
// ~100 KB, measured on the oracle for text, XML, records, hex and mixes); what does not converge is a chain that reads
// stale candidates for the positions it has just parsed ITSELF.
//
// A live chain therefore parses one whole input block with the reference's own data structure -- num[key] (u16) and
// buckets[key << block_bits | slot] (mod.rs:932-941) -- in a private copy that it updates as it goes (Store, StoreRange
// with the masked entries, Store4Vec4 / StoreEvenVec4).  The copy is MATERIALISED at the block start from the stored /
// masked flags of all earlier positions (positions sorted by (key, position) + a prefix count of the stored flags: the
// last `depth` stored positions of every key in front of the block, and the ring counter).  Flags of earlier blocks
// come from the previous round, so the scheme is still speculate + verify -- but a chain's own stores are exact.  A chain
// runs through a SPAN of several blocks (longer than the healing distance), so that what a wrong history does to the
// head of a span has died out before its tail, which is what the next span sees most of.
// Verification is separate and fully parallel: every search is logged (distance cache, what the cache and ring stages
// found); afterwards each search is repeated on its own against the ring that the flags of NOW imply for its position
// (br_live_ring_at, same index), and a block is parsed again only if one of its searches comes out differently.  When
// every search of every block checks out and the entries chain, the flags are a fixed point of the sequential parse.
// Shared by the gfx950 kernels and the host emulation (test infrastructure).
#ifndef BROTLI_MI355X_LZ77_LIVE_H_
#define BROTLI_MI355X_LZ77_LIVE_H_

#include "lz77_types.h"

namespace brotli_mi355x {

// ring entry values besides positions
static constexpr uint32_t kLiveMasked = 0xffffffffu;  // a masked position filed by this chain: ends the bucket walk
static constexpr uint32_t kLiveBreak = 0xfffffffeu;   // materialised: a masked entry, or one that lies in front of the text
                                                      // (further back than any max_backward): ends the walk as well

// The private table of one chain.
struct LiveRing {
  uint16_t* num;        // [1 << bucket_bits]
  uint32_t* buckets;    // [(1 << bucket_bits) << block_bits]
  const uint16_t* keys; // hash key of every text position
  uint32_t bits;        // block_bits: ring depth = 1 << bits
};

#if BR_SCALAR
#define BR_LIVE_LD16(p) (*(p))
#define BR_LIVE_LD32(p) (*(p))
#define BR_LIVE_ST16(p, v) (*(p) = (uint16_t)(v))
#define BR_LIVE_ST32(p, v) (*(p) = (uint32_t)(v))
#define BR_LIVE_ST8(p, v) (*(p) = (uint8_t)(v))
#else
// The chain reads back what it wrote a moment ago: device-scope accesses (served by the L2) so that no stale line of
// the vector L1 gets in between; a wavefront's accesses to one address arrive there in program order.
#define BR_LIVE_LD16(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BR_LIVE_LD32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BR_LIVE_ST16(p, v) __hip_atomic_store((p), (uint16_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BR_LIVE_ST32(p, v) __hip_atomic_store((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BR_LIVE_ST8(p, v) __hip_atomic_store((p), (uint8_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// AdvHasher::Store, mod.rs:1644-1656, for the position whose ring counter `n` the caller has read already (uniform call)
BR_DEV void br_live_insert(const LiveRing& lr, uint32_t key, uint32_t n, uint32_t q) {
  if (BR_LANE == 0) {
    BR_LIVE_ST32(lr.buckets + (((size_t)key << lr.bits) | (n & ((1u << lr.bits) - 1u))), q);
    BR_LIVE_ST16(lr.num + key, n + 1u);
  }
}

// Stores `count` positions base, base + step, ... in ascending order; those in [masked_lo, masked_hi) as masked entries.
// Store / StoreRange + StoreRangeOptBatch / Store4Vec4 / StoreEvenVec4, mod.rs:1644-1661, 1163-1232, 1581-1643, 1526-1580:
// all of them are "Store in ascending order" as far as ring counters and slots go (mod.rs test.rs:101-111); the batch
// form writes (ix & mask) where Store writes ix.
BR_DEV void br_live_store(const LiveRing& lr, uint32_t base, uint32_t step, uint32_t count, uint32_t masked_lo, uint32_t masked_hi,
                          uint32_t kwin = 0, uint32_t kwin_base = 0xffffff00u) {
  const uint32_t depth = 1u << lr.bits;
#if BR_SCALAR
  (void)kwin;
  (void)kwin_base;
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t q = base + i * step;
    const uint32_t key = lr.keys[q];
    const uint32_t n = lr.num[key];
    lr.buckets[((size_t)key << lr.bits) | (n & (depth - 1u))] = (q >= masked_lo && q < masked_hi) ? kLiveMasked : q;
    lr.num[key] = (uint16_t)(n + 1u);
  }
#else
  // (No fence between what was filed before and the loads here: a wavefront's accesses to one address reach the L2 in
  // program order, and both sides are device-scope accesses that do not stop in the vector L1.)
  // kwin / kwin_base: the chain's window of 64 hash keys, one per lane (ProbeMeta); positions inside it need no load
  for (uint32_t first = 0; first < count; first += 64) {
    const uint32_t i = first + (uint32_t)BR_LANE;
    const bool active = i < count;
    const uint32_t q = base + i * step;
    const uint32_t off = q - kwin_base;
    const uint32_t from_window = (uint32_t)__shfl((int)kwin, (int)(off & 63u), 64);
    const uint32_t key = !active ? 0xffffffffu : (off < 64u ? from_window : (uint32_t)lr.keys[q]);
    const uint32_t n = active ? (uint32_t)BR_LIVE_LD16(lr.num + key) : 0u;
    // lanes with the same key take consecutive slots in lane (= position) order
    unsigned long long todo = __ballot(active);
    uint32_t rank = 0, cnt = 1;
    while (todo != 0) {
      const uint32_t l = (uint32_t)__ffsll((long long)todo) - 1u;
      const uint32_t k = BR_READLANE(key, l);
      const unsigned long long m = __ballot(active && key == k);
      if (key == k) {
        rank = (uint32_t)__popcll(m & ((1ull << BR_LANE) - 1ull));
        cnt = (uint32_t)__popcll(m);
      }
      todo &= ~m;
    }
    if (active) {
      if (rank + depth >= cnt)  // (more than `depth` of one key in the batch: the later ones overwrite the earlier)
        BR_LIVE_ST32(lr.buckets + (((size_t)key << lr.bits) | ((n + rank) & (depth - 1u))), (q >= masked_lo && q < masked_hi) ? kLiveMasked : q);
      if (rank + 1 == cnt) BR_LIVE_ST16(lr.num + key, n + cnt);
    }
  }
#endif
}

// HasherReset at the reference's 32-bit position wrap (encode.rs:1623-1631, 1705-1710): Prepare() zeroes the ring counters,
// the bucket contents stay but are never looked at again (mod.rs:1491-1510)
BR_DEV void br_live_reset(const LiveRing& lr, uint32_t bucket_bits) {
  BR_SYNC();
  const uint32_t words = 1u << (bucket_bits - 1);  // two counters per 32-bit word
  uint32_t* w = (uint32_t*)lr.num;
  for (uint32_t i = BR_LANE; i < words; i += BR_NLANES) BR_LIVE_ST32(w + i, 0u);
  BR_SYNC();
}

// What StoreRange(first, last) of a copy files as masked entries (FlagWriter::copy_value says the same per position):
// the first 4 * floor(n / 4) positions of a range of n >= 8, from masked_from on.
BR_DEV void br_live_store_copy(const LiveRing& lr, uint32_t first, uint32_t last, uint32_t masked_from, uint32_t kwin = 0,
                               uint32_t kwin_base = 0xffffff00u) {
  if (last <= first) return;
  const uint32_t n = last - first;
  const uint32_t masked_hi = n >= 8 ? first + (n & ~3u) : first;
  // (the batch files a quad as (start & mask) + 0..3: masked from the first quad that STARTS at or behind masked_from; the quad
  // that straddles it keeps true positions)
  const uint32_t masked_lo = masked_from <= first ? first : (masked_from - first > 0xfffffff0u - first ? 0xffffffffu : first + ((masked_from - first + 3u) & ~3u));
  br_live_store(lr, first, 1, n, masked_lo, masked_hi, kwin, kwin_base);
}

// ---- materialisation --------------------------------------------------------------------------------------------------
// Slots = positions in (key, position) order.  rank[i] = number of stored slots in [0, i); entry[r] = what the r-th stored
// slot (in slot order) holds in its ring: its position, or kLiveBreak if it was filed as a masked position.
struct LiveIndex {
  const uint32_t* by_key;
  const uint32_t* rank;   // [total + 1]
  const uint32_t* entry;  // [total]
  const uint32_t* key_first;
  const uint32_t* key_last;
  const uint32_t* slot_of;     // slot of every position (inverse of by_key)
  const uint32_t* count_base;  // optional: stored positions per key in front of the text (a later piece of a stream)
  uint32_t reset_pos, reset_vis;  // Lz77Params
};

// first slot of [lo, hi) whose position is >= x
BR_DEV uint32_t br_live_lower_bound(const uint32_t* by_key, uint32_t lo, uint32_t hi, uint32_t x) {
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (by_key[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// The ring of `key` as a search at text position x finds it when every stored position < x has been filed: the ring
// counter (u16 arithmetic of the reference), how many entries a walk may look at, and where they are: entry i (newest
// first, i < visible) is ix.entry[top - 1 - i] for i < here, and lies in front of the text (kLiveBreak) otherwise.
struct LiveRingAt {
  uint32_t num, visible, here, top;
};
// idx = first slot of the key whose position is >= x
BR_DEV LiveRingAt br_live_ring_at_slot(const LiveIndex& ix, uint32_t key, uint32_t x, uint32_t idx, uint32_t depth) {
  const uint32_t kf = ix.key_first[key];
  uint32_t lo = kf, base = ix.count_base ? ix.count_base[key] : 0u;
  if (ix.reset_pos != 0 && x >= ix.reset_pos) {  // the table was emptied at reset_pos; reset_vis.. were filed again
    lo = br_live_lower_bound(ix.by_key, kf, idx, ix.reset_vis);
    base = 0;
  }
  LiveRingAt r;
  r.top = ix.rank[idx];
  r.here = r.top - ix.rank[lo];  // stored positions of this key inside the text, in front of x
  r.num = (base + r.here) & 0xffffu;
  r.visible = r.num < depth ? r.num : depth;
  return r;
}
BR_DEV LiveRingAt br_live_ring_at(const LiveIndex& ix, uint32_t key, uint32_t x, uint32_t depth) {
  return br_live_ring_at_slot(ix, key, x, br_live_lower_bound(ix.by_key, ix.key_first[key], ix.key_last[key], x), depth);
}
BR_DEV uint32_t br_live_ring_entry(const LiveIndex& ix, const LiveRingAt& r, uint32_t i) { return i < r.here ? ix.entry[r.top - 1u - i] : kLiveBreak; }

// Fills one ring of a table.
BR_DEV void br_live_materialise_key(const LiveIndex& ix, uint32_t key, uint32_t x, uint32_t bits, uint16_t* num, uint32_t* buckets) {
  const uint32_t depth = 1u << bits;
  const LiveRingAt r = br_live_ring_at(ix, key, x, depth);
  num[key] = (uint16_t)r.num;
  for (uint32_t i = 0; i < r.visible; ++i) buckets[((size_t)key << bits) | ((r.num - 1u - i) & (depth - 1u))] = br_live_ring_entry(ix, r, i);
}

}  // namespace brotli_mi355x
#endif
