// tests/emu/device_emu.cpp -- serial CPU stand-in for the device seam (device_api.h).
//
// TEST INFRASTRUCTURE ONLY.  It lets the host driver (speculative-parse resolver, meta-block
// planning) and the chain code (lz77_chain.h compiled with BROTLI_HOST_EMU, one lane per "wave")
// run in a container without a GPU.  It is never linked into the product library.
#define BROTLI_HOST_EMU 1
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "../../rust-brotli_amd/csrc/device_api.h"
#include "../../rust-brotli_amd/csrc/lz77_chain.h"
#include "../../rust-brotli_amd/csrc/lz77_rows.h"
#include "../../rust-brotli_amd/csrc/lz77_groups.h"
#include "../../rust-brotli_amd/csrc/zopfli_device.h"
#include "../../rust-brotli_amd/csrc/quick_device.h"
#include "../../rust-brotli_amd/csrc/quick_spec.h"
#include "../../rust-brotli_amd/csrc/fragment_device.h"
#include "../../tables/brotli_tables.h"
#include "../../tables/brotli_static_dict_lut.h"

namespace brotli_mi355x {

void* dev_alloc(size_t bytes) {
  void* p = calloc(bytes ? bytes : 16, 1);
  if (!p) throw std::runtime_error("emu alloc failed");
  return p;
}
void* dev_alloc_uninit(size_t bytes) {
  // poisoned, so that the CPU tests notice a read of something the device build leaves undefined
  void* p = malloc(bytes ? bytes : 16);
  if (!p) throw std::runtime_error("emu alloc failed");
  memset(p, 0xA5, bytes ? bytes : 16);
  return p;
}
void dev_free(void* p) { free(p); }
void dev_memset(void* p, int value, size_t bytes) { memset(p, value, bytes); }
void dev_h2d(void* dst, const void* src, size_t bytes) {
  if (bytes) memcpy(dst, src, bytes);  // (an empty vector hands over a null pointer)
}
void dev_d2h(void* dst, const void* src, size_t bytes) {
  if (bytes) memcpy(dst, src, bytes);
}
void dev_d2h_async(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
void dev_d2d(void* dst, const void* src, size_t bytes) { memmove(dst, src, bytes); }
void dev_sync() {}
void dev_mark() {}
void dev_wait_mark() {}
void dev_mark_n(int) {}
void dev_wait_mark_n(int) {}
void dev_make_room(unsigned) {}
void dev_make_room_for(size_t, size_t) {}
size_t dev_trim_pool() { return 0; }
void dev_h2d_bulk(void* dst, const void* src, size_t bytes) { dev_h2d(dst, src, bytes); }
void dev_d2h_bulk(void* dst, const void* src, size_t bytes) { dev_d2h(dst, src, bytes); }
int dev_current_device() { return 0; }
void dev_use_device(int) {}
int dev_device_count() { return 1; }
void* dev_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void dev_host_free(void* p) { free(p); }
void dev_pool_counters(double* out, bool) {
  for (int i = 0; i < 4; ++i) out[i] = 0;
}
const char* dev_name() { return "host-emulation"; }

const DeviceTables& dev_tables() {
  static DeviceTables t;
  t.dict_hash = kBrotliStaticDictionaryHash;
  t.dict_data = kBrotliDictionaryData;
  t.dict_offsets_by_length = kBrotliDictionaryOffsetsByLength;
  t.dict_size_bits_by_length = kBrotliDictionarySizeBitsByLength;
  t.logs_16 = (const float*)(const void*)kBrotliLog2Table16_bits;
  t.logs_8 = (const float*)(const void*)kBrotliLog2Table8_bits;
  t.utf8_context_lookup = kBrotliUTF8ContextLookup;
  t.signed_context_lookup = kBrotliSigned3BitContextLookup;
  t.dict_lut_buckets = kStaticDictionaryBuckets;
  t.dict_lut_words = kStaticDictionaryWords;
  return t;
}

size_t lz77_sort_tmp_bytes(uint32_t) { return 64; }

void lz77_compute_keys(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  const uint32_t valid_n = n >= P.htl ? n - P.htl + 1 : 0;
  const uint64_t hash_mask = P.hasher_kind == 6 ? (0xffffffffffffffffull >> (64 - 8 * P.hash_len)) : 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t key = 0xffffu;
    if (i < valid_n) {
      if (P.hasher_kind == 6) {
        key = (uint32_t)(((br_load64(B.text + i) & hash_mask) * 0x1fe35a7bd3579bd3ull) >> (64 - P.bucket_bits));
      } else {
        key = (br_load32(B.text + i) * 0x1e35a7bdu) >> (32 - P.bucket_bits);
      }
    }
    B.keys[i] = (uint16_t)key;
  }
  B.changed_count[8] = 0;  // (the emulation never asks for the run table: its match-length code has no use for it)
}

void lz77_run_table(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  if (!B.run_end) return;
  for (uint32_t p = n; p-- > 0;) B.run_end[p] = (p + 1 < n && B.text[p + 1] == B.text[p]) ? B.run_end[p + 1] : p + 1;
}

void lz77_init_flags(const Lz77Params& P, const Lz77Buffers& B, uint32_t first_block_start, const uint8_t* prefix_flags_host,
                     uint32_t prefix_flags_bytes) {
  const uint32_t M = P.total_bytes, P0 = P.prefix_bytes, htl = P.htl;
  uint8_t* f = B.flags[0];
  memset(f, 0, (size_t)M + 64);
  if (P0 > htl - 1) memset(f, 1, P0 - (htl - 1));
  if (prefix_flags_host && prefix_flags_bytes) memcpy(f, prefix_flags_host, prefix_flags_bytes);
  if (M > first_block_start) memset(f + first_block_start, 1, M - first_block_start);
  // (live chains: where masked entries exist most positions lie inside copies and are filed as such)
  if (P.masked_from != kNeverMasked && M > std::max(first_block_start, P.masked_from))
    memset(f + std::max(first_block_start, P.masked_from), kFlagStored | kFlagMasked, M - std::max(first_block_start, P.masked_from));
  for (uint32_t k = 0; k < P.num_segments; ++k) {
    const Segment& g = B.segments[k];
    if (!(g.flags & kSegFirstInBlock)) continue;
    const uint32_t bs = g.blk_start, be = g.blk_end;
    if (g.block_index == 0 && be - bs >= htl - 1 && bs >= 3) f[bs - 3] = f[bs - 2] = f[bs - 1] = 1;
    const uint32_t store_end = (be - bs >= htl) ? be - htl + 1 : bs;
    for (uint32_t q = store_end; q < be; ++q) f[q] = (q + 3 >= be && (g.flags & kSegTailStitched)) ? 1 : 0;
  }
  memcpy(B.flags[1], f, (size_t)M + 64);
}

void lz77_sort_by_key(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  std::vector<uint32_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0u);
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return B.keys[a] < B.keys[b]; });
  for (uint32_t i = 0; i < n; ++i) {
    B.by_key[i] = idx[i];
    B.sorted_keys[i] = B.keys[idx[i]];
    if (B.stag) B.stag[i] = (uint16_t)br_tag16(br_load32(B.text + idx[i]));
  }
  memset(B.key_first, 0, 65537 * 4);
  memset(B.key_last, 0, 65537 * 4);
  for (uint32_t i = 0; i < n; ++i) {
    const uint16_t k = B.sorted_keys[i];
    if (i == 0 || B.sorted_keys[i - 1] != k) B.key_first[k] = i;
    if (i + 1 == n || B.sorted_keys[i + 1] != k) B.key_last[k] = i + 1;
  }
}

// see ring_count in lz77_kernels.hip
static uint32_t emu_ring_count(uint32_t local_rank, uint32_t base) {
  const uint32_t num = (local_rank + base) & 0xffffu;
  return num < local_rank ? num : local_rank;
}

void lz77_key_counts(const Lz77Params& P, const Lz77Buffers& B, int which, uint32_t upto, uint32_t* out, bool with_base) {
  for (uint32_t key = 0; key < 65536; ++key) {
    uint32_t c = (with_base && B.count_base) ? B.count_base[key] : 0u;
    for (uint32_t i = B.key_first[key]; i < B.key_last[key]; ++i)
      if (B.by_key[i] < upto) c += B.flags[which][B.by_key[i]] & 1u;
    out[key] = c;
  }
  (void)P;
}

static uint32_t emu_ring_count_at(const Lz77Params& P, const Lz77Buffers& B, uint32_t p, uint32_t local_rank, uint32_t key) {
  if (P.reset_pos != 0 && p >= P.reset_pos) return (local_rank - B.reset_counts[key]) & 0xffffu;
  return emu_ring_count(local_rank, B.count_base ? B.count_base[key] : 0u);
}

void lz77_rank_flags(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const RankInitialHint*) {
  if (P.reset_pos) lz77_key_counts(P, B, which, P.reset_vis, B.reset_counts, false);
  const uint32_t n = P.total_bytes;
  uint32_t first = 0, local = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t pos = B.by_key[i];
    if (i == 0 || B.sorted_keys[i - 1] != B.sorted_keys[i]) {
      first = i;
      local = 0;
    }
    B.info[rbuf][2 * (size_t)pos] = first + local;
    B.info[rbuf][2 * (size_t)pos + 1] = emu_ring_count_at(P, B, pos, local, B.sorted_keys[i]);
    if (B.flags[which][pos] & 1) B.sorted[rbuf][first + local++] = (B.flags[which][pos] & kFlagMasked) ? (pos | kMaskedEntry) : pos;
  }
}

// see chain_in_front_if_near_boundary in lz77_kernels.hip
static uint32_t emu_chain_in_front(uint32_t p, const SegGeometry& geo) {
  const uint32_t blk = (p - geo.prefix_bytes) / geo.block_bytes;
  const uint32_t bs = blk == 0 ? geo.first_block_start : geo.prefix_bytes + blk * geo.block_bytes;
  const uint32_t off = p - bs;
  const uint32_t seg_bytes = geo.block_segment_bytes[blk];
  const uint32_t first = geo.block_first_segment[blk];
  const uint32_t k = first + off / seg_bytes;
  if (k >= geo.block_first_segment[blk + 1]) return 0xffffffffu;
  if (k == first || (off % seg_bytes) >= 8) return 0xffffffffu;
  return k - 1;
}

static void emu_note_rows_changed(const Lz77Buffers& B, uint32_t k, uint32_t p);
static void emu_mark_dirty(uint32_t p, const SegGeometry& geo, uint8_t* dirty, const Lz77Buffers* B = nullptr) {
  const uint32_t blk = (p - geo.prefix_bytes) / geo.block_bytes;
  const uint32_t bs = blk == 0 ? geo.first_block_start : geo.prefix_bytes + blk * geo.block_bytes;
  const uint32_t off = p - bs;
  const uint32_t seg_bytes = geo.block_segment_bytes[blk];
  uint32_t k = geo.block_first_segment[blk] + off / seg_bytes;
  if (k >= geo.block_first_segment[blk + 1]) k = geo.block_first_segment[blk + 1] - 1;
  dirty[k] = 1;
  if (B) emu_note_rows_changed(*B, k, p);
  if (k > 0 && (off % seg_bytes) < 8) {
    dirty[k - 1] = 1;
    if (B) emu_note_rows_changed(*B, k - 1, p);
  }
}

// a searched position whose candidate list changed (see list_or_mark in lz77_kernels.hip)
static void emu_list_or_mark(const Lz77Buffers& B, uint32_t p, const SegGeometry& geo, uint8_t* dirty) {
  if (B.recheck_list != nullptr) {
    const uint32_t at = (*B.recheck_count)++;
    if (at < B.recheck_cap) {
      B.recheck_list[at] = p;
      return;
    }
  }
  emu_mark_dirty(p, geo, dirty);
}

void lz77_recheck_searches(const Lz77Params& P, const Lz77Buffers& B, int rbuf, const SegGeometry& geo, uint8_t* dirty) {
  if (B.recheck_list == nullptr || B.search_log == nullptr || B.rows != nullptr) return;
  const DeviceTables& dt = dev_tables();
  ChainTables T;
  T.text = B.text;
  T.info = B.info[rbuf];
  T.sorted = B.sorted[rbuf];
  T.rows = nullptr;
  T.run_end = nullptr;
  T.work = nullptr;
  T.search_log = B.search_log;
  T.flags_next = nullptr;
  T.cmds = nullptr;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  T.dist_postfix_bits = P.dist_postfix_bits;
  T.num_direct_distance_codes = P.num_direct_distance_codes;
  ChainScratchT<false, false> scratch;
  ChainScratchT<false, false, true> scratch_deep;  // (512-deep rings: the same dispatch as lz77_kernels.hip)
  ChainScratchT<true, false> scratch9;
  const uint32_t n = *B.recheck_count < B.recheck_cap ? *B.recheck_count : B.recheck_cap;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t p = B.recheck_list[i];
    const uint32_t blk = (p - geo.prefix_bytes) / geo.block_bytes;  // (blocks are cut at prefix_bytes + k * block_bytes: see k_recheck_searches)
    const uint64_t end64 = (uint64_t)geo.prefix_bytes + (uint64_t)(blk + 1) * geo.block_bytes;
    const uint32_t blk_end = end64 < P.total_bytes ? (uint32_t)end64 : P.total_bytes;
    const bool same = P.hasher_kind == 9 ? br_recheck_search<true>(P, T, scratch9, p, blk_end)
                      : P.block_bits > 7 ? br_recheck_search<false>(P, T, scratch_deep, p, blk_end)
                                         : br_recheck_search<false>(P, T, scratch, p, blk_end);
    if (!same) emu_mark_dirty(p, geo, dirty);
  }
}

void lz77_rerank_keys(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const RerankChunk* chunks, uint32_t num_chunks,
                      uint32_t* sums, const SegGeometry& geo, uint8_t* dirty) {
  (void)sums;
  if (P.reset_pos) lz77_key_counts(P, B, which, P.reset_vis, B.reset_counts, false);
  const uint8_t* flags = B.flags[which];
  for (uint32_t e = 0; e < num_chunks; ++e) {
    if (chunks[e].first_sum != chunks[e].my_sum) continue;  // one pass per key, at its first chunk
    const uint32_t lo = chunks[e].key_lo;
    uint32_t hi = chunks[e].end;
    for (uint32_t f = e + 1; f < num_chunks && chunks[f].first_sum == chunks[e].first_sum; ++f) hi = chunks[f].end;
    std::vector<uint32_t> new_sorted, new_rank(hi - lo);
    for (uint32_t i = lo; i < hi; ++i) {
      new_rank[i - lo] = (uint32_t)new_sorted.size();
      if (flags[B.by_key[i]] & 1) new_sorted.push_back((flags[B.by_key[i]] & kFlagMasked) ? (B.by_key[i] | kMaskedEntry) : B.by_key[i]);
    }
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t p = B.by_key[i];
      if (p < geo.first_block_start) continue;
      const bool searched = (flags[p] & kFlagSearched) != 0;
      const uint32_t in_front = searched ? 0xffffffffu : emu_chain_in_front(p, geo);
      if (!searched && in_front == 0xffffffffu) continue;
      const uint32_t ax = B.info[rbuf][2 * (size_t)p], ay = B.info[rbuf][2 * (size_t)p + 1];
      const uint32_t rb = new_rank[i - lo];
      const uint32_t na = std::min(ay & 0xffffu, geo.block_size), nb = std::min(emu_ring_count_at(P, B, p, rb, B.keys[p]), geo.block_size);
      bool same = na == nb;
      for (uint32_t j = 0; same && j < na; ++j) same = B.sorted[rbuf][ax - 1 - j] == new_sorted[rb - 1 - j];
      if (!same && br_row_change_matters(B.text, p, B.sorted[rbuf] + ax - 1, na, new_sorted.data() + rb - 1, nb)) {
        if (searched) emu_list_or_mark(B, p, geo, dirty);
        else dirty[in_front] = 1;
      }
    }
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t p = B.by_key[i];
      B.info[rbuf][2 * (size_t)p] = lo + new_rank[i - lo];
      B.info[rbuf][2 * (size_t)p + 1] = emu_ring_count_at(P, B, p, new_rank[i - lo], B.keys[p]);
    }
    for (size_t j = 0; j < new_sorted.size(); ++j) B.sorted[rbuf][lo + j] = new_sorted[j];
  }
}

// ---- candidate rows: the same algorithm as the kernels of lz77_kernels.hip, one slot at a time ----
static void emu_note_rows_changed(const Lz77Buffers& B, uint32_t k, uint32_t p) {
  if (B.rows_changed_lo == nullptr) return;
  B.rows_changed_lo[k] = std::min(B.rows_changed_lo[k], p);
  B.rows_changed_hi[k] = std::max(B.rows_changed_hi[k], p);
}
static void emu_row_changed(const Lz77Buffers& B, int which, uint32_t p, const SegGeometry& geo, uint8_t* dirty) {
  if (p < geo.first_block_start) return;
  if (B.flags[which][p] & kFlagSearched) {
    emu_mark_dirty(p, geo, dirty, &B);
  } else {
    const uint32_t k = emu_chain_in_front(p, geo);
    if (k != 0xffffffffu) {
      dirty[k] = 1;
      emu_note_rows_changed(B, k, p);
    }
  }
}

static void emu_slot_masks(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes, groups = (n + 63) / 64;
  uint32_t last = 0;
  for (uint32_t g = 0; g < groups; ++g) {
    unsigned long long m = 0;
    for (uint32_t b = 0; b < 64 && g * 64 + b < n; ++b)
      if (B.fbits[g * 64 + b] & 1u) m |= 1ull << b;
    B.smask[g] = m;
    B.gprev[g] = last;
    if (m) last = g * 64 + 64 - (uint32_t)__builtin_clzll(m);
  }
}

// per-slot byte from a flag byte: stored bit, and kSlotMasked for kFlagMasked (lz77_rows.h)
static inline uint8_t emu_fb(uint8_t flag) { return (uint8_t)((flag & 1u) | ((flag & kFlagMasked) ? kSlotMasked : 0u)); }

static void emu_build_all_rows(const Lz77Params& P, const Lz77Buffers& B, int which, bool validate, const SegGeometry* geo, uint8_t* dirty) {
  const uint32_t n = P.total_bytes;
  const uint32_t depth0 = 1u << P.block_bits;
  SlotsInMemory sl{B.by_key, B.fbits, B.stag, B.smask, B.gprev};
  uint32_t kf = 0, stored_before = 0, since_reset = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (i == 0 || B.sorted_keys[i - 1] != B.sorted_keys[i]) {
      kf = i;
      stored_before = B.count_base ? B.count_base[B.sorted_keys[i]] : 0u;
      since_reset = 0;
    }
    const uint32_t key = B.sorted_keys[i];
    uint32_t depth = depth0;
    // (searches behind a hasher reset count the insertions from reset_vis on; those in front of it all of them)
    const uint32_t count = (P.reset_pos != 0 && B.by_key[i] >= P.reset_pos) ? since_reset : stored_before;
    if (P.reset_pos || B.count_base || B.key_last[key] - kf >= 65536u) depth = std::min(depth, count & 0xffffu);
    if (br_build_row(sl, B.rows, P.max_backward_limit, i, kf, depth, validate, P.reset_pos, P.reset_vis) && validate)
      emu_row_changed(B, which, B.by_key[i], *geo, dirty);
    stored_before += B.fbits[i] & 1u;
    if (P.reset_pos != 0 && B.by_key[i] >= P.reset_vis) since_reset += B.fbits[i] & 1u;
  }
}

void lz77_rows_init(const Lz77Params& P, const Lz77Buffers& B, int which, const RankInitialHint*, bool) {
  for (uint32_t i = 0; i < P.total_bytes; ++i) B.fbits[i] = emu_fb(B.flags[which][B.by_key[i]]);
  emu_slot_masks(P, B);
  emu_build_all_rows(P, B, which, false, nullptr, nullptr);
}

void lz77_rows_update(const Lz77Params& P, const Lz77Buffers& B, int prev, int next, const SegGeometry& geo, uint8_t* dirty, bool) {
  (void)prev;
  const uint32_t n = *B.changed_count, cap = B.changed_cap;
  const uint32_t depth = 1u << P.block_bits;
  // EMU_ROWS_FULL=1 forces the full rebuild (so that both paths get exercised by the CPU sweeps)
  static const bool force_full = getenv("EMU_ROWS_FULL") != nullptr;
  bool need_full = n > cap || force_full;
  if (n > cap) {
    for (uint32_t i = 0; i < P.total_bytes; ++i) B.fbits[i] = emu_fb(B.flags[next][B.by_key[i]]);
  } else {
    for (uint32_t c = 0; c < n; ++c) {
      const uint32_t p = B.changed_keys[c];
      const uint32_t key = B.keys[p];
      uint32_t lo = B.key_first[key], hi = B.key_last[key];
      if (hi - lo >= 65536u || B.count_base || P.reset_pos) need_full = true;
      while (lo + 1 < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (B.by_key[mid] <= p) lo = mid; else hi = mid;
      }
      B.fbits[lo] = (uint8_t)(emu_fb(B.flags[next][p]) | 2u);
      B.changed_slot[c] = lo;
    }
  }
  emu_slot_masks(P, B);
  if (!need_full) {
    SlotsInMemory sl{B.by_key, B.fbits, B.stag, B.smask, B.gprev};
    for (uint32_t c = 0; c < n; ++c) {
      const uint32_t s = B.changed_slot[c];
      const uint32_t key = B.keys[B.by_key[s]];
      const uint32_t kf = B.key_first[key], kl = B.key_last[key];
      uint32_t stable = 0;
      for (uint32_t i = s + 1; i < kl && stable < depth; ++i) {
        if (br_build_row(sl, B.rows, P.max_backward_limit, i, kf, depth, true, P.reset_pos, P.reset_vis)) emu_row_changed(B, next, B.by_key[i], geo, dirty);
        if ((B.fbits[i] & 3u) == 1u) ++stable;
      }
    }
  } else {
    emu_build_all_rows(P, B, next, true, &geo, dirty);
  }
  if (n <= cap)
    for (uint32_t c = 0; c < n; ++c) B.fbits[B.changed_slot[c]] &= (uint8_t)(kSlotStored | kSlotMasked);
}

// (the device launcher reads the same variable, lz77_kernels.hip)
static uint32_t emu_continuation() {
  static const uint32_t tuned = getenv("BROTLI_MI355X_CONT") ? (uint32_t)atoi(getenv("BROTLI_MI355X_CONT")) : kMaxContinuation;
  return tuned;
}

static void run_parse(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const Segment* segments, SegEntry* entries,
                      SegExit* exits, uint32_t first_segment, const uint32_t* list, uint8_t* sched, uint32_t count) {
  const DeviceTables& dt = dev_tables();
  ChainTables T;
  T.text = B.text;
  T.info = B.info[rbuf];
  T.sorted = B.sorted[rbuf];
  T.rows = B.rows;
  T.run_end = nullptr;
  T.work = nullptr;
  T.search_log = B.rows ? nullptr : B.search_log;
  T.flags_next = B.flags[which ^ 1];
  T.cmds = B.cmds;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  T.dist_postfix_bits = P.dist_postfix_bits;
  T.num_direct_distance_codes = P.num_direct_distance_codes;
  ChainScratchT<false, false> scratch;
  ChainScratchT<false, false, true> scratch_deep;
  ChainScratchT<false, true> scratch_rows;
  ChainScratchT<true, false> scratch9;
  // checkpoints: recorded by every parse of the real segments (not the warm-up's), used by the list launches
  const bool own_segments = segments == B.segments;
  T.checkpoints = own_segments ? (Checkpoint*)B.checkpoints : nullptr;
  T.rows_changed_lo = B.rows_changed_lo;
  T.rows_changed_hi = B.rows_changed_hi;
  static const uint32_t splice_part_off = getenv("BROTLI_MI355X_SPLICE_OFF") ? (uint32_t)atoi(getenv("BROTLI_MI355X_SPLICE_OFF")) : 0u;
  T.splice_off = splice_part_off;
  static const bool splice_off = getenv("BROTLI_MI355X_NO_SPLICE") != nullptr;
  const bool splice = !splice_off && sched != nullptr && own_segments && B.rows != nullptr && B.checkpoints != nullptr && B.splice_lists != 0;
  // four chains per wavefront (lz77_groups.h), an opt-in experiment on the device (BROTLI_MI355X_GROUPS_MIN=<chains>): here the same
  // switch runs its state machine with one lane per group and the sequential search (tests/test_emu_parity.py sets it to 0)
  static const uint32_t groups_min = getenv("BROTLI_MI355X_GROUPS_MIN") ? (uint32_t)atoi(getenv("BROTLI_MI355X_GROUPS_MIN")) : 0xffffffffu;
  const bool groups = B.rows && list == nullptr && sched == nullptr && count >= groups_min && P.hasher_kind != 9 && P.ndist == 4 && P.block_bits == 4 &&
                      P.spree_window == 64 && P.score_per_byte == 135 && P.dict_break == 0 && P.reset_pos == 0 && P.masked_from == kNeverMasked &&
                      (P.htl == 4 || P.htl == 8) && getenv("BROTLI_MI355X_NO_SPEC") == nullptr;
  if (groups) {
    for (uint32_t i = 0; i < count; ++i) {
      const uint32_t k = first_segment + i;
      uint32_t w, sr, cm;
      if (P.htl == 8) br_group_parse<8>(P, T, segments[k], entries[k], exits[k], true, nullptr, &w, &sr, &cm);
      else br_group_parse<4>(P, T, segments[k], entries[k], exits[k], true, nullptr, &w, &sr, &cm);
    }
    return;
  }
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t k = list ? list[i] : first_segment + i;
    if (P.hasher_kind == 9) {
      br_parse_chain<true, false>(P, T, scratch9, segments, entries, exits, k, sched, count <= 256 ? 1024u : emu_continuation());
    } else if (splice) {
      br_parse_chain<false, true, true>(P, T, scratch_rows, segments, entries, exits, k, sched, count <= 256 ? 1024u : emu_continuation());
    } else if (B.rows) {
      br_parse_chain<false, true>(P, T, scratch_rows, segments, entries, exits, k, sched, count <= 256 ? 1024u : emu_continuation());
    } else if (P.block_bits > 7) {
      br_parse_chain<false, false>(P, T, scratch_deep, segments, entries, exits, k, sched, count <= 256 ? 1024u : emu_continuation());
    } else {
      br_parse_chain<false, false>(P, T, scratch, segments, entries, exits, k, sched, count <= 256 ? 1024u : emu_continuation());
    }
  }
}

void lz77_parse_round(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, uint32_t first_segment) {
  if (first_segment >= P.num_segments) return;
  run_parse(P, B, which, rbuf, B.segments, B.entries, B.exits, first_segment, nullptr, nullptr, P.num_segments - first_segment);
}

void lz77_parse_list(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const uint32_t* list, uint8_t* sched,
                     uint32_t count) {
  run_parse(P, B, which, rbuf, B.segments, B.entries, B.exits, 0, list, sched, count);
}

void lz77_parse_custom(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const Segment* segments_dev,
                       SegEntry* entries_dev, SegExit* exits_dev, uint32_t count) {
  run_parse(P, B, which, rbuf, segments_dev, entries_dev, exits_dev, 0, nullptr, nullptr, count);
}

// ---- live chains (lz77_live.h)
static LiveIndex emu_live_index(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which) {
  LiveIndex ix;
  ix.by_key = B.by_key;
  ix.rank = L.rank[which];
  ix.entry = L.entry[which];
  ix.key_first = B.key_first;
  ix.key_last = B.key_last;
  ix.slot_of = L.slot_of;
  ix.count_base = B.count_base;
  ix.reset_pos = P.reset_pos;
  ix.reset_vis = P.reset_vis;
  return ix;
}

static ChainTables emu_chain_tables(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int flags_out) {
  const DeviceTables& dt = dev_tables();
  ChainTables T;
  T.text = B.text;
  T.info = nullptr;
  T.sorted = nullptr;
  T.rows = nullptr;
  T.run_end = nullptr;
  T.work = nullptr;
  T.search_log = B.search_log;
  T.flags_next = B.flags[flags_out];
  T.cmds = B.cmds;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  T.dist_postfix_bits = P.dist_postfix_bits;
  T.num_direct_distance_codes = P.num_direct_distance_codes;
  T.keys = B.keys;
  T.live_num = L.num;
  T.live_buckets = L.buckets;
  T.live_state = L.state;
  T.logs.logs_16 = dt.logs_16;
  T.logs.logs_8 = dt.logs_8;
  return T;
}

void lz77_live_slots(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L) {
  for (uint32_t i = 0; i < P.total_bytes; ++i) L.slot_of[B.by_key[i]] = i;
}

void lz77_live_index(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which) {
  const uint32_t n = P.total_bytes;
  uint32_t r = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t pos = B.by_key[i];
    const uint8_t f = B.flags[which][pos];
    L.rank[which][i] = r;
    if (f & kFlagStored) L.entry[which][r++] = (f & kFlagMasked) ? kLiveBreak : pos;
  }
  L.rank[which][n] = r;
}

void lz77_live_materialise(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which, const uint32_t* first, const uint32_t* start,
                           uint32_t count) {
  const LiveIndex ix = emu_live_index(P, B, L, which);
  const size_t K = (size_t)1 << P.bucket_bits;
  for (uint32_t i = 0; i < count; ++i) {
    const size_t t = first[i] / L.span_blocks;
    for (uint32_t key = 0; key < K; ++key) br_live_materialise_key(ix, key, start[i], P.block_bits, L.num + t * K, L.buckets + ((t * K) << P.block_bits));
  }
}

void lz77_live_parse(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which, const uint32_t* first, uint32_t count) {
  const ChainTables T = emu_chain_tables(P, B, L, which ^ 1);
  ChainScratchT<false, false> scratch;
  ChainScratchT<false, false, true> scratch_deep;
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t t = first[i] / L.span_blocks;
    const uint32_t last = std::min<uint32_t>((t + 1) * L.span_blocks, P.num_segments);
    uint32_t histo[256];
    if (P.block_bits > 7) {
      br_parse_live<false>(P, T, scratch_deep, B.segments, B.entries, B.exits, first[i], last, t, histo);
    } else {
      br_parse_live<false>(P, T, scratch, B.segments, B.entries, B.exits, first[i], last, t, histo);
    }
  }
}

void lz77_live_verify(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int prev, int next, const SegGeometry& geo,
                      const uint8_t* reparsed, uint8_t* dirty) {
  if (prev >= 0) {
    memset(L.changed_key, 0, 65536);
    for (uint32_t q = 0; q < P.total_bytes; ++q)
      if ((B.flags[prev][q] ^ B.flags[next][q]) & (kFlagStored | kFlagMasked)) L.changed_key[B.keys[q]] = 1;
  }
  const LiveIndex ix = emu_live_index(P, B, L, next);
  const ChainTables T = emu_chain_tables(P, B, L, next);
  ChainScratchT<false, false> scratch;
  ChainScratchT<false, false, true> scratch_deep;
  for (uint32_t p = geo.first_block_start; p < P.total_bytes; ++p) {
    if (!(B.flags[next][p] & kFlagSearched)) continue;
    const uint32_t block = (p - geo.prefix_bytes) / geo.block_bytes;
    if (dirty[block]) continue;
    if (prev >= 0 && !reparsed[block] && !L.changed_key[B.keys[p]]) continue;
    const uint32_t blk_end = std::min<uint64_t>(P.total_bytes, (uint64_t)geo.prefix_bytes + ((uint64_t)block + 1) * geo.block_bytes);
    const bool same = P.block_bits > 7 ? br_verify_search<false>(P, T, ix, scratch_deep, p, blk_end) : br_verify_search<false>(P, T, ix, scratch, p, blk_end);
    if (!same) dirty[block] = 1;
  }
}

void lz77_validate(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf_old, int rbuf_new, const SegGeometry& geo,
                   uint8_t* dirty) {
  const uint32_t n = P.total_bytes;
  const uint8_t* flags = B.flags[which];
  for (uint32_t p = geo.first_block_start; p < n; ++p) {
    const bool searched = (flags[p] & kFlagSearched) != 0;
    const uint32_t in_front = searched ? 0xffffffffu : emu_chain_in_front(p, geo);
    if (!searched && in_front == 0xffffffffu) continue;
    const uint32_t ga = B.info[rbuf_old][2 * (size_t)p], ca = B.info[rbuf_old][2 * (size_t)p + 1] & 0xffffu;
    const uint32_t gb = B.info[rbuf_new][2 * (size_t)p], cb = B.info[rbuf_new][2 * (size_t)p + 1] & 0xffffu;
    const uint32_t na = ca < geo.block_size ? ca : geo.block_size, nb = cb < geo.block_size ? cb : geo.block_size;
    bool same = na == nb;
    for (uint32_t j = 0; same && j < na; ++j) same = B.sorted[rbuf_old][ga - 1 - j] == B.sorted[rbuf_new][gb - 1 - j];
    if (same) continue;
    if (!br_row_change_matters(B.text, p, B.sorted[rbuf_old] + ga - 1, na, B.sorted[rbuf_new] + gb - 1, nb)) continue;
    if (searched) emu_list_or_mark(B, p, geo, dirty);
    else dirty[in_front] = 1;
  }
}

void lz77_parse_timing(double* total_ms, uint32_t* launches, uint64_t* segments, uint64_t* work) {
  if (work) work[0] = work[1] = work[2] = 0;
  *total_ms = 0;
  *launches = 0;
  *segments = 0;
}

void lz77_sample_histograms(const uint8_t* text, const uint32_t* ranges, uint32_t count, uint32_t* out) {
  memset(out, 0, (size_t)count * 256 * 4);
  for (uint32_t r = 0; r < count; ++r) {
    const uint32_t samples = (ranges[2 * r + 1] + 12) / 13;
    for (uint32_t i = 0; i < samples; ++i) out[(size_t)r * 256 + text[ranges[2 * r] + i * 13u]]++;
  }
}

void lz77_sample_histogram(const uint8_t* text, uint32_t start, uint32_t bytes, uint32_t* histo) {
  memset(histo, 0, 256 * 4);
  const uint32_t samples = (bytes + 12) / 13;
  for (uint32_t i = 0; i < samples; ++i) histo[text[start + i * 13u]]++;
}

void lz77_check_cache(const Lz77Params& P, const Lz77Buffers& B, int which, const CacheCheck* items, uint32_t count, uint8_t* ok) {
  for (uint32_t n = 0; n < count; ++n) {
    const Segment& seg = B.segments[items[n].segment];
    int32_t dc[16] = {0};
    for (int i = 0; i < 4; ++i) dc[i] = items[n].cache[i];
    br_prepare_distance_cache(dc, P.ndist);
    bool hit = false;
    for (uint32_t p = seg.start; p < seg.end && !hit; ++p) {
      if (!(B.flags[which][p] & kFlagSearched)) continue;
      const uint32_t max_backward = p < P.max_backward_limit ? p : P.max_backward_limit;
      for (uint32_t i = 0; i < P.ndist; ++i) {
        const int64_t d = (int64_t)dc[i];
        if (d <= 0 || d > (int64_t)max_backward) continue;
        const uint32_t q = p - (uint32_t)d;
        hit |= B.text[p] == B.text[q] && B.text[p + 1] == B.text[q + 1];
      }
    }
    ok[n] = hit ? 0 : 1;
  }
}

void lz77_gather_commands(const Lz77Params& P, const Lz77Buffers& B, uint32_t num_segments, const uint32_t* offsets,
                          const uint32_t* counts, Command* out) {
  for (uint32_t k = 0; k < num_segments; ++k)
    for (uint32_t i = 0; i < counts[k]; ++i)
      out[offsets[k] + i] = br_finish_command(B.cmds[(size_t)B.segments[k].cmd_base + i], P.num_direct_distance_codes, P.dist_postfix_bits);
}

// ---- bursts (device_api.h)
void lz77_chain_check(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U) {
  for (uint32_t k = 0; k < P.num_segments; ++k)
    br_chain_check(B.segments, B.entries, B.exits, P.num_segments, k, U.sched, U.touched, U.entry_dirty, U.new_entries, B.rows_changed_lo, B.rows_changed_hi, U.stale);
}
void lz77_flags_catch_up(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, int src, int dst) {
  for (uint32_t k = 0; k < P.num_segments; ++k) {
    if (!U.stale[k]) continue;
    for (uint32_t q = B.segments[k].start; q < B.segments[k].end; ++q) B.flags[dst][q] = B.flags[src][q];
    U.stale[k] = 0;
  }
}
void lz77_diff_flags_touched(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, int prev, int next) {
  uint32_t count = 0;
  for (uint32_t k = 0; k < P.num_segments; ++k) {
    if (!U.stale[k]) continue;
    for (uint32_t q = B.segments[k].start; q < B.segments[k].end; ++q) {
      if ((B.flags[prev][q] ^ B.flags[next][q]) & (kFlagStored | kFlagMasked)) {
        if (count < B.changed_cap) B.changed_keys[count] = q;
        count++;
      }
    }
  }
  *B.changed_count = count;
}
void lz77_reset_rows_changed(const Lz77Params& P, const Lz77Buffers& B) {
  if (B.rows_changed_lo == nullptr) return;
  for (uint32_t k = 0; k < P.num_segments; ++k) {
    B.rows_changed_lo[k] = 0xffffffffu;
    B.rows_changed_hi[k] = 0;
  }
}
void lz77_drop_checkpoints(const Lz77Params& P, const Lz77Buffers& B, const uint32_t* list, uint32_t count) {
  if (B.checkpoints == nullptr) return;
  for (uint32_t i = 0; i < count; ++i) br_drop_checkpoints((Checkpoint*)B.checkpoints, B.segments[list[i]]);
}
void lz77_burst_count(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U) {
  uint32_t n = 0;
  for (uint32_t k = 0; k < P.num_segments; ++k) n += (U.cand_dirty[k] != 0 || U.entry_dirty[k] != 0 || U.sched[k] == 2) ? 1u : 0u;
  U.counters[0] = n;
}
void lz77_burst_schedule(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U) {
  uint32_t n = 0;
  for (uint32_t k = 0; k < P.num_segments; ++k)
    if (br_burst_schedule_one(B.entries, k, U.sched, U.cand_dirty, U.entry_dirty, U.new_entries)) U.list[n++] = k;
  U.counters[0] = n;
}
void lz77_gather_touched(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, uint32_t* index_out, SegExit* exits_out, SegEntry* entries_out) {
  uint32_t n = 0;
  for (uint32_t k = P.num_segments; k-- > 0;) {  // (any order will do: the host must not rely on one)
    if (!U.touched[k]) continue;
    U.touched[k] = 0;
    index_out[n] = k;
    exits_out[n] = B.exits[k];
    entries_out[n] = B.entries[k];
    ++n;
  }
  U.counters[1] = n;
}

void lz77_scatter_entries(const Lz77Buffers& B, const uint32_t* index, const SegEntry* entries, uint32_t count) {
  for (uint32_t i = 0; i < count; ++i) B.entries[index[i]] = entries[i];
}

void lz77_gather_results(const Lz77Buffers& B, const uint32_t* list, uint32_t count, const uint8_t* sched, uint32_t num_segments,
                         SegExit* exits_out, uint32_t* cont_count, uint32_t* cont_index, SegExit* cont_exits, SegEntry* cont_entries) {
  for (uint32_t i = 0; i < count; ++i) exits_out[i] = B.exits[list[i]];
  uint32_t n = 0;
  for (uint32_t k = num_segments; k-- > 0;) {  // (any order will do: the host must not rely on one)
    if (sched[k] != 3) continue;
    cont_index[n] = k;
    cont_exits[n] = B.exits[k];
    cont_entries[n] = B.entries[k];
    ++n;
  }
  *cont_count = n;
}

void lz77_diff_flags(const Lz77Params& P, const Lz77Buffers& B, int prev, int next) {
  uint32_t count = 0;
  for (uint32_t q = 0; q < P.total_bytes; ++q) {
    if ((B.flags[prev][q] ^ B.flags[next][q]) & (kFlagStored | kFlagMasked)) {
      if (count < B.changed_cap) B.changed_keys[count] = B.rows ? q : (uint32_t)B.keys[q];
      count++;
    }
  }
  *B.changed_count = count;
}

void lz77_patch_commands(Command* cmds, const CmdPatch* patches, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) br_apply_patch(cmds, patches[i]);
}


// ---- qualities 10 / 11 (zopfli_device.h): the same item code, called directly
static ZopfliParams emu_zopfli_params(const Lz77Params& P, const ZopfliJob& J) {
  ZopfliParams Z;
  Z.quality = J.quality;
  Z.lgwin = J.lgwin;
  Z.max_backward_limit = P.max_backward_limit;
  Z.ring_mask = P.ring_mask;
  Z.dict_break = P.dict_break;
  Z.use_dictionary = J.use_dictionary;
  Z.dist_max_distance = P.dist_max_distance;
  Z.dist_alphabet_size = J.dist_alphabet_size;
  Z.ndirect = P.num_direct_distance_codes;
  Z.npostfix = P.dist_postfix_bits;
  return Z;
}
static ZopfliBuffers emu_zopfli_buffers(const ZopfliJob& J) {
  ZopfliBuffers Z;
  Z.buckets = J.buckets;
  Z.forest = J.forest;
  Z.nodes = (ZNode*)J.nodes;
  Z.literal_costs = J.literal_costs;
  Z.cost_dist = J.cost_dist;
  Z.cost_cmd = J.cost_cmd;
  Z.matches = J.matches;
  Z.num_matches = J.num_matches;
  Z.tmp_cmds = J.tmp_cmds;
  Z.histo = J.histo;
  return Z;
}
void lz77_zopfli_init(const ZopfliJob& J) {
  const uint32_t window_mask = (1u << J.lgwin) - 1u;
  for (size_t i = 0; i < ((size_t)1 << kZBucketBits); ++i) J.buckets[i] = 0u - window_mask;
  memset(J.forest, 0, ((size_t)2 << J.lgwin) * 4);
}
void lz77_zopfli_import(const ZopfliJob& J, const uint32_t* buckets_src, const uint32_t* forest_src, uint32_t delta) {
  const uint32_t invalid = 0u - ((1u << J.lgwin) - 1u);
  for (size_t i = 0; i < ((size_t)1 << kZBucketBits); ++i) J.buckets[i] = (buckets_src[i] != invalid && buckets_src[i] >= delta) ? buckets_src[i] - delta : invalid;
  for (size_t i = 0; i < ((size_t)2 << J.lgwin); ++i) J.forest[i] = (forest_src[i] != invalid && forest_src[i] >= delta) ? forest_src[i] - delta : invalid;
}
void lz77_zopfli_prepend(const Lz77Params& P, const Lz77Buffers& B, const ZopfliJob& J, uint32_t dict_bytes) {
  const ZopfliParams Z = emu_zopfli_params(P, J);
  const ZH10 h = z_hasher_of(Z, emu_zopfli_buffers(J));
  for (uint32_t i = 0; i + (kZMaxTreeCompLength - 1) < dict_bytes; ++i) z_h10_store(h, Z, B.text, i);
}
bool lz77_zopfli_block(const Lz77Params& P, const Lz77Buffers& B, const ZopfliJob& J, uint32_t block) {
  static const bool sequential_only = getenv("BROTLI_MI355X_ZOPFLI_SEQUENTIAL") != nullptr;
  const DeviceTables& dt = dev_tables();
  ZopfliTables T;
  T.lut_buckets = dt.dict_lut_buckets;
  T.lut_words = dt.dict_lut_words;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  T.logs.logs_16 = dt.logs_16;
  T.logs.logs_8 = dt.logs_8;
  const ZopfliParams Z = emu_zopfli_params(P, J);
  const ZopfliBuffers ZB = emu_zopfli_buffers(J);
  const Segment seg = B.segments[block];
  ZBlockCtl* ctl = (ZBlockCtl*)J.ctl;
  const size_t forest_bytes = ((size_t)2 << J.lgwin) * 4, bucket_bytes = ((size_t)1 << kZBucketBits) * 4;
  br_zopfli_begin(Z, ZB, B.text, seg, B.entries[block], ctl);
  bool redo = sequential_only;
  if (!sequential_only) {
    memcpy(J.forest_bak, J.forest, forest_bytes);
    memcpy(J.buckets_bak, J.buckets, bucket_bytes);
    // (the groups in an order of their own: what one of them leaves in the node arrays must not matter to another)
    for (uint32_t n = 0; n < 65536; ++n) {
      const uint32_t g = (n * 40503u + 12345u) & 0xffffu;
      if (B.key_first[g] < B.key_last[g]) br_zopfli_matches_of_group(Z, T, ZB, J.forest_new, J.rerooted, B.text, B.by_key, B.key_first[g], B.key_last[g], ctl);
    }
    for (uint32_t i = 0; i < J.block_bytes; ++i) br_zopfli_merge_node(Z, ZB, J.forest_new, J.rerooted, ctl, i);
    redo = br_zopfli_parse(Z, T, ZB, B.text, seg, B.entries[block], ctl, true, B.cmds + seg.cmd_base, B.exits + block) == kZopfliRedo;
    if (redo) {
      memcpy(J.forest, J.forest_bak, forest_bytes);
      memcpy(J.buckets, J.buckets_bak, bucket_bytes);
    }
  }
  if (redo) br_zopfli_parse(Z, T, ZB, B.text, seg, B.entries[block], ctl, false, B.cmds + seg.cmd_base, B.exits + block);
  return redo;
}

// ---- qualities 2 .. 4 (quick_device.h): the same item code, called directly
void lz77_quick_init(const QuickJob& J) { memset(J.table, 0, (size_t)quick_table_words(J) * 4); }
void lz77_quick_import(const QuickJob& J, const uint32_t* table_src, uint32_t delta) {
  const uint32_t slots = quick_slots(J), words = quick_table_words(J);
  for (uint32_t i = 0; i < words; ++i) J.table[i] = i < slots ? (table_src[i] >= delta ? table_src[i] - delta : 0u) : table_src[i];
}
void lz77_quick_prepend(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, uint32_t dict_bytes) { br_quick_prepend(J, B.text, dict_bytes); }
void lz77_quick_block(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, uint32_t block) {
  const DeviceTables& dt = dev_tables();
  QuickTables T;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  const Segment seg = B.segments[block];
  br_quick_block(J, P, T, B.text, seg, B.entries[block], B.cmds + seg.cmd_base, B.exits + block);
}

// ---- qualities 2 .. 4 on the speculative path (quick_spec.h): the same item code, one item after the other
size_t lz77_qspec_sort_tmp_bytes(uint32_t) { return 64; }
void lz77_qspec_index(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S) {
  std::vector<uint32_t> slot(S.events), order(S.events);
  for (uint32_t id = 0; id < S.events; ++id) slot[id] = qs_event_slot(J, B.text, id);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return slot[a] < slot[b]; });
  for (uint32_t i = 0; i < S.events; ++i) {
    S.ev_slot[i] = slot[order[i]];
    S.ev_id[i] = order[i];
    S.ev_of[order[i]] = i;
  }
  uint32_t i = 0;
  for (uint32_t t = 0; t <= S.slots + 1u; ++t) {
    while (i < S.events && S.ev_slot[i] < t) ++i;
    S.slot_first[t] = i;
  }
  for (uint32_t p = 0; p < S.n; ++p)
    for (uint32_t j = 0; j < J.sweep; ++j) {
      const uint32_t s = qs_hash(J, B.text + p) + j;
      S.qrank[(size_t)p * J.sweep + j] = qs_rank_in_slot(J, S.ev_id, S.slot_first[s], S.slot_first[s + 1], p);
    }
}
void lz77_qspec_init_flags(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t first_block_start, bool prefix_is_dictionary) {
  for (uint32_t q = 0; q < P.total_bytes; ++q) S.flags[q] = qs_initial_flag(P, q, first_block_start, prefix_is_dictionary);
}
void lz77_qspec_candidates(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const SegGeometry* geo, uint8_t* dirty) {
  for (uint32_t q = 0; q < S.n; ++q) qs_item_activate(J, P, S, q);
  uint32_t running = 0;
  for (uint32_t i = 0; i < S.events; ++i) {  // inclusive max-scan
    running = std::max(running, S.actraw[i]);
    S.act[i] = running;
  }
  for (uint32_t p = 0; p < S.n; ++p)
    for (uint32_t j = 0; j < J.sweep; ++j) {
      const size_t item = (size_t)p * J.sweep + j;
      const uint32_t c = qs_candidate(J, S, qs_hash(J, B.text + p) + j, S.qrank[item]);
      if (geo != nullptr) {
        const uint32_t was = S.cand[item];
        if (c == was) continue;
        if (qs_change_matters(J, P, B.text, p, was, c)) qs_note_changed(S, p, *geo, dirty);
      }
      S.cand[item] = c;
    }
}
void lz77_qspec_diff(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list, uint32_t count) {
  S.chg_count[0] = S.chg_count[1] = 0;
  auto mark = [&](uint32_t e) {
    const uint32_t at = S.chg_count[0]++;
    if (at < S.chg_cap) S.chg_list[at] = e;
  };
  for (uint32_t i = 0; i < count; ++i) {
    const Segment& seg = B.segments[list[i]];
    for (uint32_t q = seg.start; q < seg.end; ++q) qs_item_diff(J, P, S, q, mark);
  }
}
void lz77_qspec_repair(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t changed, const SegGeometry& geo, uint8_t* dirty) {
  for (uint32_t n = 0; n < changed; ++n)
    if (!qs_item_repair(J, S, n)) {
      S.chg_count[1] = 1;
      S.chg_range[3u * n] = S.ev_slot[S.chg_list[n]];
      S.chg_range[3u * n + 1u] = 0xffffffffu;
      S.chg_range[3u * n + 2u] = 0u;
    }
  const uint32_t around = 2u * J.sweep - 1u;
  for (uint32_t n = 0; n < changed; ++n)
    for (uint32_t d = 0; d < around; ++d) {
      const uint32_t slot = S.chg_range[3u * n];
      if (slot + d < J.sweep - 1u) continue;
      const uint32_t t = slot + d - (J.sweep - 1u);
      if (t >= S.slots) continue;
      if (!qs_item_recand(J, S, n, t, [&](uint32_t p, uint32_t was, uint32_t now) {
            if (qs_change_matters(J, P, B.text, p, was, now)) qs_note_changed(S, p, geo, dirty);
          }))
        S.chg_count[1] = 1;
    }
}
void lz77_qspec_parse(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list, uint32_t count, uint32_t* own_tables,
                      uint32_t own_stride) {
  const DeviceTables& dt = dev_tables();
  QsTables T;
  T.text = B.text;
  T.cand = S.cand;
  T.flags = S.flags;
  T.own_stride = own_stride;
  T.dict.dict_hash = dt.dict_hash;
  T.dict.dict_data = dt.dict_data;
  T.dict.dict_offsets_by_length = dt.dict_offsets_by_length;
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t k = list ? list[i] : i;
    const Segment seg = B.segments[k];
    T.own = own_tables ? own_tables + (size_t)i * own_stride : nullptr;
    br_quick_segment(J, P, T, seg, B.entries[k], B.cmds + seg.cmd_base, B.exits + k);
  }
}
void lz77_qspec_block_tables(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list, uint32_t count, uint32_t* tables,
                             uint32_t stride) {
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t upto = B.segments[list ? list[i] : i].start;
    for (uint32_t s = 0; s < S.slots; ++s)
      tables[(size_t)i * stride + s] = qs_candidate(J, S, s, qs_rank_in_slot_guess(J, S.ev_id, S.slot_first[s], S.slot_first[s + 1], upto, S.n));
  }
}
void lz77_qspec_first_change(const Lz77Buffers& B, const QuickSpec& S, const uint32_t* list, uint32_t count, uint32_t* out) {
  const uint8_t filing_bits = (uint8_t)(kQsStored | kQsQuad | 0x18u);
  uint32_t first = 0xffffffffu;
  for (uint32_t i = 0; i < count; ++i) {
    const Segment seg = B.segments[list ? list[i] : i];
    for (uint32_t q = seg.start; q < seg.end && q < first; ++q)
      if ((S.flags[q] ^ S.flags_prev[q]) & filing_bits) first = q;
  }
  *out = first;
}
void lz77_qspec_parse_custom(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const Segment* segments, const SegEntry* entries,
                             SegExit* exits, uint32_t count) {
  const DeviceTables& dt = dev_tables();
  QsTables T;
  T.text = B.text;
  T.cand = S.cand;
  T.flags = S.flags;
  T.dict.dict_hash = dt.dict_hash;
  T.dict.dict_data = dt.dict_data;
  T.dict.dict_offsets_by_length = dt.dict_offsets_by_length;
  for (uint32_t k = 0; k < count; ++k) br_quick_segment(J, P, T, segments[k], entries[k], B.cmds + segments[k].cmd_base, exits + k);
}
void lz77_qspec_gather_exits(const Lz77Buffers& B, const uint32_t* list, uint32_t count, SegExit* out) {
  for (uint32_t i = 0; i < count; ++i) out[i] = B.exits[list[i]];
}
void lz77_qspec_table(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t upto, uint32_t* out) {
  for (uint32_t s = 0; s < S.slots; ++s) {
    const uint32_t lo = S.slot_first[s], hi = S.slot_first[s + 1];
    out[s] = qs_candidate(J, S, s, upto == 0xffffffffu ? hi : qs_rank_in_slot(J, S.ev_id, lo, hi, upto));
  }
}

// ---- qualities 0 and 1 (fragment_device.h): the same item code, called directly, one fragment after the other
void frag_compress_batch(int quality, const uint8_t* input, const FragmentJob* jobs, uint32_t n, const FragmentBuffers& B, const FragmentState* states_in,
                         FragmentState* states_out, FragmentResult* results, uint8_t* out) {
  const DeviceTables& dt = dev_tables();
  EntropyTables et;
  et.logs_16 = dt.logs_16;
  et.logs_8 = dt.logs_8;
  static thread_local FragmentScratch S;
  static thread_local uint64_t cmd_code_words[kTreeBitsWords];
  for (uint32_t j = 0; j < n; ++j) {
    memset(B.table + (size_t)j * B.table_stride, 0, ((size_t)1 << jobs[j].table_bits) * 4);
    br_fragment_job(quality, et, input, jobs[j], j, B, states_in, states_out, results, out, S, cmd_code_words);
  }
}
void frag_join(const uint8_t* src, const FragmentPiece* pieces, uint32_t n, uint8_t* dst) {
  for (uint32_t i = 0; i < n; ++i)
    for (uint64_t b = 0; b < pieces[i].nbits; ++b) {
      const uint64_t s = pieces[i].src_bit + b, d = pieces[i].dst_bit + b;
      if ((src[s >> 3] >> (s & 7)) & 1) dst[d >> 3] |= (uint8_t)(1u << (d & 7));
    }
}

}  // namespace brotli_mi355x
