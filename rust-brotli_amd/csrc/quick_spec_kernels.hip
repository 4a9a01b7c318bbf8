// quick_spec_kernels.hip -- gfx950 kernels of the speculative path of qualities 2..4 (quick_api.h, quick_spec.h): the potential
// filings of every position sorted by (slot, position) once per text; per round the active filings marked from the flags, an
// exclusive max-scan over them, the `sweep` candidates of every position gathered and compared with the round before; the chains
// (one wavefront per segment) parse with those candidates.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "quick_spec.h"
#include "device_scan.h"

namespace brotli_mi355x {

static QuickTables qspec_dict_tables() {
  const DeviceTables& dt = dev_tables();
  QuickTables T;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  return T;
}
static inline uint32_t qs_blocks(uint64_t items, uint32_t per_block = 256) { return (uint32_t)((items + per_block - 1) / per_block); }

// ---- index (once per text) ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_qs_events(QuickJob J, const uint8_t* __restrict__ text, uint32_t events, uint32_t* __restrict__ slot_out,
                                                   uint32_t* __restrict__ id_out) {
  const uint32_t id = blockIdx.x * 256u + threadIdx.x;
  if (id >= events) return;
  slot_out[id] = qs_event_slot(J, text, id);
  id_out[id] = id;
}
// slot_first[t] = first event with slot >= t; ev_of = inverse of ev_id
__global__ __launch_bounds__(256) void k_qs_slot_ranges(const uint32_t* __restrict__ ev_slot, const uint32_t* __restrict__ ev_id, uint32_t events, uint32_t slots,
                                                        uint32_t* __restrict__ slot_first, uint32_t* __restrict__ ev_of) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i > events) return;
  const uint32_t lo = i == 0 ? 0u : ev_slot[i - 1] + 1u;
  const uint32_t hi = i == events ? slots + 1u : ev_slot[i];
  for (uint32_t t = lo; t <= hi && t <= slots + 1u; ++t) slot_first[t] = i;
  if (i < events) ev_of[ev_id[i]] = i;
}
__global__ __launch_bounds__(256) void k_qs_qrank(QuickJob J, QuickSpec S, const uint8_t* __restrict__ text) {
  const uint64_t item = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (item >= (uint64_t)S.n * J.sweep) return;
  const uint32_t p = (uint32_t)(item / J.sweep), j = (uint32_t)(item % J.sweep);
  const uint32_t slot = qs_hash(J, text + p) + j;
  // (in the slot of its own offset a position's rank is its own event)
  if (j == ((p >> 3) & (J.sweep - 1u))) S.qrank[item] = S.ev_of[J.sweep == 1 ? p : 2u * p];
  else S.qrank[item] = qs_rank_in_slot_guess(J, S.ev_id, S.slot_first[slot], S.slot_first[slot + 1], p, S.n);
}

size_t lz77_qspec_sort_tmp_bytes(uint32_t events) {
  size_t bytes = 0;
  HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)events, 0u, 32u,
                                      BR_STREAM));
  return bytes + 256;
}

void lz77_qspec_index(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S) {
  hipLaunchKernelGGL(k_qs_events, dim3(qs_blocks(S.events)), dim3(256), 0, BR_STREAM, J, (const uint8_t*)B.text, S.events, S.sort_keys_tmp, S.sort_ids_tmp);
  HIP_CHECK(hipGetLastError());
  // stable: the ids ascend by position, so the events of a slot come out in position order
  uint32_t bits = 1;
  while (bits < 32 && (1ull << bits) <= (unsigned long long)S.slots + 1ull) ++bits;
  size_t bytes = S.sort_tmp_bytes;
  HIP_CHECK(rocprim::radix_sort_pairs(S.sort_tmp, bytes, S.sort_keys_tmp, S.ev_slot, S.sort_ids_tmp, S.ev_id, (size_t)S.events, 0u, bits, BR_STREAM));
  hipLaunchKernelGGL(k_qs_slot_ranges, dim3(qs_blocks((uint64_t)S.events + 1)), dim3(256), 0, BR_STREAM, (const uint32_t*)S.ev_slot, (const uint32_t*)S.ev_id, S.events,
                     S.slots, S.slot_first, S.ev_of);
  hipLaunchKernelGGL(k_qs_qrank, dim3(qs_blocks((uint64_t)S.n * J.sweep)), dim3(256), 0, BR_STREAM, J, S, (const uint8_t*)B.text);
  HIP_CHECK(hipGetLastError());
}

// ---- flags --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_qs_init_flags(Lz77Params P, uint8_t* __restrict__ flags, uint32_t first_block_start, bool prefix_is_dictionary) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= P.total_bytes) return;
  flags[q] = qs_initial_flag(P, q, first_block_start, prefix_is_dictionary);
}
void lz77_qspec_init_flags(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t first_block_start, bool prefix_is_dictionary) {
  hipLaunchKernelGGL(k_qs_init_flags, dim3(qs_blocks(P.total_bytes)), dim3(256), 0, BR_STREAM, P, S.flags, first_block_start, prefix_is_dictionary);
  HIP_CHECK(hipGetLastError());
}

// ---- candidates: the pass over everything --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_qs_activate(QuickJob J, Lz77Params P, QuickSpec S) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q < S.n) qs_item_activate(J, P, S, q);
}

// max-scan of a uint32 array (the pattern of device_scan.h with max for +): dst[i] = max(src[0 .. i]) (inclusive) or max(src[0 .. i - 1])
static constexpr uint32_t kQsScanTile = 1024;
__global__ __launch_bounds__(256) void k_qs_maxscan_tiles(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n, uint32_t* __restrict__ tile_max,
                                                          bool inclusive) {
  __shared__ uint32_t wave_max[4];
  const uint32_t base = blockIdx.x * kQsScanTile + threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (base + j < n) ? src[base + j] : 0u;
  const uint32_t local = max(max(v[0], v[1]), max(v[2], v[3]));
  uint32_t x = local;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x = max(x, y);
  }
  if (lane == 63) wave_max[w] = x;
  __syncthreads();
  uint32_t before = 0;
  for (int i = 0; i < w; ++i) before = max(before, wave_max[i]);
  const uint32_t up = __shfl_up(x, 1, 64);
  uint32_t run = max(before, lane == 0 ? 0u : up);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t incl = max(run, v[j]);
    if (base + j < n) dst[base + j] = inclusive ? incl : run;
    run = incl;
  }
  if (threadIdx.x == 255 && tile_max) tile_max[blockIdx.x] = max(before, x);
}
__global__ __launch_bounds__(256) void k_qs_maxscan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ tile_before) {
  const uint32_t base = blockIdx.x * kQsScanTile + threadIdx.x * 4;
  const uint32_t add = tile_before[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) data[base + j] = max(data[base + j], add);
}
static void qs_maxscan(const uint32_t* src, uint32_t* dst, uint32_t n, uint32_t* scratch, bool inclusive) {
  if (n == 0) return;
  const uint32_t tiles = (n + kQsScanTile - 1) / kQsScanTile;
  hipLaunchKernelGGL(k_qs_maxscan_tiles, dim3(tiles), dim3(256), 0, BR_STREAM, src, dst, n, tiles > 1 ? scratch : (uint32_t*)nullptr, inclusive);
  if (tiles > 1) {
    qs_maxscan(scratch, scratch, tiles, scratch + tiles, false);
    hipLaunchKernelGGL(k_qs_maxscan_add, dim3(tiles), dim3(256), 0, BR_STREAM, dst, n, (const uint32_t*)scratch);
  }
}

__global__ __launch_bounds__(256) void k_qs_candidates(QuickJob J, Lz77Params P, QuickSpec S, const uint8_t* __restrict__ text, bool compare, SegGeometry geo,
                                                       uint8_t* __restrict__ dirty) {
  const uint64_t item = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (item >= (uint64_t)S.n * J.sweep) return;
  const uint32_t p = (uint32_t)(item / J.sweep), j = (uint32_t)(item % J.sweep);
  const uint32_t c = qs_candidate(J, S, qs_hash(J, text + p) + j, S.qrank[item]);
  if (compare) {
    const uint32_t was = S.cand[item];
    if (c == was) return;
    if (qs_change_matters(J, P, text, p, was, c)) qs_note_changed(S, p, geo, dirty);
  }
  S.cand[item] = c;
}

void lz77_qspec_candidates(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const SegGeometry* geo, uint8_t* dirty_dev) {
  hipLaunchKernelGGL(k_qs_activate, dim3(qs_blocks(S.n)), dim3(256), 0, BR_STREAM, J, P, S);
  qs_maxscan(S.actraw, S.act, S.events, S.scan_tmp, true);
  SegGeometry g{};
  if (geo) g = *geo;
  hipLaunchKernelGGL(k_qs_candidates, dim3(qs_blocks((uint64_t)S.n * J.sweep)), dim3(256), 0, BR_STREAM, J, P, S, (const uint8_t*)B.text, geo != nullptr, g, dirty_dev);
  HIP_CHECK(hipGetLastError());
}

// ---- candidates: in proportion to the changes ---------------------------------------------------------------------------------------
// one wavefront per segment that was parsed again: its flags against the ones the candidates stand for
__global__ __launch_bounds__(64) void k_qs_diff(QuickJob J, Lz77Params P, QuickSpec S, const Segment* __restrict__ segments, const uint32_t* __restrict__ list,
                                                uint32_t count) {
  if (blockIdx.x >= count) return;
  const Segment seg = segments[list[blockIdx.x]];
  auto mark = [&](uint32_t e) {
    const uint32_t at = atomicAdd(&S.chg_count[0], 1u);
    if (at < S.chg_cap) S.chg_list[at] = e;
  };
  // eight flags per lane and load; almost all of them are what they were
  const uint32_t a0 = (seg.start + 7u) & ~7u, a1 = seg.end & ~7u;
  if (a0 >= a1) {
    for (uint32_t q = seg.start + threadIdx.x; q < seg.end; q += 64u) qs_item_diff(J, P, S, q, mark);
    return;
  }
  for (uint32_t q = seg.start + threadIdx.x; q < a0; q += 64u) qs_item_diff(J, P, S, q, mark);
  for (uint32_t q = a1 + threadIdx.x; q < seg.end; q += 64u) qs_item_diff(J, P, S, q, mark);
  for (uint32_t q8 = a0 + 8u * threadIdx.x; q8 < a1; q8 += 512u) {
    const unsigned long long f = *(const unsigned long long*)(S.flags + q8), g = *(const unsigned long long*)(S.flags_prev + q8);
    if (f == g) continue;
    for (uint32_t b = 0; b < 8u; ++b)
      if (((f ^ g) >> (8u * b)) & 0xffull) qs_item_diff(J, P, S, q8 + b, mark);
  }
}
void lz77_qspec_diff(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list_dev, uint32_t count) {
  dev_memset(S.chg_count, 0, 64);
  if (count == 0) return;
  hipLaunchKernelGGL(k_qs_diff, dim3(count), dim3(64), 0, BR_STREAM, J, P, S, (const Segment*)B.segments, list_dev, count);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_qs_repair(QuickJob J, QuickSpec S, uint32_t changed) {
  const uint32_t n = blockIdx.x * 256u + threadIdx.x;
  if (n >= changed) return;
  if (!qs_item_repair(J, S, n)) {
    S.chg_count[1] = 1u;
    S.chg_range[3u * n + 1u] = 0xffffffffu;  // (nothing to derive again: the caller takes the pass over everything)
    S.chg_range[3u * n + 2u] = 0u;
    S.chg_range[3u * n] = S.ev_slot[S.chg_list[n]];
  }
}
__global__ __launch_bounds__(256) void k_qs_recand(QuickJob J, Lz77Params P, QuickSpec S, const uint8_t* __restrict__ text, uint32_t changed, SegGeometry geo,
                                                   uint8_t* __restrict__ dirty) {
  const uint32_t around = 2u * J.sweep - 1u;
  const uint32_t item = blockIdx.x * 256u + threadIdx.x;
  if (item >= changed * around) return;
  const uint32_t n = item / around, d = item % around;
  const uint32_t slot = S.chg_range[3u * n];
  if (slot + d < J.sweep - 1u) return;
  const uint32_t t = slot + d - (J.sweep - 1u);
  if (t >= S.slots) return;
  if (!qs_item_recand(J, S, n, t, [&](uint32_t p, uint32_t was, uint32_t now) {
        if (qs_change_matters(J, P, text, p, was, now)) qs_note_changed(S, p, geo, dirty);
      }))
    S.chg_count[1] = 1u;
}
void lz77_qspec_repair(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t changed, const SegGeometry& geo, uint8_t* dirty_dev) {
  if (changed == 0) return;
  hipLaunchKernelGGL(k_qs_repair, dim3(qs_blocks(changed)), dim3(256), 0, BR_STREAM, J, S, changed);
  hipLaunchKernelGGL(k_qs_recand, dim3(qs_blocks((uint64_t)changed * (2u * J.sweep - 1u))), dim3(256), 0, BR_STREAM, J, P, S, (const uint8_t*)B.text, changed, geo, dirty_dev);
  HIP_CHECK(hipGetLastError());
}

// ---- chains -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_qs_parse(QuickJob J, Lz77Params P, QsTables T, const Segment* __restrict__ segments, const SegEntry* __restrict__ entries,
                                                 Command* __restrict__ cmds, SegExit* __restrict__ exits, const uint32_t* __restrict__ list, uint32_t count) {
  if (blockIdx.x >= count) return;
  const uint32_t k = list ? list[blockIdx.x] : blockIdx.x;
  const Segment seg = segments[k];
  if (T.own) T.own += (size_t)blockIdx.x * T.own_stride;
  br_quick_segment(J, P, T, seg, entries[k], cmds + seg.cmd_base, exits + k);
}
void lz77_qspec_parse(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list_dev, uint32_t count,
                      uint32_t* own_tables, uint32_t own_stride) {
  if (count == 0) return;
  QsTables T;
  T.text = B.text;
  T.cand = S.cand;
  T.flags = S.flags;
  T.own = own_tables;
  T.own_stride = own_stride;
  T.dict = qspec_dict_tables();
  hipLaunchKernelGGL(k_qs_parse, dim3(count), dim3(64), 0, BR_STREAM, J, P, T, (const Segment*)B.segments, (const SegEntry*)B.entries, B.cmds, B.exits, list_dev, count);
  HIP_CHECK(hipGetLastError());
}

void lz77_qspec_parse_custom(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const Segment* segments_dev, const SegEntry* entries_dev,
                             SegExit* exits_dev, uint32_t count) {
  if (count == 0) return;
  QsTables T;
  T.text = B.text;
  T.cand = S.cand;
  T.flags = S.flags;
  T.dict = qspec_dict_tables();
  hipLaunchKernelGGL(k_qs_parse, dim3(count), dim3(64), 0, BR_STREAM, J, P, T, segments_dev, entries_dev, B.cmds, exits_dev, (const uint32_t*)nullptr, count);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_qs_gather_exits(const SegExit* __restrict__ exits, const uint32_t* __restrict__ list, uint32_t count, SegExit* __restrict__ out) {
  // one 4-byte word per thread
  constexpr uint32_t kWords = sizeof(SegExit) / 4;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count * kWords) return;
  const uint32_t r = i / kWords, w = i % kWords;
  ((uint32_t*)out)[i] = ((const uint32_t*)(exits + list[r]))[w];
}
void lz77_qspec_gather_exits(const Lz77Buffers& B, const uint32_t* list_dev, uint32_t count, SegExit* out_dev) {
  if (count == 0) return;
  static_assert(sizeof(SegExit) % 4 == 0, "SegExit is made of 32-bit words");
  hipLaunchKernelGGL(k_qs_gather_exits, dim3(qs_blocks((uint64_t)count * (sizeof(SegExit) / 4))), dim3(256), 0, BR_STREAM, (const SegExit*)B.exits, list_dev, count, out_dev);
  HIP_CHECK(hipGetLastError());
}

// ---- one chain per block on a table of its own: the tables, and where the filings of a launch begin to differ ---------------------------
__global__ __launch_bounds__(256) void k_qs_block_tables(QuickJob J, QuickSpec S, const Segment* __restrict__ segments, const uint32_t* __restrict__ list, uint32_t first,
                                                        uint32_t* __restrict__ tables, uint32_t stride) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= S.slots) return;
  const uint32_t i = first + blockIdx.y;
  const uint32_t upto = segments[list ? list[i] : i].start;
  const uint32_t lo = S.slot_first[s], hi = S.slot_first[s + 1];
  tables[(size_t)i * stride + s] = qs_candidate(J, S, s, qs_rank_in_slot_guess(J, S.ev_id, lo, hi, upto, S.n));
}
void lz77_qspec_block_tables(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list_dev, uint32_t count,
                             uint32_t* tables, uint32_t stride) {
  for (uint32_t first = 0; first < count; first += 32768u) {  // (grid.y is limited to 65 535)
    const uint32_t n = count - first < 32768u ? count - first : 32768u;
    hipLaunchKernelGGL(k_qs_block_tables, dim3(qs_blocks(S.slots), n), dim3(256), 0, BR_STREAM, J, S, (const Segment*)B.segments, list_dev, first, tables, stride);
  }
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_qs_first_change(QuickSpec S, const Segment* __restrict__ segments, const uint32_t* __restrict__ list, uint32_t count,
                                                        uint32_t* __restrict__ out) {
  if (blockIdx.x >= count) return;
  const Segment seg = segments[list ? list[blockIdx.x] : blockIdx.x];
  const uint8_t filing_bits = (uint8_t)(kQsStored | kQsQuad | 0x18u);
  for (uint32_t q = seg.start + threadIdx.x; q < seg.end; q += 256u)
    if ((S.flags[q] ^ S.flags_prev[q]) & filing_bits) {
      atomicMin(out, q);
      return;
    }
}
void lz77_qspec_first_change(const Lz77Buffers& B, const QuickSpec& S, const uint32_t* list_dev, uint32_t count, uint32_t* out_dev) {
  dev_memset(out_dev, 0xff, 4);
  if (count == 0) return;
  hipLaunchKernelGGL(k_qs_first_change, dim3(count), dim3(256), 0, BR_STREAM, S, (const Segment*)B.segments, list_dev, count, out_dev);
  HIP_CHECK(hipGetLastError());
}

// ---- the table behind the text (for the next piece of a stream) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_qs_table(QuickJob J, QuickSpec S, uint32_t upto, uint32_t* __restrict__ out) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= S.slots) return;
  const uint32_t lo = S.slot_first[s], hi = S.slot_first[s + 1];
  out[s] = qs_candidate(J, S, s, upto == 0xffffffffu ? hi : qs_rank_in_slot(J, S.ev_id, lo, hi, upto));
}
void lz77_qspec_table(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t upto, uint32_t* out) {
  hipLaunchKernelGGL(k_qs_table, dim3(qs_blocks(S.slots)), dim3(256), 0, BR_STREAM, J, S, upto, out);
  HIP_CHECK(hipGetLastError());
}

}  // namespace brotli_mi355x
