// lz77_kernels.hip -- gfx950 kernels of the backward-reference (LZ77) stage.
//
//   k_compute_keys      HashBytes of every position            (mod.rs:990-992, 1138-1140, 1521-1525)
//   k_radix_*           stable LSD radix sort of positions by key (two 8-bit passes)
//   k_rank_*            prefix sum of the "stored" flags in (key,pos) order -> rank / sorted / key_base
//   k_parse_segments    one wavefront per segment: the speculative greedy/lazy parse (lz77_chain.h)
//
// All of this is integer/byte work bound by HBM/L2 latency and bandwidth; no MFMA.  Wave width is 64
// (one wavefront per parse chain), LDS stages the per-search candidate records and the radix digit
// counters.
#include <hip/hip_runtime.h>
#include <stdexcept>
#include <string>
#include <utility>
#include <string.h>
#include <mutex>
#include <vector>

#include "device_api.h"
#include "lz77_chain.h"
#include "lz77_rows.h"
#include "lz77_groups.h"
#include "device_scan.h"
#include "lz77_parse_args.h"
#include "zopfli_device.h"

namespace brotli_mi355x {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------------ keys
// dict_items (optional, candidate rows with the static dictionary on): for every position the two hash items
// SearchInStaticDictionary would probe there (mod.rs:1942-1988: kStaticDictionaryHash[2 * hash14(first four bytes) + i]), low
// half = probe 0.  A chain refills its window of them with one coalesced load instead of text -> hash -> table.
__global__ __launch_bounds__(256) void k_compute_keys(const uint8_t* __restrict__ text, uint16_t* __restrict__ keys,
                                                      uint32_t n, uint32_t valid_n, uint32_t kind, uint32_t bucket_bits,
                                                      uint64_t hash_mask, uint32_t* __restrict__ run_samples,
                                                      const uint16_t* __restrict__ dict_hash, uint32_t* __restrict__ dict_items) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    if (dict_items != nullptr) {  // (the text is padded by 64 zero bytes)
      const uint32_t h = ((br_load32(text + i) * 0x1e35a7bdu) >> (32 - 14)) << 1;
      dict_items[i] = (uint32_t)dict_hash[h] | ((uint32_t)dict_hash[h + 1] << 16);
    }
    uint32_t key = 0xffffu;
    if (i < valid_n) {
      // sample (every 64th position): does a run of one byte start here?  Enough of those switch on the run table.
      if ((i & 63u) == 0 && i + 16 <= n) {
        const uint64_t v = br_load64(text + i);
        // (the count only matters up to the threshold the host compares it with: no need to hammer one address)
        if (v == (v & 0xffull) * 0x0101010101010101ull && br_load64(text + i + 8) == v && *(volatile uint32_t*)run_samples < 4096u)
          atomicAdd(run_samples, 1u);
      }
      if (kind == 6) {
        const uint64_t v = (br_load64(text + i) & hash_mask) * 0x1fe35a7bd3579bd3ull;
        key = (uint32_t)(v >> (64 - bucket_bits));
      } else {
        const uint32_t v = br_load32(text + i) * 0x1e35a7bdu;
        key = v >> (32 - bucket_bits);
      }
    }
    keys[i] = (uint16_t)key;
  }
  if (dict_items != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {  // (positions behind the end are looked up as position n)
    const uint32_t h = ((br_load32(text + n) * 0x1e35a7bdu) >> (32 - 14)) << 1;
    dict_items[n] = (uint32_t)dict_hash[h] | ((uint32_t)dict_hash[h + 1] << 16);
  }
}

void lz77_compute_keys(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  const uint32_t valid_n = n >= P.htl ? n - P.htl + 1 : 0;
  const uint64_t hash_mask = P.hasher_kind == 6 ? (0xffffffffffffffffull >> (64 - 8 * P.hash_len)) : 0;
  if (n == 0) return;
  uint32_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dev_memset(B.changed_count + 8, 0, 4);
  hipLaunchKernelGGL(k_compute_keys, dim3(blocks), dim3(256), 0, BR_STREAM, B.text, B.keys, n, valid_n, P.hasher_kind, P.bucket_bits,
                     hash_mask, B.changed_count + 8, dev_tables().dict_hash, B.dict_items);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ flags
// First guess of the "stored" flags: every position of the input is in the hash table except the ones
// the reference never stores: positions at or after store_end (mod.rs:2397-2404) unless the next
// block's StitchToPreviousBlock stores them (mod.rs:210-222).
__global__ __launch_bounds__(64) void k_init_flag_tails(const Segment* __restrict__ segments, uint32_t num_segments, uint32_t htl,
                                                         uint8_t* __restrict__ flags) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_segments) return;
  const Segment g = segments[k];
  if (!(g.flags & kSegFirstInBlock)) return;
  const uint32_t bs = g.blk_start, be = g.blk_end;
  if (g.block_index == 0 && be - bs >= htl - 1 && bs >= 3) flags[bs - 3] = flags[bs - 2] = flags[bs - 1] = 1;
  const uint32_t store_end = (be - bs >= htl) ? be - htl + 1 : bs;
  for (uint32_t q = store_end; q < be; ++q) flags[q] = (q + 3 >= be && (g.flags & kSegTailStitched)) ? 1 : 0;
}

void lz77_init_flags(const Lz77Params& P, const Lz77Buffers& B, uint32_t first_block_start, const uint8_t* prefix_flags_host,
                     uint32_t prefix_flags_bytes) {
  // (masked H5 ring entries, Lz77Params::masked_from: texts that have them are parsed by live chains, lz77_live.h; the row
  // and rank kernels below carry kFlagMasked through, which only the live index reads)
  const uint32_t M = P.total_bytes, P0 = P.prefix_bytes, htl = P.htl;
  dev_memset(B.flags[0], 0, (size_t)M + 64);
  if (P0 > htl - 1) HIP_CHECK(hipMemsetAsync(B.flags[0], 1, P0 - (htl - 1), BR_STREAM));  // StoreLookaheadThenStore, mod.rs:224-229
  // the catable raw head (between the prefix and the first searched block) is only stored by the stitch
  if (prefix_flags_host && prefix_flags_bytes) {
    // continuation of a stream: the earlier positions are in the table exactly as the earlier parse left them
    HIP_CHECK(hipMemcpyAsync(B.flags[0], prefix_flags_host, prefix_flags_bytes, hipMemcpyHostToDevice, BR_STREAM));
  }
  if (M > first_block_start) HIP_CHECK(hipMemsetAsync(B.flags[0] + first_block_start, 1, M - first_block_start, BR_STREAM));
  if (P.num_segments) {
    hipLaunchKernelGGL(k_init_flag_tails, dim3((P.num_segments + 63) / 64), dim3(64), 0, BR_STREAM, B.segments, P.num_segments, htl, B.flags[0]);
  }
  HIP_CHECK(hipMemcpyAsync(B.flags[1], B.flags[0], (size_t)M + 64, hipMemcpyDeviceToDevice, BR_STREAM));
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ radix sort
static constexpr uint32_t kSortTile = 4096;  // elements per workgroup (16 rounds of 256)

__global__ __launch_bounds__(256) void k_radix_hist(const uint16_t* __restrict__ keys, uint32_t n, uint32_t shift,
                                                     uint32_t num_tiles, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  for (uint32_t r = 0; r < 16; ++r) {
    const uint32_t i = base + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * num_tiles + blockIdx.x] = h[threadIdx.x];
}

// stable scatter: elements keep their input order inside each digit.  The tile is first reordered by digit in LDS, then
// written out run by run, so that the lanes of a wave write neighbouring addresses (a direct scatter writes one isolated
// 2- and 4-byte element per digit and round).
// `tags`: optional third column (16-bit tag of every position, br_tag16): computed from the text in the first pass
// (tags_in == nullptr, text != nullptr), carried along in the second.
__global__ __launch_bounds__(256) void k_radix_scatter(const uint16_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                        uint16_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                        uint32_t shift, uint32_t num_tiles, const uint32_t* __restrict__ offsets,
                                                        const uint8_t* __restrict__ text, const uint16_t* __restrict__ tags_in,
                                                        uint16_t* __restrict__ tags_out) {
  __shared__ uint32_t gbase[256];   // where digit d of this tile goes in the output
  __shared__ uint32_t lstart[256];  // where digit d starts inside the reordered tile
  __shared__ uint32_t run[256];     // next free local slot of digit d
  __shared__ uint32_t wcount[4][256];
  __shared__ uint32_t wave_total[4];
  __shared__ uint16_t skey[kSortTile];
  __shared__ uint32_t sval[kSortTile];
  __shared__ uint16_t stagv[kSortTile];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t tile_base = blockIdx.x * kSortTile;
  const uint32_t tile_n = n - tile_base < kSortTile ? n - tile_base : kSortTile;
  {
    // digit counts of this tile out of the scanned histogram ([digit][tile] order), then their exclusive scan
    const uint32_t idx = (uint32_t)tid * num_tiles + blockIdx.x;
    const uint32_t here = offsets[idx];
    const uint32_t next = idx + 1 < 256u * num_tiles ? offsets[idx + 1] : n;
    const uint32_t count = next - here;
    gbase[tid] = here;
    uint32_t incl = count;
    for (uint32_t off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if ((uint32_t)lane >= off) incl += up;
    }
    if (lane == 63) wave_total[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int i = 0; i < w; ++i) before += wave_total[i];
    lstart[tid] = before + incl - count;
    run[tid] = before + incl - count;
  }
  __syncthreads();
  for (uint32_t r = 0; r < 16; ++r) {
    const uint32_t i = tile_base + r * 256 + tid;
    const bool valid = i < n;
    uint32_t key = 0, val = 0, d = 0, tag = 0;
    if (valid) {
      key = keys_in[i];
      val = vals_in ? vals_in[i] : i;
      d = (key >> shift) & 255u;
      if (tags_out) tag = tags_in ? tags_in[i] : br_tag16(br_load32(text + val));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) wcount[w][lane * 4 + j] = 0;
    __syncthreads();
    // lanes of this wavefront that hold the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long b = __ballot(one);
      peers &= one ? b : ~b;
    }
    const uint32_t rank_in_wave = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank_in_wave == 0) wcount[w][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t at = run[d] + rank_in_wave;
      for (int i2 = 0; i2 < w; ++i2) at += wcount[i2][d];
      skey[at] = (uint16_t)key;
      sval[at] = val;
      if (tags_out) stagv[at] = (uint16_t)tag;
    }
    __syncthreads();
    run[tid] += wcount[0][tid] + wcount[1][tid] + wcount[2][tid] + wcount[3][tid];
    __syncthreads();
  }
  for (uint32_t at = tid; at < tile_n; at += 256) {
    const uint32_t key = skey[at];
    const uint32_t d = (key >> shift) & 255u;
    const uint32_t dst = gbase[d] + (at - lstart[d]);
    keys_out[dst] = (uint16_t)key;
    vals_out[dst] = sval[at];
    if (tags_out) tags_out[dst] = stagv[at];
  }
}

size_t lz77_sort_tmp_bytes(uint32_t total_bytes) {
  const size_t n = total_bytes;
  const size_t tiles = (n + kSortTile - 1) / kSortTile + 1;
  // ping-pong keys + values, digit histograms, scan scratch
  // (the incremental re-rank uses the same scratch for two u32 arrays in (key,pos) index space)
  return n * 10 + 2048 + tiles * 256 * 4 + (tiles * 256 / kScanTile + 1024) * 8 + (n / kScanTile + 1024) * 8 + 4096;
}

void lz77_key_ranges(const Lz77Params& P, const Lz77Buffers& B);

void lz77_sort_by_key(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  if (n == 0) return;
  const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
  uint8_t* tmp = (uint8_t*)B.sort_tmp;
  uint16_t* keys_tmp = (uint16_t*)tmp;
  tmp += ((size_t)n * 2 + 255) & ~(size_t)255;
  uint32_t* vals_tmp = (uint32_t*)tmp;
  tmp += ((size_t)n * 4 + 255) & ~(size_t)255;
  uint16_t* tags_tmp = (uint16_t*)tmp;
  tmp += ((size_t)n * 2 + 255) & ~(size_t)255;
  uint32_t* hist = (uint32_t*)tmp;
  tmp += (size_t)tiles * 256 * 4;
  uint32_t* scratch = (uint32_t*)tmp;
  // pass 1: low 8 bits, keys -> tmp
  hipLaunchKernelGGL(k_radix_hist, dim3(tiles), dim3(256), 0, BR_STREAM, B.keys, n, 0u, tiles, hist);
  exclusive_scan_u32(hist, tiles * 256, scratch);
  hipLaunchKernelGGL(k_radix_scatter, dim3(tiles), dim3(256), 0, BR_STREAM, B.keys, (const uint32_t*)nullptr, keys_tmp, vals_tmp, n, 0u, tiles,
                     hist, (const uint8_t*)B.text, (const uint16_t*)nullptr, B.stag ? tags_tmp : (uint16_t*)nullptr);
  // pass 2: high 8 bits, tmp -> by_key / sorted_keys
  hipLaunchKernelGGL(k_radix_hist, dim3(tiles), dim3(256), 0, BR_STREAM, keys_tmp, n, 8u, tiles, hist);
  exclusive_scan_u32(hist, tiles * 256, scratch);
  hipLaunchKernelGGL(k_radix_scatter, dim3(tiles), dim3(256), 0, BR_STREAM, keys_tmp, vals_tmp, B.sorted_keys, B.by_key, n, 8u, tiles, hist,
                     (const uint8_t*)B.text, (const uint16_t*)tags_tmp, B.stag);
  HIP_CHECK(hipGetLastError());
  lz77_key_ranges(P, B);
}

// ------------------------------------------------------------------------------------------ rank
// The "hash table" of the reference as a function of the stored flags.  Index space = (key, position) order
// (by_key): key k owns the slots [key_first[k], key_last[k]); the stored positions of that key are compacted
// to the front of its slots in `sorted`, and info[p] = {slot one past ... i.e. key_first + local rank, local rank}
// where local rank = number of stored same-key positions before p.  Because a key's slots do not depend on any
// other key, a handful of flag changes can be applied key by key (k_rerank_keys) without touching the rest.
__global__ __launch_bounds__(256) void k_key_ranges(const uint16_t* __restrict__ sorted_keys, uint32_t n, uint32_t* __restrict__ key_first,
                                                     uint32_t* __restrict__ key_last) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t k = sorted_keys[i];
  if (i == 0 || sorted_keys[i - 1] != k) key_first[k] = i;
  if (i + 1 == n || sorted_keys[i + 1] != k) key_last[k] = i + 1;
}

// pass A: gather the stored bits into (key,pos) order (one byte each) and sum them per tile.
// `initial`: the flags are still the first guess of lz77_init_flags, which is 1 everywhere except next to the block
// ends and in front of the first block -- only positions there are fetched (a random 1-byte gather costs a 64-byte line).
struct InitialFlagGeometry {
  uint32_t enabled, first_block_start, prefix_bytes, block_bytes, total_bytes, prefix_stored_end;
};

__global__ __launch_bounds__(256) void k_rank_gather(const uint32_t* __restrict__ by_key, const uint8_t* __restrict__ flags, uint32_t n,
                                                      uint8_t* __restrict__ fbits, uint32_t* __restrict__ tile_sums,
                                                      InitialFlagGeometry ig, uint32_t masked_bits) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t local = 0, packed = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) {
      const uint32_t p = by_key[base + j];
      uint32_t f;
      bool known = false;
      if (ig.enabled) {
        if (p >= ig.first_block_start) {
          known = ((p - ig.prefix_bytes) % ig.block_bytes) + 16 < ig.block_bytes && p + 16 < ig.total_bytes;
        } else {
          known = p + 16 < ig.prefix_stored_end;
        }
      }
      const uint32_t fl = known ? 1u : (uint32_t)flags[p];
      f = fl & 1u;
      // (candidate rows with masked H5 entries modelled: kFlagMasked becomes kSlotMasked, lz77_rows.h)
      packed |= (f | ((masked_bits && (fl & kFlagMasked)) ? kSlotMasked : 0u)) << (8 * j);
      local += f;
    }
  if (base < n) *(uint32_t*)(fbits + base) = packed;
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// pass K: global count of stored positions in front of every key's first slot
__global__ __launch_bounds__(256) void k_key_bases(const uint8_t* __restrict__ fbits, const uint32_t* __restrict__ tile_offsets,
                                                    const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_last,
                                                    uint32_t* __restrict__ key_base) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 65536) return;
  const uint32_t i0 = key_first[k];
  if (key_last[k] <= i0) return;
  const uint32_t tile = i0 / kScanTile;
  uint32_t g = tile_offsets[tile];
  uint32_t i = tile * kScanTile;
  for (; i + 4 <= i0; i += 4) g += (*(const uint32_t*)(fbits + i) * 0x01010101u) >> 24;
  for (; i < i0; ++i) g += fbits[i];
  key_base[k] = g;
}

// Second word of an info record: how many entries of the key's ring the reference looks at -- its u16 counter num[key]
// (mod.rs:1752-1760) counts the insertions since the start of the STREAM (base = those in front of this text), but no
// more entries than were inserted within this text can be reached (older ones lie beyond max_backward: the text of a
// later piece of a stream starts at least a window in front of its input).
// (positions behind a hasher reset, Lz77Params::reset_pos: only the insertions since the reset count, and nothing older
// can be reached)
__device__ __forceinline__ uint32_t ring_count_at(uint32_t p, uint32_t local_rank, uint32_t key, const uint32_t* count_base, uint32_t reset_pos,
                                                  const uint32_t* reset_counts);
__device__ __forceinline__ uint32_t ring_count(uint32_t local_rank, uint32_t base) {
  const uint32_t num = (local_rank + base) & 0xffffu;
  return num < local_rank ? num : local_rank;
}

__device__ __forceinline__ uint32_t ring_count_at(uint32_t p, uint32_t local_rank, uint32_t key, const uint32_t* count_base, uint32_t reset_pos,
                                                  const uint32_t* reset_counts) {
  if (reset_pos != 0 && p >= reset_pos) {
    const uint32_t since = local_rank - reset_counts[key];
    return since & 0xffffu;  // (since <= local_rank: what was inserted before the reset is out of reach)
  }
  return ring_count(local_rank, count_base ? count_base[key] : 0u);
}

// pass B: local ranks -> sorted / info
__global__ __launch_bounds__(256) void k_rank_apply(const uint32_t* __restrict__ by_key, const uint16_t* __restrict__ sorted_keys,
                                                     const uint8_t* __restrict__ fbits, uint32_t n, const uint32_t* __restrict__ tile_offsets,
                                                     const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_base,
                                                     uint32_t* __restrict__ sorted, uint2* __restrict__ info,
                                                     const uint32_t* __restrict__ count_base, uint32_t reset_pos,
                                                     const uint32_t* __restrict__ reset_counts, const uint16_t* __restrict__ stag,
                                                     uint16_t* __restrict__ sorted_tag, const uint8_t* __restrict__ flags_if_masked) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t pos[4], f[4], key[4];
  uint32_t local = 0;
  const uint32_t packed = base < n ? *(const uint32_t*)(fbits + base) : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pos[j] = 0;
    key[j] = 0;
    f[j] = (packed >> (8 * j)) & 1u;
    if (base + j < n) {
      pos[j] = by_key[base + j];
      key[j] = sorted_keys[base + j];
    } else {
      f[j] = 0;
    }
    local += f[j];
  }
  uint32_t x = local;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_sum[w] = x;
  __syncthreads();
  uint32_t g = tile_offsets[blockIdx.x] + x - local;
  for (int i = 0; i < w; ++i) g += wave_sum[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) {
      const uint32_t lr = g - key_base[key[j]];
      const uint32_t slot = key_first[key[j]] + lr;
      info[pos[j]] = make_uint2(slot, ring_count_at(pos[j], lr, key[j], count_base, reset_pos, reset_counts));
      if (f[j]) {
        // (a masked H5 ring entry is held as position | kMaskedEntry: the probe breaks on it, lz77_chain.h)
        sorted[slot] = pos[j] | ((flags_if_masked != nullptr && (flags_if_masked[pos[j]] & kFlagMasked)) ? kMaskedEntry : 0u);
        if (sorted_tag) sorted_tag[slot] = stag[base + j];
      }
      g += f[j];
    }
  }
}

void lz77_key_ranges(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  dev_memset(B.key_first, 0, 65537 * 4);
  dev_memset(B.key_last, 0, 65537 * 4);
  if (n == 0) return;
  hipLaunchKernelGGL(k_key_ranges, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, B.sorted_keys, n, B.key_first, B.key_last);
  HIP_CHECK(hipGetLastError());
}

void lz77_rank_flags(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const RankInitialHint* initial) {
  const uint32_t n = P.total_bytes;
  if (n == 0) return;
  InitialFlagGeometry ig{};
  if (initial) {
    ig.enabled = 1;
    ig.first_block_start = initial->first_block_start;
    ig.prefix_bytes = P.prefix_bytes;
    ig.block_bytes = initial->block_bytes;
    ig.total_bytes = n;
    ig.prefix_stored_end = (initial->prefix_is_dictionary && P.prefix_bytes > P.htl - 1) ? P.prefix_bytes - (P.htl - 1) : 0;  // StoreLookaheadThenStore, mod.rs:224-229
  }
  const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
  uint32_t* tile_sums = (uint32_t*)B.sort_tmp;
  uint32_t* scratch = tile_sums + tiles + 64;
  if (P.reset_pos) lz77_key_counts(P, B, which, P.reset_vis, B.reset_counts, false);
  // (the rank passes sum the per-slot bytes as they are: no kSlotMasked here, k_rank_apply reads kFlagMasked from the flags)
  hipLaunchKernelGGL(k_rank_gather, dim3(tiles), dim3(256), 0, BR_STREAM, B.by_key, B.flags[which], n, B.fbits, tile_sums, ig, 0u);
  exclusive_scan_u32(tile_sums, tiles, scratch);
  hipLaunchKernelGGL(k_key_bases, dim3(256), dim3(256), 0, BR_STREAM, B.fbits, tile_sums, B.key_first, B.key_last, B.key_base);
  hipLaunchKernelGGL(k_rank_apply, dim3(tiles), dim3(256), 0, BR_STREAM, B.by_key, B.sorted_keys, B.fbits, n, tile_sums, B.key_first, B.key_base,
                     B.sorted[rbuf], (uint2*)B.info[rbuf], B.count_base, P.reset_pos, B.reset_counts, B.stag, B.stag ? B.sorted_tag[rbuf] : nullptr,
                     P.masked_from != kNeverMasked ? (const uint8_t*)B.flags[which] : (const uint8_t*)nullptr);
  HIP_CHECK(hipGetLastError());
}

// marks the chain(s) that searched position p
// rows_lo / rows_hi (optional, candidate rows): lowest / highest such position per segment (Lz77Buffers::rows_changed_lo / _hi)
__device__ __forceinline__ void note_rows_changed(uint32_t k, uint32_t p, uint32_t* rows_lo, uint32_t* rows_hi) {
  if (rows_lo == nullptr) return;
  atomicMin(&rows_lo[k], p);
  atomicMax(&rows_hi[k], p);
}
__device__ __forceinline__ void mark_dirty(uint32_t p, const SegGeometry& geo, uint8_t* __restrict__ dirty, uint32_t* rows_lo = nullptr,
                                           uint32_t* rows_hi = nullptr) {
  const uint32_t blk = (p - geo.prefix_bytes) / geo.block_bytes;
  const uint32_t bs = blk == 0 ? geo.first_block_start : geo.prefix_bytes + blk * geo.block_bytes;
  const uint32_t off = p - bs;
  const uint32_t seg_bytes = geo.block_segment_bytes[blk];
  uint32_t k = geo.block_first_segment[blk] + off / seg_bytes;
  if (k >= geo.block_first_segment[blk + 1]) k = geo.block_first_segment[blk + 1] - 1;
  dirty[k] = 1;
  note_rows_changed(k, p, rows_lo, rows_hi);
  // a lazy probe just behind a segment boundary belongs to the previous chain
  if (k > 0 && (off % seg_bytes) < 8) {
    dirty[k - 1] = 1;
    note_rows_changed(k - 1, p, rows_lo, rows_hi);
  }
}

// A chain evaluates the lazy alternative one position ahead, up to four times in a row (mod.rs:2455-2480), so its last
// probes can fall on the first positions of the NEXT segment.  Their "searched" flags are written by that segment's
// chain (flag ownership) -- possibly a round later -- so for validation the first 8 positions of a segment count as
// searched by the chain in front whatever their flag says.  Returns that chain's index, or 0xffffffff.
__device__ __forceinline__ uint32_t chain_in_front_if_near_boundary(uint32_t p, const SegGeometry& geo) {
  const uint32_t blk = (p - geo.prefix_bytes) / geo.block_bytes;
  const uint32_t bs = blk == 0 ? geo.first_block_start : geo.prefix_bytes + blk * geo.block_bytes;
  const uint32_t off = p - bs;
  const uint32_t seg_bytes = geo.block_segment_bytes[blk];
  const uint32_t first = geo.block_first_segment[blk];
  uint32_t k = first + off / seg_bytes;
  if (k >= geo.block_first_segment[blk + 1]) return 0xffffffffu;  // (the tail of the block belongs to its last segment)
  if (k == first || (off % seg_bytes) >= 8) return 0xffffffffu;
  return k - 1;
}

// Incremental update after a few flag changes: the slots of every changed key are cut into chunks of
// kRerankChunk; one workgroup per chunk (1) counts the stored bits, (2) recomputes the local ranks into scratch,
// (3) marks the chains that searched a position of that key whose candidate list is no longer what they saw and
// (4) commits the new ranks in place.  Keys that did not change are not touched.
__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* wave_sum) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = v;
  __syncthreads();
  return wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

__global__ __launch_bounds__(256) void k_rerank_count(const RerankChunk* __restrict__ chunks, const uint32_t* __restrict__ by_key,
                                                       const uint8_t* __restrict__ flags, uint32_t* __restrict__ sums) {
  __shared__ uint32_t wave_sum[4];
  const RerankChunk c = chunks[blockIdx.x];
  uint32_t local = 0;
  for (uint32_t i = c.begin + threadIdx.x; i < c.end; i += 256) local += flags[by_key[i]] & 1u;
  const uint32_t total = block_sum_256(local, wave_sum);
  if (threadIdx.x == 0) sums[c.my_sum] = total;
}

__global__ __launch_bounds__(256) void k_rerank_apply(const RerankChunk* __restrict__ chunks, const uint32_t* __restrict__ sums,
                                                       const uint32_t* __restrict__ by_key, const uint8_t* __restrict__ flags,
                                                       uint32_t* __restrict__ rank_tmp, uint32_t* __restrict__ sorted_tmp) {
  __shared__ uint32_t wave_sum[4];
  __shared__ uint32_t running;
  const RerankChunk c = chunks[blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t before = 0;
  for (uint32_t j = c.first_sum + threadIdx.x; j < c.my_sum; j += 256) before += sums[j];
  before = block_sum_256(before, wave_sum);
  if (threadIdx.x == 0) running = before;
  __syncthreads();
  for (uint32_t tile = c.begin; tile < c.end; tile += kScanTile) {
    const uint32_t base = tile + threadIdx.x * 4;
    uint32_t pos[4], f[4], masked[4], local = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pos[j] = 0;
      f[j] = 0;
      masked[j] = 0;
      if (base + j < c.end) {
        pos[j] = by_key[base + j];
        const uint32_t fl = flags[pos[j]];
        f[j] = fl & 1u;
        masked[j] = (fl & kFlagMasked) ? kMaskedEntry : 0u;
      }
      local += f[j];
    }
    uint32_t x = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sum[w] = x;
    __syncthreads();
    uint32_t g = running + x - local;
    for (int i = 0; i < w; ++i) g += wave_sum[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (base + j < c.end) {
        rank_tmp[base + j] = g;
        if (f[j]) sorted_tmp[c.key_lo + g] = pos[j] | masked[j];
        g += f[j];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) running += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    __syncthreads();
  }
}

// a searched position whose candidate list changed: queued for lz77_recheck_searches, or -- no list, list full -- its
// segment is parsed again
__device__ __forceinline__ void list_or_mark(uint32_t p, const SegGeometry& geo, uint8_t* dirty, uint32_t* recheck_list, uint32_t* recheck_count,
                                             uint32_t recheck_cap) {
  if (recheck_list != nullptr) {
    const uint32_t at = atomicAdd(recheck_count, 1u);
    if (at < recheck_cap) {
      recheck_list[at] = p;
      return;
    }
  }
  mark_dirty(p, geo, dirty);
}

__global__ __launch_bounds__(256) void k_rerank_check(const uint8_t* __restrict__ text, const RerankChunk* __restrict__ chunks,
                                                       const uint32_t* __restrict__ by_key,
                                                       const uint8_t* __restrict__ flags, const uint32_t* __restrict__ sorted,
                                                       const uint2* __restrict__ info, const uint32_t* __restrict__ rank_tmp,
                                                       const uint32_t* __restrict__ sorted_tmp, SegGeometry geo,
                                                       uint8_t* __restrict__ dirty, const uint32_t* __restrict__ count_base,
                                                       const uint16_t* __restrict__ keys, uint32_t reset_pos,
                                                       const uint32_t* __restrict__ reset_counts, uint32_t* __restrict__ recheck_list,
                                                       uint32_t* __restrict__ recheck_count, uint32_t recheck_cap) {
  const RerankChunk c = chunks[blockIdx.x];
  for (uint32_t i = c.begin + threadIdx.x; i < c.end; i += 256) {
    const uint32_t p = by_key[i];
    if (p < geo.first_block_start) continue;
    const bool searched = (flags[p] & kFlagSearched) != 0;
    const uint32_t in_front = searched ? 0xffffffffu : chain_in_front_if_near_boundary(p, geo);
    if (!searched && in_front == 0xffffffffu) continue;
    const uint2 a = info[p];
    const uint32_t rb = rank_tmp[i];
    const uint32_t na = min(a.y & 0xffffu, geo.block_size), nb = min(ring_count_at(p, rb, keys[p], count_base, reset_pos, reset_counts), geo.block_size);
    bool same = na == nb;
    for (uint32_t j = 0; same && j < na; ++j) same = sorted[a.x - 1 - j] == sorted_tmp[c.key_lo + rb - 1 - j];
    if (!same && br_row_change_matters(text, p, sorted + a.x - 1, na, sorted_tmp + c.key_lo + rb - 1, nb)) {
      if (searched) list_or_mark(p, geo, dirty, recheck_list, recheck_count, recheck_cap);
      else dirty[in_front] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_rerank_commit(const RerankChunk* __restrict__ chunks, const uint32_t* __restrict__ by_key,
                                                        const uint8_t* __restrict__ flags, uint32_t* __restrict__ sorted,
                                                        uint2* __restrict__ info, const uint32_t* __restrict__ rank_tmp,
                                                        const uint32_t* __restrict__ count_base, const uint16_t* __restrict__ keys,
                                                        uint32_t reset_pos, const uint32_t* __restrict__ reset_counts,
                                                        const uint16_t* __restrict__ stag, uint16_t* __restrict__ sorted_tag) {
  const RerankChunk c = chunks[blockIdx.x];
  for (uint32_t i = c.begin + threadIdx.x; i < c.end; i += 256) {
    const uint32_t p = by_key[i];
    const uint32_t rb = rank_tmp[i];
    info[p] = make_uint2(c.key_lo + rb, ring_count_at(p, rb, keys[p], count_base, reset_pos, reset_counts));
    if (flags[p] & 1u) {
      sorted[c.key_lo + rb] = p | ((flags[p] & kFlagMasked) ? kMaskedEntry : 0u);
      if (sorted_tag) sorted_tag[c.key_lo + rb] = stag[i];
    }
  }
}

void lz77_rerank_keys(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const RerankChunk* chunks_dev, uint32_t num_chunks,
                      uint32_t* sums_dev, const SegGeometry& geo, uint8_t* dirty_dev) {
  if (num_chunks == 0 || P.total_bytes == 0) return;
  uint32_t* rank_tmp = (uint32_t*)B.sort_tmp;
  uint32_t* sorted_tmp = rank_tmp + (((size_t)P.total_bytes + 63) & ~(size_t)63);
  if (P.reset_pos) lz77_key_counts(P, B, which, P.reset_vis, B.reset_counts, false);
  hipLaunchKernelGGL(k_rerank_count, dim3(num_chunks), dim3(256), 0, BR_STREAM, chunks_dev, B.by_key, B.flags[which], sums_dev);
  hipLaunchKernelGGL(k_rerank_apply, dim3(num_chunks), dim3(256), 0, BR_STREAM, chunks_dev, sums_dev, B.by_key, B.flags[which], rank_tmp, sorted_tmp);
  hipLaunchKernelGGL(k_rerank_check, dim3(num_chunks), dim3(256), 0, BR_STREAM, B.text, chunks_dev, B.by_key, B.flags[which], B.sorted[rbuf],
                     (const uint2*)B.info[rbuf], rank_tmp, sorted_tmp, geo, dirty_dev, B.count_base, B.keys, P.reset_pos, B.reset_counts,
                     B.recheck_list, B.recheck_count, B.recheck_cap);
  hipLaunchKernelGGL(k_rerank_commit, dim3(num_chunks), dim3(256), 0, BR_STREAM, chunks_dev, B.by_key, B.flags[which], B.sorted[rbuf],
                     (uint2*)B.info[rbuf], rank_tmp, B.count_base, B.keys, P.reset_pos, B.reset_counts, B.stag,
                     B.stag ? B.sorted_tag[rbuf] : nullptr);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ candidate rows
// See kRowEntries in lz77_types.h and lz77_rows.h.  rows[p] is a pure function of the per-slot bytes (stored bit, wrap
// mark) of the slots in front of p's slot in (key, position) order: the lookback ends after `depth` stored slots, so a
// flag change reaches the rows of the next `depth` stored positions of its key and no further.
struct RowArgs {
  const uint32_t* by_key;
  const uint16_t* sorted_keys;
  const uint16_t* stag;
  const uint8_t* fbits;
  const unsigned long long* smask;  // stored bits, one word per 64 slots
  const uint32_t* gprev;            // per 64 slots: 1 + the last stored slot in front of them
  const uint32_t* key_first;
  const uint32_t* key_last;
  uint32_t* rows;
  const uint8_t* flags;  // newest per-position flags (the searched bit decides who is affected by a changed row)
  uint32_t n, depth, max_backward_limit;
  uint32_t reset_pos, reset_vis, ring_mask;  // Lz77Params
  uint32_t validate;     // compare with the row in memory and mark the chains that searched a position whose row changed
  SegGeometry geo;
  uint8_t* dirty;
  uint32_t* rows_lo;     // optional: Lz77Buffers::rows_changed_lo / _hi
  uint32_t* rows_hi;
  const uint32_t* ctl;   // conditional launches: run only if ctl[kCtlNeedFull] != 0
  uint32_t* walk_counter;  // == ctl, writable (k_update_rows)
  uint32_t conditional;
  const unsigned long long* pot;  // Lz77Buffers::pot / pot_state (null: no mask)
  const uint32_t* pot_state;
  const uint32_t* flip_cells;     // Lz77Buffers::flip_cells (null: not used in this update)
  uint32_t cell_shift, cells_per_key, cell_words;
};
// device-side control words of lz77_rows_update
enum RowCtl : uint32_t { kCtlNeedFull = 0, kCtlVirtual = 1, kCtlWalked = 2, kCtlWrapKeyFlip = 3, kCtlWords = 4 };
// kCtlWrapKeyFlip: some flag flipped in a key whose ring counter can wrap (>= 65 536 slots; every key when the counters
// run on from the stream in front or start over at a hasher reset) -- only then can wrap marks have moved this round

__device__ __forceinline__ void row_changed(const RowArgs& a, uint32_t p) {
  if (p < a.geo.first_block_start) return;
  if (a.flags[p] & kFlagSearched) {
    mark_dirty(p, a.geo, a.dirty, a.rows_lo, a.rows_hi);
  } else {
    const uint32_t k = chain_in_front_if_near_boundary(p, a.geo);
    if (k != 0xffffffffu) {
      a.dirty[k] = 1;
      note_rows_changed(k, p, a.rows_lo, a.rows_hi);
    }
  }
}

// All rows: one workgroup per tile of kRowTile slots (+ the kRowHalo slots in front of it), everything staged in LDS.
// The stored slots of the span are compacted first, so that the lookback of a slot is a walk over consecutive compact
// entries (no skipping of unstored slots, at most `depth` steps); every wave stages the rows of 64 slots in LDS and
// writes them out as whole 64-byte lines.
static constexpr uint32_t kRowTile = kScanTile, kRowHalo = 256, kRowSpan = kRowTile + kRowHalo;
static constexpr uint32_t kMaxRowWalk = 2048;  // slots one change may walk in k_update_rows before the full rebuild takes over
static_assert(kRowSpan == 256 * 5, "five span entries per thread");

__global__ __launch_bounds__(256) void k_build_rows(RowArgs a) {
  if (a.conditional && a.ctl[kCtlNeedFull] == 0) return;
  if (a.conditional && a.validate && a.pot != nullptr && a.pot_state[0] != 0 && a.pot_state[2] == 0) return;  // (k_validate_listed_rows has been over the rows that can change)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __shared__ uint32_t s_pos[kRowSpan];
  __shared__ uint32_t s_tk[kRowSpan];  // tag | key << 16
  __shared__ uint8_t s_fb[kRowSpan];
  __shared__ uint16_t s_rank[kRowSpan];  // stored slots of the span in front of each entry
  __shared__ uint2 c_ent[kRowSpan];      // the stored slots, compacted: {position | wrap mark << 31, tag | key << 16}
  __shared__ uint32_t wave_sum[4];
  __shared__ alignas(16) uint32_t rowbuf[4][64 * kRowEntries];
  const uint32_t base = blockIdx.x * kRowTile;
  const uint32_t lo = base >= kRowHalo ? base - kRowHalo : 0u;
  const uint32_t hi = min(base + kRowTile, a.n);
  const uint32_t span = hi - lo;
  if (a.validate && a.flip_cells != nullptr && a.flip_cells[a.cell_words] == 0 && base < hi) {
    // a tile of one key (slots are in position order inside a key) whose cells are all clear: every row of it is what it was
    const uint32_t key = a.sorted_keys[base];
    if (key == a.sorted_keys[hi - 1] && a.key_last[key] - a.key_first[key] < 65536u) {
      const uint32_t p0 = a.by_key[base], p1 = a.by_key[hi - 1];
      const uint32_t first = (p0 > a.max_backward_limit ? p0 - a.max_backward_limit : 0u) >> a.cell_shift, last = p1 >> a.cell_shift;
      const uint32_t cell0 = key * a.cells_per_key;
      bool any = false;
      for (uint32_t c = cell0 + first; c <= cell0 + last; ++c) any = any || ((a.flip_cells[c >> 5] >> (c & 31u)) & 1u) != 0;
      if (!any) return;
    }
  }
  for (uint32_t e = threadIdx.x; e < span; e += 256) {
    const uint32_t i = lo + e;
    s_pos[e] = a.by_key[i];
    s_tk[e] = (uint32_t)a.stag[i] | ((uint32_t)a.sorted_keys[i] << 16);
    s_fb[e] = a.fbits[i];
  }
  __syncthreads();
  {
    uint32_t f[5], local = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint32_t e = threadIdx.x * 5 + j;
      f[j] = e < span ? (s_fb[e] & kSlotStored) : 0u;
      local += f[j];
    }
    uint32_t x = local;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sum[w] = x;
    __syncthreads();
    uint32_t g = x - local;
    for (int i = 0; i < w; ++i) g += wave_sum[i];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint32_t e = threadIdx.x * 5 + j;
      if (e < span) {
        s_rank[e] = (uint16_t)g;
        // (a masked H5 entry goes in with the masked position the reference's ring holds: `p - q <= max_backward` fails for
        // it below, which ends the lookback there like the reference's bucket walk -- br_collect_row does the same)
        if (f[j]) c_ent[g] = make_uint2(((s_fb[e] & kSlotMasked) ? (s_pos[e] & a.ring_mask) : s_pos[e]) | ((s_fb[e] & kSlotWrap) ? 0x80000000u : 0u), s_tk[e]);
      }
      g += f[j];
    }
  }
  __syncthreads();
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  SlotsInMemory sl{a.by_key, a.fbits, a.stag, a.smask, a.gprev};
  // validation passes look only at the slots that can have a candidate at all (Lz77Buffers::pot): the others hold an empty
  // row and keep it whatever the flags do
  const bool use_pot = a.validate && a.pot != nullptr && a.pot_state[0] != 0;
  const bool use_cells = a.validate && a.flip_cells != nullptr && a.flip_cells[a.cell_words] == 0;
  for (uint32_t r = 0; r < kRowTile / 256; ++r) {
    const uint32_t e = (base - lo) + w * 256 + r * 64 + lane;
    bool valid = lo + e < hi;
    if (use_pot) {
      const unsigned long long pw = a.pot[(base + w * 256 + r * 64) >> 6];  // (wave-uniform: the tile starts at a multiple of 64)
      if (pw == 0) continue;
      valid = valid && ((pw >> lane) & 1ull) != 0;
    }
    if (use_cells && valid) {
      // no flag of this key flipped within reach of the position (Lz77Buffers::flip_cells): its row is what it was
      const uint32_t p = s_pos[e];
      const uint32_t first = (p > a.max_backward_limit ? p - a.max_backward_limit : 0u) >> a.cell_shift, last = p >> a.cell_shift;
      const uint32_t key = s_tk[e] >> 16;
      const uint32_t cell0 = key * a.cells_per_key;
      bool any = a.key_last[key] - a.key_first[key] >= 65536u;  // (its ring counter can wrap: wrap marks move with every stored slot in front)
      for (uint32_t c = cell0 + first; c <= cell0 + last; ++c) any = any || ((a.flip_cells[c >> 5] >> (c & 31u)) & 1u) != 0;
      valid = any;
    }
    bool slow = false;
    if (valid) {
      const uint32_t p = s_pos[e], tk = s_tk[e];
      const uint32_t key = tk >> 16, tag = tk & 0xffffu;
      const uint32_t max_backward = p < a.max_backward_limit ? p : a.max_backward_limit;
      const uint32_t oldest = p >= a.reset_pos ? a.reset_vis : 0u;
      // (the row is staged with its groups of four words permuted -- word j of lane l at j ^ swz(l) -- so that the lanes of
      // a wave, which all write "their" j-th word at the same time, spread over the LDS banks)
      uint32_t* out = &rowbuf[w][lane * kRowEntries];
      const uint32_t swz = ((lane >> 2) & 3u) << 2;
      const uint32_t k0 = s_rank[e];
      uint32_t n = 0;
      const uint32_t depth = (s_fb[e] & kSlotWrap) ? 0u : a.depth;
      // the candidates are the `depth` compact entries in front of the slot: fetched together, judged in order
      uint2 ent[kRowEntries];
#pragma unroll
      for (uint32_t j = 0; j < kRowEntries; ++j) ent[j] = c_ent[k0 > j ? k0 - 1 - j : 0u];
      bool live = true;
#pragma unroll
      for (uint32_t j = 0; j < kRowEntries; ++j) {
        const bool wanted = live && j < depth;
        // out of staged slots: slots of this key in front of the span need the slow walk through memory
        if (wanted && k0 <= j) slow = lo > 0 && (s_tk[0] >> 16) == key;
        const uint32_t q = ent[j].x & 0x7fffffffu;
        const bool ok = wanted && k0 > j && (ent[j].y >> 16) == key && p - q <= max_backward && q >= oldest;
        if (ok && (ent[j].y & 0xffffu) == tag) {
          out[n ^ swz] = q;
          n++;
        }
        live = ok && (ent[j].x >> 31) == 0;
      }
      if (slow) {
        br_collect_row(sl, a.max_backward_limit, lo + e, a.key_first[key], a.depth, out, a.reset_pos, a.reset_vis);
        uint32_t tmp[kRowEntries];
#pragma unroll
        for (uint32_t j = 0; j < kRowEntries; ++j) tmp[j] = out[j];
#pragma unroll
        for (uint32_t j = 0; j < kRowEntries; ++j) out[j ^ swz] = tmp[j];
      } else {
        for (; n < kRowEntries; ++n) out[n ^ swz] = kRowEnd;
      }
    }
    const unsigned long long rows_ready = __ballot(valid);
    // (rowbuf[w] belongs to this wave alone and the LDS serves a wave in order: no workgroup barrier, which would also
    // wait for the row stores of the previous pass to land)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the 64 rows of this wave go out as 256 pieces of 16 bytes: four lanes write one 64-byte line
    for (uint32_t it = 0; it < 4; ++it) {
      const uint32_t idx = it * 64 + lane;
      const uint32_t row = idx >> 2, part = idx & 3u;
      const bool go = ((rows_ready >> row) & 1ull) != 0;
      bool diff = false;
      uint32_t p2 = 0;
      if (go) {
        p2 = s_pos[(base - lo) + w * 256 + r * 64 + row];
        const u32x4 v = ((const u32x4*)rowbuf[w])[row * 4 + (part ^ ((row >> 2) & 3u))];
        u32x4* dst = (u32x4*)(a.rows + (size_t)p2 * kRowEntries) + part;
        if (a.validate) {
          const u32x4 old = *dst;
          diff = old.x != v.x || old.y != v.y || old.z != v.z || old.w != v.w;
          if (diff) *dst = v;
        } else {
          *dst = v;
        }
      }
      if (a.validate) {
        const unsigned long long m = __ballot(diff);
        if (go && part == 0 && ((m >> (lane & ~3u)) & 0xfull) != 0) row_changed(a, p2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__global__ __launch_bounds__(256) void k_fbits_tile_sums(const uint8_t* __restrict__ fbits, uint32_t n, const uint8_t* __restrict__ big_tile,
                                                          uint32_t* __restrict__ tile_sums, const uint32_t* __restrict__ skip_unless) {
  if (skip_unless && *skip_unless == 0) return;  // (round update without a flip in a key that can wrap)
  __shared__ uint32_t wave_sum[4];
  const uint32_t base = blockIdx.x * kRowTile + threadIdx.x * 4;
  uint32_t local = 0;
  if (base < n) {
    uint32_t v = *(const uint32_t*)(fbits + base) & 0x01010101u;  // (the array is padded by 64 bytes)
    for (uint32_t j = 0; j < 4; ++j)
      if (base + j >= n) v &= ~(0xffu << (8 * j));
    local = (v * 0x01010101u) >> 24;
  }
  const uint32_t total = block_sum_256(local, wave_sum);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
  (void)big_tile;
}

__global__ __launch_bounds__(256) void k_row_key_bases(const uint8_t* __restrict__ fbits, const uint32_t* __restrict__ tile_offsets,
                                                        const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_last,
                                                        uint32_t* __restrict__ key_base, uint32_t all_keys,
                                                        const uint32_t* __restrict__ skip_unless) {
  if (skip_unless && *skip_unless == 0) return;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 65536) return;
  const uint32_t i0 = key_first[k];
  if (key_last[k] <= i0 || (!all_keys && key_last[k] - i0 < 65536u)) return;  // only keys whose counter can wrap
  const uint32_t tile = i0 / kRowTile;
  uint32_t g = tile_offsets[tile];
  for (uint32_t i = tile * kRowTile; i < i0; ++i) g += fbits[i] & kSlotStored;
  key_base[k] = g;
}

// tiles that hold slots of a key with >= 65 536 slots
__global__ __launch_bounds__(256) void k_flag_big_tiles(const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_last,
                                                         uint8_t* __restrict__ big_tile, uint32_t all_keys) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 65536) return;
  const uint32_t i0 = key_first[k], i1 = key_last[k];
  if (i1 <= i0 || (!all_keys && i1 - i0 < 65536u)) return;
  for (uint32_t t = i0 / kRowTile; t <= (i1 - 1) / kRowTile; ++t) big_tile[t] = 1;
}

// Wrap marks (kSlotWrap) of the keys whose counter can wrap, from the exact count of stored slots in front of every slot.  append: every
// slot whose mark changes is added to the list of changed slots behind the real ones (its neighbourhood has to be
// rebuilt like that of a flipped slot).
__global__ __launch_bounds__(256) void k_mark_wraps(uint8_t* __restrict__ fbits, uint32_t n, const uint8_t* __restrict__ big_tile,
                                                     const uint32_t* __restrict__ tile_offsets, const uint16_t* __restrict__ sorted_keys,
                                                     const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_last,
                                                     const uint32_t* __restrict__ key_base, uint32_t append,
                                                     uint32_t* __restrict__ changed_slot, const uint32_t* __restrict__ changed_count, uint32_t cap,
                                                     uint32_t* __restrict__ ctl, const uint32_t* __restrict__ count_base,
                                                     const uint32_t* __restrict__ by_key, uint32_t reset_pos,
                                                     const uint32_t* __restrict__ reset_counts) {
  if (append && ctl[kCtlWrapKeyFlip] == 0) return;
  if (!big_tile[blockIdx.x]) return;
  __shared__ uint32_t wave_sum[4];
  const uint32_t base = blockIdx.x * kRowTile + threadIdx.x * 4;
  uint32_t f[4], local = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = base + j < n ? fbits[base + j] : 0u;
    local += f[j] & kSlotStored;
  }
  uint32_t x = local;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_sum[w] = x;
  __syncthreads();
  uint32_t g = tile_offsets[blockIdx.x] + x - local;
  for (int i = 0; i < w; ++i) g += wave_sum[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t i = base + j;
    if (i < n) {
      const uint32_t key = sorted_keys[i];
      bool want = false;
      if (count_base != nullptr || reset_pos != 0 || key_last[key] - key_first[key] >= 65536u) {
        // insertions into the key's ring in front of this slot: since the start of the stream, or since the hasher reset
        uint32_t count = g - key_base[key];
        if (reset_pos != 0 && by_key[i] >= reset_pos) count -= reset_counts[key];
        else count += count_base ? count_base[key] : 0u;
        want = count != 0 && (count & 0xffffu) == 0;
      }
      if (want != ((f[j] & kSlotWrap) != 0)) {
        fbits[i] = (uint8_t)(f[j] ^ kSlotWrap);
        if (append) {
          const uint32_t real = *changed_count;
          const uint32_t at = real + atomicAdd(&ctl[kCtlVirtual], 1u);
          if (real <= cap && at < cap) changed_slot[at] = i; else atomicMax(&ctl[kCtlNeedFull], 1u);
        }
      }
    }
    g += f[j] & kSlotStored;
  }
}

// stored bit, and kSlotMasked for kFlagMasked (set by the chains only where masked H5 entries are modelled, Lz77Params::masked_from)
__device__ __forceinline__ uint32_t slot_bits_of_flag(uint32_t flag) { return (flag & 1u) | ((flag & kFlagMasked) ? kSlotMasked : 0u); }

// stored bits from scratch (list overflow): a random one-byte gather per slot; the wrap marks are set afterwards
__global__ __launch_bounds__(256) void k_regather_fbits(const uint32_t* __restrict__ by_key, const uint8_t* __restrict__ flags, uint32_t n,
                                                         uint8_t* __restrict__ fbits, const uint32_t* __restrict__ ctl) {
  if (ctl[kCtlNeedFull] != 2) return;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    fbits[i] = (uint8_t)(slot_bits_of_flag(flags[by_key[i]]) | (fbits[i] & kSlotWrap));
}

// One thread per changed position: find its slot (binary search in its key's slot range), flip the stored bit there.
__global__ __launch_bounds__(256) void k_apply_flips(const uint32_t* __restrict__ changed_pos, const uint32_t* __restrict__ changed_count,
                                                      uint32_t cap, const uint16_t* __restrict__ keys, const uint32_t* __restrict__ key_first,
                                                      const uint32_t* __restrict__ key_last, const uint32_t* __restrict__ by_key,
                                                      const uint8_t* __restrict__ flags_new, uint8_t* __restrict__ fbits,
                                                      uint32_t* __restrict__ changed_slot, uint32_t* __restrict__ ctl, uint32_t total_slots,
                                                      uint32_t all_keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = *changed_count;
  if (n > cap) {
    if (i == 0) {
      atomicMax(&ctl[kCtlNeedFull], 2u);
      atomicMax(&ctl[kCtlWrapKeyFlip], 1u);  // (the list is incomplete: anything may have flipped)
    }
    return;
  }
  if (i >= n) return;
  // many changes: one pass over all rows is cheaper than a walk per change.  Measured (profiles/r03_c5_xorshift_1GiB_q5.json): a
  // walked slot costs 0.45 ns, a slot of the full pass 0.017 ns -- 27 x less; k_update_rows gives up after n / 64 walked
  // slots (it used to walk n / 16 of them, 30 ms at 1 GiB, before the full pass ran anyway), and every change walks at
  // least 64.
  if (i == 0 && n > total_slots / 4096) atomicMax(&ctl[kCtlNeedFull], 1u);
  const uint32_t p = changed_pos[i];
  const uint32_t key = keys[p];
  uint32_t lo = key_first[key], hi = key_last[key];
  if (all_keys || hi - lo >= 65536u) atomicMax(&ctl[kCtlWrapKeyFlip], 1u);
  while (lo + 1 < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (by_key[mid] <= p) lo = mid; else hi = mid;
  }
  fbits[lo] = (uint8_t)(slot_bits_of_flag(flags_new[p]) | kSlotChanged | (fbits[lo] & kSlotWrap));
  changed_slot[i] = lo;
}

// One wavefront per changed slot s (flipped, or wrap mark moved): rebuild the rows of the slots behind it that can
// have s in their lookback, i.e. up to the `depth`-th slot behind s that is stored and did not change itself in this
// round (such slots are stored in the old AND in the new flag state, so `depth` of them push s out of the ring in both).
__global__ __launch_bounds__(64) void k_update_rows(RowArgs a, const uint32_t* __restrict__ changed_slot,
                                                     const uint32_t* __restrict__ changed_count, const uint16_t* __restrict__ keys) {
  if (a.ctl[kCtlNeedFull] != 0) return;
  const uint32_t n = *changed_count + a.ctl[kCtlVirtual];
  SlotsInMemory sl{a.by_key, a.fbits, a.stag, a.smask, a.gprev};
  for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
    // A change in front of a long stretch of unstored slots reaches every row of the stretch, and several such changes
    // walk the same stretch again and again: when the walks add up to a good part of all rows, the full rebuild behind
    // this kernel is the cheaper way (what was updated here has been checked and marked already).
    if (a.walk_counter[kCtlWalked] > a.n / 64 || *(volatile const uint32_t*)&a.ctl[kCtlNeedFull] != 0) {
      if (threadIdx.x == 0) atomicMax(&a.walk_counter[kCtlNeedFull], 1u);
      return;
    }
    const uint32_t s = changed_slot[item];
    const uint32_t key = keys[a.by_key[s]];
    const uint32_t kf = a.key_first[key], kl = a.key_last[key];
    uint32_t stable = 0;
    const bool use_pot = a.pot != nullptr && a.pot_state[0] != 0;
    const uint32_t max_walk = use_pot ? 32u * kMaxRowWalk : kMaxRowWalk;  // (a walked slot without a row to rebuild costs one ballot)
    // (slot s itself is rebuilt as well: its own row changes when its wrap mark does)
    for (uint32_t base = s; base < kl && stable < a.depth; base += 64) {
      if (base - s >= max_walk) {
        // a stretch of unstored slots (e.g. an extended copy through zero fill): every row of it changes, and every
        // change in front of it would walk it again -- leave it to the full rebuild
        if (threadIdx.x == 0) atomicMax(&a.walk_counter[kCtlNeedFull], 1u);
        return;
      }
      const uint32_t i = base + threadIdx.x;
      const bool in = i < kl;
      const bool st = in && i != s && (a.fbits[i] & (kSlotStored | kSlotChanged)) == kSlotStored;
      const unsigned long long m = __ballot(st);
      const uint32_t before = (uint32_t)__popcll(m & ((1ull << threadIdx.x) - 1ull));
      bool build = in && stable + before < a.depth;
      if (use_pot && build) build = ((a.pot[i >> 6] >> (i & 63u)) & 1ull) != 0;  // (no candidate possible: the row is empty and stays so)
      if (build) {
        if (br_build_row(sl, a.rows, a.max_backward_limit, i, kf, a.depth, true, a.reset_pos, a.reset_vis)) row_changed(a, a.by_key[i]);
      }
      stable += (uint32_t)__popcll(m);
      const uint32_t cost = use_pot ? (__ballot(build) != 0 ? 64u : 4u) : 64u;
      if (threadIdx.x == 0) atomicAdd(&a.walk_counter[kCtlWalked], cost);
    }
  }
}

__global__ __launch_bounds__(256) void k_clear_flip_marks(const uint32_t* __restrict__ changed_slot, const uint32_t* __restrict__ changed_count,
                                                           uint32_t cap, uint8_t* __restrict__ fbits) {
  const uint32_t n = *changed_count;
  if (n > cap) return;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) fbits[changed_slot[i]] &= (uint8_t)~kSlotChanged;
}

static RowArgs row_args(const Lz77Params& P, const Lz77Buffers& B, int which, bool validate, const SegGeometry* geo, uint8_t* dirty_dev) {
  RowArgs a{};
  a.by_key = B.by_key;
  a.sorted_keys = B.sorted_keys;
  a.stag = B.stag;
  a.fbits = B.fbits;
  a.key_first = B.key_first;
  a.key_last = B.key_last;
  a.rows = B.rows;
  a.flags = B.flags[which];
  a.n = P.total_bytes;
  a.depth = 1u << P.block_bits;
  a.max_backward_limit = P.max_backward_limit;
  a.reset_pos = P.reset_pos;
  a.reset_vis = P.reset_vis;
  a.ring_mask = P.ring_mask;
  a.validate = validate ? 1 : 0;
  if (geo) a.geo = *geo;
  a.dirty = dirty_dev;
  a.rows_lo = validate ? B.rows_changed_lo : nullptr;
  a.rows_hi = validate ? B.rows_changed_hi : nullptr;
  a.ctl = B.row_ctl;
  a.walk_counter = B.row_ctl;
  a.smask = B.smask;
  a.gprev = B.gprev;
  a.pot = B.pot;
  a.pot_state = B.pot_state;
  // (wrap marks move with the count of ALL stored slots in front of a slot: no cell filter where ring counters can wrap --
  // every key when the counters run on from the stream in front or start over inside the text, else the keys with >= 65 536 slots)
  a.flip_cells = (validate && B.count_base == nullptr && P.reset_pos == 0) ? B.flip_cells : nullptr;
  a.cell_shift = B.cell_shift;
  a.cells_per_key = B.cells_per_key;
  a.cell_words = (uint32_t)(((size_t)65536 * B.cells_per_key) / 32);
  return a;
}

// ---- the potential mask (Lz77Buffers::pot): one bit per slot, set when some slot of the same key in front of it, at most
// max_backward_limit bytes back, carries the same tag -- the only slots whose candidate row can ever hold an entry
// (br_collect_row keeps tag-equal predecessors only).  Conservative where the scan is cut short.  Runs once per call, and
// only when a full validation pass is due (ctl[kCtlNeedFull]) and the mask is not there yet.
// One workgroup per tile of kPotTile slots with kPotHalo slots in front of it, staged in LDS.  A slot gets the bit when an
// earlier slot of the staged span carries the same (key, tag) at most max_backward_limit bytes back -- the first occurrence of
// every value in the span comes out of two hash tables in LDS (entries carry the number of the tile they were written for,
// so the tables are cleared once per workgroup, not once per tile); where a value collides in both, or its first occurrence
// lies outside the window, the wave scans the slots in front of it 64 at a time -- or when its key's slots reach back beyond
// the staged span inside the window (dense keys, i.e. text: nearly every slot, but such input never asks for the mask).
// The slots with the bit are also LISTED (pot_list; pot_state[1] = their number, pot_state[2] = the list overflowed): the
// validation pass of sparse input is a pass over that list (k_validate_listed_rows), not over all slots.
#if !defined(BR_POT_TABLE)
#define BR_POT_TABLE 8192
#define BR_POT_SHIFT 19
#endif
static constexpr uint32_t kPotTile = 1024, kPotHalo = 512, kPotSpan = kPotTile + kPotHalo, kPotTable = BR_POT_TABLE, kPotShift = BR_POT_SHIFT;
__global__ __launch_bounds__(256) void k_row_potential(const uint32_t* __restrict__ by_key, const uint16_t* __restrict__ sorted_keys,
                                                        const uint16_t* __restrict__ stag, uint32_t n, uint32_t max_backward_limit,
                                                        const uint32_t* __restrict__ ctl, uint32_t* __restrict__ pot_state,
                                                        unsigned long long* __restrict__ pot, uint32_t* __restrict__ pot_list, uint32_t list_cap) {
  if (ctl[kCtlNeedFull] == 0 || pot_state[0] != 0) return;
  __shared__ uint32_t s_val[kPotSpan];  // tag | key << 16
  __shared__ uint32_t s_pos[kPotSpan];
  // two tables under two hash functions; an entry = generation << 16 | (0xffff - first span entry whose value hashes here), atomicMax
  __shared__ uint32_t tab[kPotTable], tab2[kPotTable];
  __shared__ uint32_t s_count, s_base;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t h = threadIdx.x; h < kPotTable; h += 256) tab[h] = tab2[h] = 0;
  uint32_t gen = 0;
  for (uint32_t base = blockIdx.x * kPotTile; base < n; base += gridDim.x * kPotTile) {
    ++gen;  // (at most n / kPotTile / gridDim.x + 1 tiles per workgroup: far below 65 536)
    const uint32_t lo = base >= kPotHalo ? base - kPotHalo : 0u;
    const uint32_t hi = min(base + kPotTile, n);
    const uint32_t span = hi - lo;
    __syncthreads();  // (the previous tile is done with the arrays)
    for (uint32_t e = threadIdx.x; e < span; e += 256) {
      s_val[e] = (uint32_t)stag[lo + e] | ((uint32_t)sorted_keys[lo + e] << 16);
      s_pos[e] = by_key[lo + e];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < span; e += 256) {
      // (an entry that repeats the value in front of it is not a first occurrence: runs of one value -- zero fill -- would
      // all hit one table word)
      if (e != 0 && s_val[e - 1] == s_val[e]) continue;
      atomicMax(&tab[(s_val[e] * 0x9E3779B1u) >> kPotShift], (gen << 16) | (0xffffu - e));
      atomicMax(&tab2[(s_val[e] * 0x85EBCA6Bu) >> kPotShift], (gen << 16) | (0xffffu - e));
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    unsigned long long masks[kPotTile / 256];
    for (uint32_t r = 0; r < kPotTile / 256; ++r) {
      const uint32_t e = (base - lo) + r * 256 + threadIdx.x;
      const bool in = lo + e < hi;
      bool bit = false, scan = false;
      uint32_t val = 0, p = 0;
      if (in) {
        val = s_val[e];
        p = s_pos[e];
        uint32_t f = 0xffffu - (tab[(val * 0x9E3779B1u) >> kPotShift] & 0xffffu);  // (this tile's: e or the head of its run was entered)
        if (f >= span || s_val[f] != val) f = 0xffffu - (tab2[(val * 0x85EBCA6Bu) >> kPotShift] & 0xffffu);
        if (f >= span) f = e;  // (cannot happen: both words hold an entry of this tile)
        scan = true;
        if (s_val[f] == val) {  // f is the first occurrence of this value in the span
          if (f == e) scan = false;                                  // nothing in front of it
          else if (p - s_pos[f] <= max_backward_limit) bit = true;   // (f < e) and within reach
        }
        scan = scan && !bit;
      }
      // the slow way, one slot at a time with the whole wave: the slots in front of it, 64 per step, nearest first
      unsigned long long todo = __ballot(scan);
      if (__popcll(todo) > 4) {  // dense values (text): not worth looking closer, such slots simply count as possible
        if (scan) bit = true;
        todo = 0;
      }
      while (todo != 0) {
        const uint32_t l = (uint32_t)__ffsll((long long)todo) - 1u;
        todo &= todo - 1ull;
        const uint32_t ee = (uint32_t)__shfl((int)e, (int)l, 64), vv = (uint32_t)__shfl((int)val, (int)l, 64), pp = (uint32_t)__shfl((int)p, (int)l, 64);
        bool found = false;
        for (uint32_t top = ee; top > 0;) {
          const bool have = top > lane;
          const uint32_t j = have ? top - 1u - lane : 0u;
          const uint32_t v = s_val[j];
          const bool reach = have && (v >> 16) == (vv >> 16) && pp - s_pos[j] <= max_backward_limit;
          const unsigned long long reach_m = __ballot(reach), hit_m = __ballot(reach && v == vv);
          // (slots of a key are in position order: the ones within reach are the nearest, a prefix of the lanes)
          if (hit_m != 0) {
            found = true;
            break;
          }
          if (reach_m != ~0ull) break;  // the walk ends inside this step
          top = top > 64u ? top - 64u : 0u;
        }
        if (lane == l) bit = found;
      }
      // the key's slots go on in front of the span, still inside the window: anything may be there
      if (in && !bit && lo > 0 && (s_val[0] >> 16) == (val >> 16) && p - s_pos[0] <= max_backward_limit) bit = true;
      const unsigned long long m = __ballot(bit);
      const uint32_t i = lo + e;
      if (lane == 0 && (i >> 6) <= ((n - 1) >> 6)) pot[i >> 6] = m;
      masks[r] = m;
    }
    // the list: one reservation per tile (and none once the list has overflowed: dense input)
    uint32_t mine = 0, wave_at = 0;
#pragma unroll
    for (uint32_t r = 0; r < kPotTile / 256; ++r) mine += (uint32_t)__popcll(masks[r]);
    if (lane == 0 && mine != 0) wave_at = atomicAdd(&s_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t total = s_count;
      uint32_t at = 0xffffffffu;
      if (total != 0) {
        if (*(volatile uint32_t*)&pot_state[2] == 0) {
          at = atomicAdd(&pot_state[1], total);
          if (at + total > list_cap) {
            pot_state[2] = 1;
            at = 0xffffffffu;
          }
        }
      }
      s_base = at;
    }
    __syncthreads();
    if (s_base != 0xffffffffu) {
      uint32_t at = s_base + (uint32_t)__shfl((int)wave_at, 0, 64);
#pragma unroll
      for (uint32_t r = 0; r < kPotTile / 256; ++r) {
        const unsigned long long m = masks[r];
        if ((m >> lane) & 1ull) pot_list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = base + r * 256 + threadIdx.x;
        at += (uint32_t)__popcll(m);
      }
    }
  }
}
// The validation pass over the listed slots: every row that can hold a candidate is built afresh from the slots in memory and
// compared with the row in memory (br_build_row); the chains that searched a changed row are marked.
__global__ __launch_bounds__(256) void k_validate_listed_rows(RowArgs a, const uint32_t* __restrict__ pot_list) {
  if (a.ctl[kCtlNeedFull] == 0 || a.pot_state[0] == 0 || a.pot_state[2] != 0) return;
  const uint32_t count = a.pot_state[1];
  SlotsInMemory sl{a.by_key, a.fbits, a.stag, a.smask, a.gprev};
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
    const uint32_t i = pot_list[t];
    if (a.flip_cells != nullptr && a.flip_cells[a.cell_words] == 0) {
      // no flag of this key flipped within reach of the position: its row is what it was
      const uint32_t p = a.by_key[i];
      const uint32_t first = (p > a.max_backward_limit ? p - a.max_backward_limit : 0u) >> a.cell_shift, last = p >> a.cell_shift;
      const uint32_t key = a.sorted_keys[i];
      const uint32_t cell0 = key * a.cells_per_key;
      bool any = a.key_last[key] - a.key_first[key] >= 65536u;  // (its ring counter can wrap: wrap marks move with every stored slot in front)
      for (uint32_t c = cell0 + first; c <= cell0 + last; ++c) any = any || ((a.flip_cells[c >> 5] >> (c & 31u)) & 1u) != 0;
      if (!any) continue;
    }
    const uint32_t kf = a.key_first[a.sorted_keys[i]];
    if (br_build_row(sl, a.rows, a.max_backward_limit, i, kf, a.depth, true, a.reset_pos, a.reset_vis)) row_changed(a, a.by_key[i]);
  }
}
__global__ void k_row_potential_ready(const uint32_t* __restrict__ ctl, uint32_t* __restrict__ pot_state) {
  if (ctl[kCtlNeedFull] != 0 && threadIdx.x == 0 && blockIdx.x == 0) pot_state[0] = 1;
}

// ---- stored-bit masks and skip pointers (SlotsInMemory::prev_stored) from the per-slot bytes ----
__global__ __launch_bounds__(256) void k_slot_masks(const uint8_t* __restrict__ fbits, uint32_t n, unsigned long long* __restrict__ smask,
                                                     uint32_t* __restrict__ glast) {
  const uint32_t groups = (n + 63) / 64;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
    unsigned long long m = 0;
    const uint4* src = (const uint4*)(fbits + (size_t)g * 64);  // (the array is padded by 64 bytes)
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
      const uint4 v = src[q];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        // gather bit 0 of the four bytes of a word into four adjacent bits
        const uint32_t b = w[k] & 0x01010101u;
        const uint32_t nib = (b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xfu;
        m |= (unsigned long long)nib << (q * 16 + k * 4);
      }
    }
    if ((size_t)g * 64 + 64 > n) m &= (1ull << (n - g * 64)) - 1ull;
    smask[g] = m;
    glast[g] = m ? g * 64 + 64 - (uint32_t)__builtin_clzll(m) : 0u;
  }
}

// exclusive prefix MAXIMUM of a uint32 array, in place (same shape as exclusive_scan_u32)
__global__ __launch_bounds__(256) void k_maxscan_tiles(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ tile_max) {
  __shared__ uint32_t wave_max[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (base + j < n) ? data[base + j] : 0;
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
  uint32_t before = 0;  // maximum over everything in front of this thread's four elements
  for (int i = 0; i < w; ++i) before = max(before, wave_max[i]);
  const uint32_t prev_lane = __shfl_up(x, 1, 64);
  if (lane > 0) before = max(before, prev_lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) data[base + j] = before;
    before = max(before, v[j]);
  }
  if (threadIdx.x == 255 && tile_max) tile_max[blockIdx.x] = max(max(wave_max[0], wave_max[1]), max(wave_max[2], x));
}
__global__ __launch_bounds__(256) void k_maxscan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ tile_before) {
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
  const uint32_t add = tile_before[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) data[base + j] = max(data[base + j], add);
}
static void exclusive_maxscan_u32(uint32_t* data, uint32_t n, uint32_t* scratch) {
  if (n == 0) return;
  const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_maxscan_tiles, dim3(tiles), dim3(256), 0, BR_STREAM, data, n, tiles > 1 ? scratch : (uint32_t*)nullptr);
  if (tiles > 1) {
    exclusive_maxscan_u32(scratch, tiles, scratch + tiles);
    hipLaunchKernelGGL(k_maxscan_add, dim3(tiles), dim3(256), 0, BR_STREAM, data, n, scratch);
  }
}

// smask / gprev from the current per-slot bytes (after every change of stored bits)
static void launch_slot_masks(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  const uint32_t groups = (n + 63) / 64;
  uint32_t blocks = (groups + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_slot_masks, dim3(blocks), dim3(256), 0, BR_STREAM, B.fbits, n, B.smask, B.gprev);
  // (scratch: behind the tile sums of the wrap-mark pass in sort_tmp)
  uint32_t* scratch = (uint32_t*)B.sort_tmp + ((size_t)n / kScanTile + 4096);
  exclusive_maxscan_u32(B.gprev, groups, scratch);
}

// wrap marks from the current stored bits (keys with >= 65 536 slots only)
static void launch_wrap_marks(const Lz77Params& P, const Lz77Buffers& B, bool append) {
  const uint32_t n = P.total_bytes;
  const uint32_t tiles = (n + kRowTile - 1) / kRowTile;
  uint32_t* tile_sums = (uint32_t*)B.sort_tmp;
  uint32_t* scratch = tile_sums + tiles + 64;
  const uint32_t* skip_unless = append ? B.row_ctl + kCtlWrapKeyFlip : (const uint32_t*)nullptr;
  hipLaunchKernelGGL(k_fbits_tile_sums, dim3(tiles), dim3(256), 0, BR_STREAM, B.fbits, n, B.big_tile, tile_sums, skip_unless);
  exclusive_scan_u32(tile_sums, tiles, scratch);
  const bool all_keys = B.count_base != nullptr || P.reset_pos != 0;
  hipLaunchKernelGGL(k_row_key_bases, dim3(256), dim3(256), 0, BR_STREAM, B.fbits, tile_sums, B.key_first, B.key_last, B.key_base,
                     all_keys ? 1u : 0u, skip_unless);
  hipLaunchKernelGGL(k_mark_wraps, dim3(tiles), dim3(256), 0, BR_STREAM, B.fbits, n, B.big_tile, tile_sums, B.sorted_keys, B.key_first, B.key_last,
                     B.key_base, append ? 1u : 0u, B.changed_slot, B.changed_count, B.changed_cap, B.row_ctl, B.count_base, B.by_key, P.reset_pos,
                     B.reset_counts);
}

void lz77_rows_init(const Lz77Params& P, const Lz77Buffers& B, int which, const RankInitialHint* initial, bool has_big_keys) {
  const uint32_t n = P.total_bytes;
  if (n == 0) return;
  if ((1u << P.block_bits) > kRowEntries) throw std::runtime_error("candidate rows: ring deeper than a row");
  InitialFlagGeometry ig{};
  if (initial) {
    ig.enabled = 1;
    ig.first_block_start = initial->first_block_start;
    ig.prefix_bytes = P.prefix_bytes;
    ig.block_bytes = initial->block_bytes;
    ig.total_bytes = n;
    ig.prefix_stored_end = (initial->prefix_is_dictionary && P.prefix_bytes > P.htl - 1) ? P.prefix_bytes - (P.htl - 1) : 0;
  }
  const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
  uint32_t* tile_sums = (uint32_t*)B.sort_tmp;
  hipLaunchKernelGGL(k_rank_gather, dim3(tiles), dim3(256), 0, BR_STREAM, B.by_key, B.flags[which], n, B.fbits, tile_sums, ig,
                     P.masked_from != kNeverMasked ? 1u : 0u);
  dev_memset(B.row_ctl, 0, kCtlWords * 4);
  if (P.reset_pos) lz77_key_counts(P, B, which, P.reset_vis, B.reset_counts, false);
  if (has_big_keys) {
    dev_memset(B.big_tile, 0, tiles + 64);
    hipLaunchKernelGGL(k_flag_big_tiles, dim3(256), dim3(256), 0, BR_STREAM, B.key_first, B.key_last, B.big_tile,
                       (B.count_base || P.reset_pos) ? 1u : 0u);
    launch_wrap_marks(P, B, false);
  }
  launch_slot_masks(P, B);
  RowArgs a = row_args(P, B, which, false, nullptr, nullptr);
  hipLaunchKernelGGL(k_build_rows, dim3((n + kRowTile - 1) / kRowTile), dim3(256), 0, BR_STREAM, a);
  HIP_CHECK(hipGetLastError());
}

void lz77_rows_update(const Lz77Params& P, const Lz77Buffers& B, int prev, int next, const SegGeometry& geo, uint8_t* dirty_dev,
                      bool has_big_keys) {
  const uint32_t n = P.total_bytes;
  if (n == 0) return;
  const uint32_t cap = B.changed_cap;
  dev_memset(B.row_ctl, 0, kCtlWords * 4);
  // (B.changed_keys holds POSITIONS here, B.changed_count their number -- see lz77_diff_flags)
  const uint32_t flip_blocks = (cap + 255) / 256;
  hipLaunchKernelGGL(k_apply_flips, dim3(flip_blocks), dim3(256), 0, BR_STREAM, B.changed_keys, B.changed_count, cap, B.keys, B.key_first,
                     B.key_last, B.by_key, B.flags[next], B.fbits, B.changed_slot, B.row_ctl, n,
                     (B.count_base != nullptr || P.reset_pos != 0) ? 1u : 0u);
  uint32_t gather_blocks = (n + 255) / 256;
  if (gather_blocks > 8192) gather_blocks = 8192;
  hipLaunchKernelGGL(k_regather_fbits, dim3(gather_blocks), dim3(256), 0, BR_STREAM, B.by_key, B.flags[next], n, B.fbits, B.row_ctl);
  if (P.reset_pos) lz77_key_counts(P, B, next, P.reset_vis, B.reset_counts, false);
  if (has_big_keys) launch_wrap_marks(P, B, true);
  launch_slot_masks(P, B);
  RowArgs a = row_args(P, B, next, true, &geo, dirty_dev);
  hipLaunchKernelGGL(k_update_rows, dim3(cap < 65536u ? cap : 65536u), dim3(64), 0, BR_STREAM, a, B.changed_slot, B.changed_count, B.keys);
  a.conditional = 1;
  if (B.pot) {
    uint32_t pot_blocks = (n + kPotTile - 1) / kPotTile;
    if (pot_blocks > 8192u) pot_blocks = 8192u;
    hipLaunchKernelGGL(k_row_potential, dim3(pot_blocks), dim3(256), 0, BR_STREAM, B.by_key, B.sorted_keys, B.stag, n, P.max_backward_limit,
                       B.row_ctl, B.pot_state, B.pot, B.pot_list, B.pot_list_cap);
    hipLaunchKernelGGL(k_row_potential_ready, dim3(1), dim3(64), 0, BR_STREAM, B.row_ctl, B.pot_state);
    hipLaunchKernelGGL(k_validate_listed_rows, dim3(2048), dim3(256), 0, BR_STREAM, a, B.pot_list);
  }
  hipLaunchKernelGGL(k_build_rows, dim3((n + kRowTile - 1) / kRowTile), dim3(256), 0, BR_STREAM, a);
  hipLaunchKernelGGL(k_clear_flip_marks, dim3(flip_blocks < 1024u ? flip_blocks : 1024u), dim3(256), 0, BR_STREAM, B.changed_slot, B.changed_count,
                     cap, B.fbits);
  HIP_CHECK(hipGetLastError());
  (void)prev;
}

// ------------------------------------------------------------------------------------------ key counts
// one wavefront per key: stored positions of the key in front of text position `upto`
__global__ __launch_bounds__(64) void k_key_counts(const uint32_t* __restrict__ by_key, const uint8_t* __restrict__ flags,
                                                    const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_last,
                                                    const uint32_t* __restrict__ base, uint32_t upto, uint32_t* __restrict__ out) {
  const uint32_t key = blockIdx.x;
  uint32_t local = 0;
  for (uint32_t i = key_first[key] + threadIdx.x; i < key_last[key]; i += 64) {
    const uint32_t p = by_key[i];
    if (p < upto) local += flags[p] & 1u;  // (slots are in position order, but a strided lane cannot stop early for the others)
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if (threadIdx.x == 0) out[key] = local + (base ? base[key] : 0u);
}

void lz77_key_counts(const Lz77Params& P, const Lz77Buffers& B, int which, uint32_t upto, uint32_t* out_dev, bool with_base) {
  (void)P;
  hipLaunchKernelGGL(k_key_counts, dim3(65536), dim3(64), 0, BR_STREAM, B.by_key, B.flags[which], B.key_first, B.key_last,
                     with_base ? B.count_base : (const uint32_t*)nullptr, upto, out_dev);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ run table
// run_end[p] = first position behind p whose byte differs from text[p] (ChainTables::run_end): a maximum scan over the
// positions in REVERSE order (j = n - 1 - p) of "a run ends here".
__global__ __launch_bounds__(256) void k_run_marks(const uint8_t* __restrict__ text, uint32_t n, uint32_t* __restrict__ marks) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t p = n - 1 - j;
    marks[j] = (p + 1 >= n || text[p + 1] != text[p]) ? j + 1 : 0u;
  }
}
__global__ __launch_bounds__(256) void k_run_finish(const uint8_t* __restrict__ text, uint32_t n, uint32_t* __restrict__ data) {
  // data[j] = exclusive prefix maximum of the marks; position p <-> j = n - 1 - p: every thread turns one pair around
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < (n + 1) / 2; p += gridDim.x * blockDim.x) {
    const uint32_t q = n - 1 - p;
    auto end_of = [&](uint32_t pos, uint32_t excl) {
      const uint32_t j = n - 1 - pos;
      const uint32_t own = (pos + 1 >= n || text[pos + 1] != text[pos]) ? j + 1 : 0u;
      const uint32_t m = excl > own ? excl : own;  // 1 + reversed index of the last position of the run
      return n - m + 1;
    };
    const uint32_t xp = data[n - 1 - p], xq = data[n - 1 - q];
    const uint32_t ep = end_of(p, xp), eq = end_of(q, xq);
    data[p] = ep;
    if (q != p) data[q] = eq;
  }
}

void lz77_run_table(const Lz77Params& P, const Lz77Buffers& B) {
  const uint32_t n = P.total_bytes;
  if (n == 0 || !B.run_end) return;
  uint32_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_run_marks, dim3(blocks), dim3(256), 0, BR_STREAM, B.text, n, B.run_end);
  exclusive_maxscan_u32(B.run_end, n, (uint32_t*)B.sort_tmp);
  hipLaunchKernelGGL(k_run_finish, dim3(blocks), dim3(256), 0, BR_STREAM, B.text, n, B.run_end);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ parse
struct ParseTiming {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // the first `used` are recorded; created once, reused from call to call
  size_t used = 0;
  std::vector<uint32_t> counts;  // chains per launch
  uint64_t segments = 0;
  unsigned long long* work_dev = nullptr;  // [4] what the chains did since the last lz77_parse_timing (ChainTables::work)
};
static ParseTiming& parse_timing() {
  static thread_local ParseTiming t;
  return t;
}

#if !defined(BR_PARSE_WAVES)
#define BR_PARSE_WAVES 8
#endif
// (quality 9 walks up to 256 ring entries per search through LDS staging and wants registers more than waves: 6 per SIMD
// measured 8 % faster than 8, which the fixed-layout quality-5 probe prefers)
// kSpec: 0 = every parameter is read from ParseArgs.  4 / 8 = the plain quality-5 configuration with that hash type length
// (H5 / H6 as BrotliEncoderCompress picks them: four cache candidates, 16-deep rings, the quality < 9 spree window and score,
// no custom-dictionary break, no hasher reset, no masked entries): the constants are folded into the code, which is a tenth
// shorter and keeps that many fewer scalars alive across the parse loop.  Same results: plain_q5_config() admits only
// parameter sets that equal the constants.
static bool plain_q5_config(const Lz77Params& P) {
  return P.hasher_kind != 9 && P.ndist == 4 && P.block_bits == 4 && P.spree_window == 64 && P.score_per_byte == 135 && P.dict_break == 0 &&
         P.reset_pos == 0 && P.masked_from == kNeverMasked && (P.htl == 4 || P.htl == 8);
}
// kSplice: the chains of a list launch, which restart from and stop at checkpoints (lz77_chain.h, Checkpoint / Reparse);
// round 0 and the warm-up run without that code (round 0 records the checkpoints).
// kDeep: the 512-deep rings of quality 11 + Q9_5 (ChainScratchT<.., kDeep>): twice the candidates per search of quality 9 and 17
// trips of candidate bookkeeping in registers (120 VGPRs): four waves per SIMD is what fits
template <bool kH9, bool kRows, uint32_t kSpec = 0, bool kSplice = false, bool kDeep = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kDeep ? 4 : (kH9 ? 6 : BR_PARSE_WAVES), kDeep ? 4 : (kH9 ? 6 : BR_PARSE_WAVES)))) void k_parse_segments(ParseArgs a) {
  __shared__ ChainScratchT<kH9, kRows, kDeep> scratch;
  // workgroups are dealt round-robin to the 8 XCDs: give every XCD one contiguous run of segments so that the text
  // window and the rank rows its chains touch stay in that XCD's L2
  uint32_t item = blockIdx.x;
  if (a.per_xcd) item = (blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3);
  if (item >= a.count) return;
  const uint32_t k = a.list ? a.list[item] : a.first_segment + item;
  if constexpr (kSpec != 0) {
    Lz77Params P = a.P;
    P.ndist = 4;
    P.block_bits = 4;
    P.spree_window = 64;
    P.score_per_byte = 135;
    P.dict_break = 0;
    P.reset_pos = 0;
    P.masked_from = kNeverMasked;
    P.htl = kSpec;
    br_parse_chain<kH9, kRows, kSplice>(P, a.T, scratch, a.segments, a.entries, a.exits, k, a.sched, a.max_continuation);
  } else {
    br_parse_chain<kH9, kRows, kSplice>(a.P, a.T, scratch, a.segments, a.entries, a.exits, k, a.sched, a.max_continuation);
  }
}

// Four chains per wavefront (lz77_groups.h): round 0 and the warm-up of the plain quality-5 configuration, where a launch holds
// every segment of a large input.  The parse state of a chain lives in the vector registers of its 16 lanes: four wavefronts per
// SIMD is what the register file holds (128 registers each).
#if !defined(BR_GROUP_WAVES)
#define BR_GROUP_WAVES 3
#endif
template <uint32_t kHtl>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BR_GROUP_WAVES, BR_GROUP_WAVES))) void k_parse_groups(ParseArgs a) {
  __shared__ uint32_t dict_off[32];
  if (threadIdx.x < 32) dict_off[threadIdx.x] = threadIdx.x < 25 ? a.T.dict_offsets_by_length[threadIdx.x] : 0u;
  __builtin_amdgcn_wave_barrier();
  uint32_t item = blockIdx.x;
  if (a.per_xcd) item = (blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3);
  const uint32_t i = item * 4u + (threadIdx.x >> 4);
  const bool have = i < a.count;
  const uint32_t k = a.first_segment + (have ? i : 0u);
  Lz77Params P = a.P;
  P.ndist = 4;
  P.block_bits = 4;
  P.spree_window = 64;
  P.score_per_byte = 135;
  P.dict_break = 0;
  P.reset_pos = 0;
  P.masked_from = kNeverMasked;
  P.htl = kHtl;
  uint32_t walked, searches, commands;
  br_group_parse<kHtl>(P, a.T, a.segments[k], a.entries[k], a.exits[k], have, dict_off, &walked, &searches, &commands);
  if (a.T.work && have && (threadIdx.x & 15u) == 0) {
    atomicAdd(a.T.work + 0, (unsigned long long)walked);
    atomicAdd(a.T.work + 1, (unsigned long long)searches);
    atomicAdd(a.T.work + 2, (unsigned long long)commands);
  }
}

static void launch_parse(const Lz77Params& P, const Lz77Buffers& B, int flags_in, int flags_out, int rbuf, const Segment* segments,
                         SegEntry* entries, SegExit* exits, uint32_t first_segment, const uint32_t* list, uint8_t* sched,
                         uint32_t count) {
  const DeviceTables& dt = dev_tables();
  ParseArgs a;
  a.P = P;
  a.T.text = B.text;
  a.T.info = B.info[rbuf];
  a.T.sorted = B.sorted[rbuf];
  a.T.sorted_tag = (!B.rows && B.stag) ? B.sorted_tag[rbuf] : nullptr;
  a.T.rows = B.rows;
  a.T.dict_items = B.dict_items;
  a.T.run_end = B.run_end;
  a.T.search_log = B.rows ? nullptr : B.search_log;
  a.T.flags_next = B.flags[flags_out];
  a.T.cmds = B.cmds;
  a.T.dict_hash = dt.dict_hash;
  a.T.dict_data = dt.dict_data;
  a.T.dict_offsets_by_length = dt.dict_offsets_by_length;
  a.T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  a.T.dist_postfix_bits = P.dist_postfix_bits;
  a.T.num_direct_distance_codes = P.num_direct_distance_codes;
  {
    ParseTiming& ptw = parse_timing();
    if (!ptw.work_dev) {
      HIP_CHECK(hipMalloc((void**)&ptw.work_dev, 4 * sizeof(unsigned long long)));
      dev_memset(ptw.work_dev, 0, 4 * sizeof(unsigned long long));
    }
    a.T.work = ptw.work_dev;
  }
  a.segments = segments;
  a.entries = entries;
  a.exits = exits;
  a.first_segment = first_segment;
  a.list = list;
  a.sched = sched;
  static const uint32_t tuned = getenv("BROTLI_MI355X_CONT") ? (uint32_t)atoi(getenv("BROTLI_MI355X_CONT")) : kMaxContinuation;
  a.max_continuation = count <= 256 ? 1024u : tuned;
  a.count = count;
  // HIP events around every launch of the dominant kernel (same stream): bench.py's roofline numbers
  ParseTiming& pt = parse_timing();
  if (pt.used == pt.events.size()) {
    hipEvent_t n0, n1;
    HIP_CHECK(hipEventCreate(&n0));
    HIP_CHECK(hipEventCreate(&n1));
    pt.events.push_back(std::make_pair(n0, n1));
  }
  const hipEvent_t e0 = pt.events[pt.used].first, e1 = pt.events[pt.used].second;
  pt.used++;
  HIP_CHECK(hipEventRecord(e0, BR_STREAM));
  static const bool xcd_aware = getenv("BROTLI_MI355X_NO_XCD_MAP") == nullptr;
  a.per_xcd = (xcd_aware && count >= 64) ? (count + 7) / 8 : 0;
  const uint32_t grid = a.per_xcd ? a.per_xcd * 8 : count;
  static const uint32_t lds_pad = getenv("BROTLI_MI355X_LDS_PAD") ? (uint32_t)atoi(getenv("BROTLI_MI355X_LDS_PAD")) : 0u;  // occupancy experiments
  // four chains per wavefront where a launch holds every segment of a large input (round 0, the warm-up): lz77_groups.h
  // (measured in round 6, profiles/r06_group_kernel_experiment.json: 1.7 x fewer instructions per search than the wave-per-chain kernel
  // and byte-identical, but 139 vector registers = three wavefronts per SIMD, which do not hide the candidate-text round trip of a
  // step: round 0 + warm-up 10.3 ms against 5.2.  Off unless asked for: BROTLI_MI355X_GROUPS_MIN=<chains>.)
  static const uint32_t groups_min = getenv("BROTLI_MI355X_GROUPS_MIN") ? (uint32_t)atoi(getenv("BROTLI_MI355X_GROUPS_MIN")) : 0xffffffffu;
  const bool groups = B.rows && list == nullptr && sched == nullptr && count >= groups_min && plain_q5_config(P) && getenv("BROTLI_MI355X_NO_SPEC") == nullptr;
  if (groups) {
    const bool own = segments == B.segments;
    a.T.checkpoints = own ? (Checkpoint*)B.checkpoints : nullptr;
    const uint32_t waves = (count + 3u) / 4u;
    a.per_xcd = (xcd_aware && waves >= 64) ? (waves + 7) / 8 : 0;
    const uint32_t ggrid = a.per_xcd ? a.per_xcd * 8 : waves;
    if (P.htl == 8) hipLaunchKernelGGL((k_parse_groups<8>), dim3(ggrid), dim3(64), 0, BR_STREAM, a);
    else hipLaunchKernelGGL((k_parse_groups<4>), dim3(ggrid), dim3(64), 0, BR_STREAM, a);
  } else if (P.hasher_kind == 9) {
    hipLaunchKernelGGL((k_parse_segments<true, false>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
  } else if (B.rows) {
    static const bool spec_off = getenv("BROTLI_MI355X_NO_SPEC") != nullptr;
    static const bool splice_off = getenv("BROTLI_MI355X_NO_SPLICE") != nullptr;
    const uint32_t spec = (!spec_off && plain_q5_config(P)) ? P.htl : 0u;
    // checkpoints: recorded by every parse of the real segments (not the warm-up's), used by the list launches
    const bool own_segments = segments == B.segments;
    a.T.checkpoints = own_segments ? (Checkpoint*)B.checkpoints : nullptr;
    a.T.rows_changed_lo = B.rows_changed_lo;
    a.T.rows_changed_hi = B.rows_changed_hi;
    static const uint32_t splice_part_off = getenv("BROTLI_MI355X_SPLICE_OFF") ? (uint32_t)atoi(getenv("BROTLI_MI355X_SPLICE_OFF")) : 0u;
    a.T.splice_off = splice_part_off;
    const bool splice = !splice_off && sched != nullptr && own_segments && B.checkpoints != nullptr && B.splice_lists != 0;
    if (spec == 8) {
      if (splice) hipLaunchKernelGGL((k_parse_segments<false, true, 8, true>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
      else hipLaunchKernelGGL((k_parse_segments<false, true, 8>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
    } else if (spec == 4) {
      if (splice) hipLaunchKernelGGL((k_parse_segments<false, true, 4, true>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
      else hipLaunchKernelGGL((k_parse_segments<false, true, 4>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
    } else {
      if (splice) hipLaunchKernelGGL((k_parse_segments<false, true, 0, true>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
      else hipLaunchKernelGGL((k_parse_segments<false, true>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
    }
  } else if (P.block_bits > 7) {
    hipLaunchKernelGGL((k_parse_segments<false, false, 0, false, true>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
  } else {
    hipLaunchKernelGGL((k_parse_segments<false, false>), dim3(grid), dim3(64), lds_pad, BR_STREAM, a);
  }
  HIP_CHECK(hipEventRecord(e1, BR_STREAM));
  HIP_CHECK(hipGetLastError());
  pt.counts.push_back(count);
  pt.segments += count;
}

void lz77_parse_round(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, uint32_t first_segment) {
  if (first_segment >= P.num_segments) return;
  launch_parse(P, B, which, which ^ 1, rbuf, B.segments, B.entries, B.exits, first_segment, nullptr, nullptr, P.num_segments - first_segment);
}

void lz77_parse_list(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const uint32_t* list_dev, uint8_t* sched_dev,
                     uint32_t count) {
  if (count == 0) return;
  launch_parse(P, B, which, which ^ 1, rbuf, B.segments, B.entries, B.exits, 0, list_dev, sched_dev, count);
}

void lz77_parse_custom(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const Segment* segments_dev,
                       SegEntry* entries_dev, SegExit* exits_dev, uint32_t count) {
  if (count == 0) return;
  launch_parse(P, B, which, which ^ 1, rbuf, segments_dev, entries_dev, exits_dev, 0, nullptr, nullptr, count);
}

// ------------------------------------------------------------------------------------------ flag diff
// After a parse launch: which keys had a stored flag change?  (The chains only write flags; comparing the two
// flag arrays in one streaming pass is far cheaper than having every chain read the old flag of each position.)
// the flip cells (Lz77Buffers::flip_cells): where, coarsely, the changed positions of a launch lie
struct FlipCells {
  uint32_t* bits;  // null: not kept
  uint32_t shift, per_key, words;  // words: index of the "too many to tell" mark
};
__device__ __forceinline__ void mark_flip_cell(const FlipCells& c, uint32_t key, uint32_t pos) {
  const uint32_t cell = key * c.per_key + (pos >> c.shift);
  const uint32_t bit = 1u << (cell & 31u);
  if (!(c.bits[cell >> 5] & bit)) atomicOr(&c.bits[cell >> 5], bit);  // (look first: most changes fall into cells that are set already)
}
static FlipCells flip_cells_of(const Lz77Buffers& B) {
  FlipCells c;
  c.bits = B.rows ? B.flip_cells : nullptr;
  c.shift = B.cell_shift;
  c.per_key = B.cells_per_key;
  c.words = (uint32_t)(((size_t)65536 * B.cells_per_key) / 32);
  return c;
}
static void clear_flip_cells(const Lz77Buffers& B) {
  if (B.rows && B.flip_cells) dev_memset(B.flip_cells, 0, ((size_t)65536 * B.cells_per_key) / 8 + 8);
}

__global__ __launch_bounds__(256) void k_diff_flags(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ next, uint32_t n,
                                                     const uint16_t* __restrict__ keys, uint32_t* __restrict__ changed_keys,
                                                     uint32_t* __restrict__ changed_count, uint32_t cap, uint32_t emit_positions, FlipCells cells) {
  const uint32_t words = (n + 15) / 16;  // both arrays are padded by 64 bytes
  const uint32_t lane = threadIdx.x & 63u;
  // (the loop bound is wave-uniform: all lanes of a wave take part in the scan below)
  for (uint32_t w0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; w0 < words; w0 += gridDim.x * blockDim.x) {
    const uint32_t wi = w0 + lane;
    uint32_t d[4] = {0, 0, 0, 0};
    if (wi < words) {
      const uint4 a = ((const uint4*)prev)[wi], b = ((const uint4*)next)[wi];
      // a position changed if its stored bit or its "stored as a masked position" bit (kFlagMasked) did: one mark per byte
      d[0] = (a.x ^ b.x) & 0x05050505u;
      d[1] = (a.y ^ b.y) & 0x05050505u;
      d[2] = (a.z ^ b.z) & 0x05050505u;
      d[3] = (a.w ^ b.w) & 0x05050505u;
      for (uint32_t j = 0; j < 4; ++j) d[j] = (d[j] | (d[j] >> 2)) & 0x01010101u;
      // bytes past n are padding
      for (uint32_t j = 0; j < 16; ++j)
        if (wi * 16 + j >= n) d[j >> 2] &= ~(1u << (8 * (j & 3)));
    }
    const uint32_t mine = (uint32_t)(__popc(d[0]) + __popc(d[1]) + __popc(d[2]) + __popc(d[3]));
    if (__ballot(mine != 0) == 0) continue;
    // one atomic per wave: exclusive scan of the per-lane counts, the last lane reserves the range
    uint32_t incl = mine;
    for (uint32_t off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= off) incl += up;
    }
    uint32_t base = 0;
    // (once the list has overflowed only "more than cap" matters: incompressible input flips most of its flags in round 0, and a
    // million reservations on one word were 10 of the 12 ms this pass took at 1 GiB)
    uint32_t sat = 0;
    if (lane == 63) {
      if (*(volatile const uint32_t*)changed_count > cap) {
        base = cap;
        sat = 1;
        if (cells.bits) cells.bits[cells.words] = 1;  // (too many changes to tell where: every cell counts as set)
      } else {
        base = atomicAdd(changed_count, incl);
      }
    }
    base = (uint32_t)__shfl((int)base, 63, 64);
    sat = (uint32_t)__shfl((int)sat, 63, 64);
    uint32_t idx = base + incl - mine;
    const bool mark = cells.bits != nullptr && sat == 0;
    if (mine == 0 || (idx >= cap && !mark)) continue;
    for (uint32_t j = 0; j < 16; ++j) {
      if ((d[j >> 2] >> (8 * (j & 3))) & 1u) {
        if (idx < cap) changed_keys[idx] = emit_positions ? wi * 16 + j : (uint32_t)keys[wi * 16 + j];
        if (mark) mark_flip_cell(cells, keys[wi * 16 + j], wi * 16 + j);
        ++idx;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_scatter_entries(SegEntry* __restrict__ entries, const uint32_t* __restrict__ index,
                                                          const SegEntry* __restrict__ src, uint32_t count) {
  // 16 words per entry: one thread per word
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 16u) return;
  ((uint32_t*)(entries + index[t >> 4]))[t & 15u] = ((const uint32_t*)src)[t];
}

void lz77_scatter_entries(const Lz77Buffers& B, const uint32_t* index_dev, const SegEntry* entries_dev, uint32_t count) {
  static_assert(sizeof(SegEntry) == 64, "k_scatter_entries copies 16 words per entry");
  if (count == 0) return;
  hipLaunchKernelGGL(k_scatter_entries, dim3((count * 16u + 255) / 256), dim3(256), 0, BR_STREAM, B.entries, index_dev, entries_dev, count);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_gather_results(const SegExit* __restrict__ exits, const SegEntry* __restrict__ entries,
                                                         const uint32_t* __restrict__ list, uint32_t count, const uint8_t* __restrict__ sched,
                                                         uint32_t num_segments, SegExit* __restrict__ exits_out, uint32_t* __restrict__ cont_count,
                                                         uint32_t* __restrict__ cont_index, SegExit* __restrict__ cont_exits,
                                                         SegEntry* __restrict__ cont_entries) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) exits_out[t] = exits[list[t]];
  if (t < num_segments && sched[t] == 3) {
    const uint32_t j = atomicAdd(cont_count, 1u);
    cont_index[j] = t;
    cont_exits[j] = exits[t];
    cont_entries[j] = entries[t];
  }
}

void lz77_gather_results(const Lz77Buffers& B, const uint32_t* list_dev, uint32_t count, const uint8_t* sched_dev, uint32_t num_segments,
                         SegExit* exits_out, uint32_t* cont_count, uint32_t* cont_index, SegExit* cont_exits, SegEntry* cont_entries) {
  dev_memset(cont_count, 0, 4);
  const uint32_t n = count > num_segments ? count : num_segments;
  hipLaunchKernelGGL(k_gather_results, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, B.exits, B.entries, list_dev, count, sched_dev, num_segments,
                     exits_out, cont_count, cont_index, cont_exits, cont_entries);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ bursts (device_api.h)
__global__ __launch_bounds__(256) void k_chain_check(const Segment* __restrict__ segments, const SegEntry* __restrict__ entries,
                                                      const SegExit* __restrict__ exits, uint32_t num_segments, const uint8_t* __restrict__ sched,
                                                      uint8_t* __restrict__ touched, uint8_t* __restrict__ entry_dirty, SegEntry* __restrict__ new_entries,
                                                      uint32_t* __restrict__ rows_lo, uint32_t* __restrict__ rows_hi, uint8_t* __restrict__ stale) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < num_segments) br_chain_check(segments, entries, exits, num_segments, k, sched, touched, entry_dirty, new_entries, rows_lo, rows_hi, stale);
}
void lz77_chain_check(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U) {
  hipLaunchKernelGGL(k_chain_check, dim3((P.num_segments + 255) / 256), dim3(256), 0, BR_STREAM, B.segments, B.entries, B.exits, P.num_segments,
                     U.sched, U.touched, U.entry_dirty, U.new_entries, B.rows_changed_lo, B.rows_changed_hi, U.stale);
  HIP_CHECK(hipGetLastError());
}

// ---- flag arrays between the launches of a burst (device_api.h): one wavefront per stale segment
__global__ __launch_bounds__(64) void k_flags_catch_up(const Segment* __restrict__ segments, uint32_t num_segments, uint8_t* __restrict__ stale,
                                                        const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
  const uint32_t k = blockIdx.x;
  if (k >= num_segments || !stale[k]) return;
  const uint32_t a = segments[k].start, b = segments[k].end;
  // whole 16-byte words inside [a, b) as words, the ragged ends byte by byte (a word on the boundary belongs to two segments)
  const uint32_t wa = (a + 15u) / 16u, wb = b / 16u;
  if (wa < wb) {
    for (uint32_t w = wa + threadIdx.x; w < wb; w += 64) ((uint4*)dst)[w] = ((const uint4*)src)[w];
    for (uint32_t q = a + threadIdx.x; q < wa * 16u; q += 64) dst[q] = src[q];
    for (uint32_t q = wb * 16u + threadIdx.x; q < b; q += 64) dst[q] = src[q];
  } else {
    for (uint32_t q = a + threadIdx.x; q < b; q += 64) dst[q] = src[q];
  }
  if (threadIdx.x == 0) stale[k] = 0;
}
void lz77_flags_catch_up(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, int src, int dst) {
  if (P.num_segments == 0) return;
  hipLaunchKernelGGL(k_flags_catch_up, dim3(P.num_segments), dim3(64), 0, BR_STREAM, B.segments, P.num_segments, U.stale, B.flags[src], B.flags[dst]);
  HIP_CHECK(hipGetLastError());
}

// k_diff_flags over the positions of the stale segments only (same list, same saturating count)
__global__ __launch_bounds__(64) void k_diff_flags_touched(const Segment* __restrict__ segments, uint32_t num_segments, const uint8_t* __restrict__ stale,
                                                            const uint8_t* __restrict__ prev, const uint8_t* __restrict__ next,
                                                            uint32_t* __restrict__ changed_pos, uint32_t* __restrict__ changed_count, uint32_t cap,
                                                            const uint16_t* __restrict__ keys, FlipCells cells) {
  const uint32_t k = blockIdx.x;
  if (k >= num_segments || !stale[k]) return;
  const uint32_t a = segments[k].start, b = segments[k].end;
  const uint32_t lane = threadIdx.x;
  for (uint32_t w0 = a / 16u; w0 * 16u < b; w0 += 64) {
    const uint32_t wi = w0 + lane;
    uint32_t d[4] = {0, 0, 0, 0};
    if (wi * 16u < b) {
      const uint4 x = ((const uint4*)prev)[wi], y = ((const uint4*)next)[wi];  // (both arrays are padded by 64 bytes)
      d[0] = (x.x ^ y.x) & 0x05050505u;
      d[1] = (x.y ^ y.y) & 0x05050505u;
      d[2] = (x.z ^ y.z) & 0x05050505u;
      d[3] = (x.w ^ y.w) & 0x05050505u;
      for (uint32_t j = 0; j < 4; ++j) d[j] = (d[j] | (d[j] >> 2)) & 0x01010101u;
      if (wi * 16u < a || wi * 16u + 16u > b)
        for (uint32_t j = 0; j < 16; ++j)
          if (wi * 16u + j < a || wi * 16u + j >= b) d[j >> 2] &= ~(1u << (8 * (j & 3)));
    }
    const uint32_t mine = (uint32_t)(__popc(d[0]) + __popc(d[1]) + __popc(d[2]) + __popc(d[3]));
    if (__ballot(mine != 0) == 0) continue;
    uint32_t incl = mine;
    for (uint32_t off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= off) incl += up;
    }
    uint32_t base = 0;
    uint32_t sat = 0;
    if (lane == 63) {
      if (*(volatile const uint32_t*)changed_count > cap) {
        base = cap;
        sat = 1;
        if (cells.bits) cells.bits[cells.words] = 1;
      } else {
        base = atomicAdd(changed_count, incl);
      }
    }
    base = (uint32_t)__shfl((int)base, 63, 64);
    sat = (uint32_t)__shfl((int)sat, 63, 64);
    uint32_t idx = base + incl - mine;
    const bool mark = cells.bits != nullptr && sat == 0;
    if (mine == 0 || (idx >= cap && !mark)) continue;
    for (uint32_t j = 0; j < 16; ++j) {
      if ((d[j >> 2] >> (8 * (j & 3))) & 1u) {
        if (idx < cap) changed_pos[idx] = wi * 16u + j;
        if (mark) mark_flip_cell(cells, keys[wi * 16u + j], wi * 16u + j);
        ++idx;
      }
    }
  }
}
void lz77_diff_flags_touched(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, int prev, int next) {
  dev_memset(B.changed_count, 0, 4);
  clear_flip_cells(B);
  if (P.num_segments == 0) return;
  hipLaunchKernelGGL(k_diff_flags_touched, dim3(P.num_segments), dim3(64), 0, BR_STREAM, B.segments, P.num_segments, U.stale, B.flags[prev], B.flags[next],
                     B.changed_keys, B.changed_count, B.changed_cap, B.keys, flip_cells_of(B));
  HIP_CHECK(hipGetLastError());
}
void lz77_reset_rows_changed(const Lz77Params& P, const Lz77Buffers& B) {
  if (B.rows_changed_lo == nullptr) return;
  HIP_CHECK(hipMemsetAsync(B.rows_changed_lo, 0xff, (size_t)P.num_segments * 4, BR_STREAM));
  dev_memset(B.rows_changed_hi, 0, (size_t)P.num_segments * 4);
}
__global__ __launch_bounds__(64) void k_drop_checkpoints(Checkpoint* __restrict__ checkpoints, const Segment* __restrict__ segments,
                                                          const uint32_t* __restrict__ list, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) br_drop_checkpoints(checkpoints, segments[list[i]]);
}
void lz77_drop_checkpoints(const Lz77Params& P, const Lz77Buffers& B, const uint32_t* list_dev, uint32_t count) {
  if (B.checkpoints == nullptr || count == 0) return;
  hipLaunchKernelGGL(k_drop_checkpoints, dim3((count + 63) / 64), dim3(64), 0, BR_STREAM, (Checkpoint*)B.checkpoints, B.segments, list_dev, count);
  HIP_CHECK(hipGetLastError());
  (void)P;
}

__global__ __launch_bounds__(256) void k_burst_count(uint32_t num_segments, const uint8_t* __restrict__ sched, const uint8_t* __restrict__ cand_dirty,
                                                      const uint8_t* __restrict__ entry_dirty, uint32_t* __restrict__ counters) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool go = k < num_segments && (cand_dirty[k] != 0 || entry_dirty[k] != 0 || sched[k] == 2);
  const unsigned long long m = __ballot(go);
  if (m != 0 && (threadIdx.x & 63u) == 0) atomicAdd(&counters[0], (uint32_t)__popcll(m));
}
void lz77_burst_count(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U) {
  dev_memset(U.counters, 0, 4);
  hipLaunchKernelGGL(k_burst_count, dim3((P.num_segments + 255) / 256), dim3(256), 0, BR_STREAM, P.num_segments, U.sched, U.cand_dirty, U.entry_dirty,
                     U.counters);
  HIP_CHECK(hipGetLastError());
  (void)B;
}

__global__ __launch_bounds__(256) void k_burst_schedule(SegEntry* __restrict__ entries, uint32_t num_segments, uint8_t* __restrict__ sched,
                                                         uint8_t* __restrict__ cand_dirty, uint8_t* __restrict__ entry_dirty,
                                                         const SegEntry* __restrict__ new_entries, uint32_t* __restrict__ list, uint32_t* __restrict__ counters) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool go = k < num_segments && br_burst_schedule_one(entries, k, sched, cand_dirty, entry_dirty, new_entries);
  // (segments of one wave keep their order in the list, waves land in the order they get here: chains of neighbouring
  // segments stay neighbours, which is all the launch cares about)
  const unsigned long long m = __ballot(go);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(&counters[0], (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, (int)leader, 64);
  if (go) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = k;
}
void lz77_burst_schedule(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U) {
  dev_memset(U.counters, 0, 4);
  hipLaunchKernelGGL(k_burst_schedule, dim3((P.num_segments + 255) / 256), dim3(256), 0, BR_STREAM, B.entries, P.num_segments, U.sched, U.cand_dirty,
                     U.entry_dirty, U.new_entries, U.list, U.counters);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_gather_touched(const SegExit* __restrict__ exits, const SegEntry* __restrict__ entries, uint32_t num_segments,
                                                         uint8_t* __restrict__ touched, uint32_t* __restrict__ counters, uint32_t* __restrict__ index_out,
                                                         SegExit* __restrict__ exits_out, SegEntry* __restrict__ entries_out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_segments || !touched[k]) return;
  touched[k] = 0;
  const uint32_t j = atomicAdd(&counters[1], 1u);
  index_out[j] = k;
  exits_out[j] = exits[k];
  entries_out[j] = entries[k];
}
void lz77_gather_touched(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, uint32_t* index_out, SegExit* exits_out, SegEntry* entries_out) {
  dev_memset(U.counters + 1, 0, 4);
  hipLaunchKernelGGL(k_gather_touched, dim3((P.num_segments + 255) / 256), dim3(256), 0, BR_STREAM, B.exits, B.entries, P.num_segments, U.touched,
                     U.counters, index_out, exits_out, entries_out);
  HIP_CHECK(hipGetLastError());
}

void lz77_diff_flags(const Lz77Params& P, const Lz77Buffers& B, int prev, int next) {
  const uint32_t n = P.total_bytes;
  dev_memset(B.changed_count, 0, 4);
  clear_flip_cells(B);
  if (n == 0) return;
  uint32_t blocks = ((n + 15) / 16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  // with candidate rows the list holds the changed POSITIONS (consumed on the device by lz77_rows_update)
  hipLaunchKernelGGL(k_diff_flags, dim3(blocks), dim3(256), 0, BR_STREAM, B.flags[prev], B.flags[next], n, B.keys, B.changed_keys, B.changed_count,
                     B.rows ? B.changed_cap : kChangedCap, B.rows ? 1u : 0u, flip_cells_of(B));
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ validate
// After some flags changed, the candidate list of a searched position may differ from the one its chain
// saw.  Comparing the two rank structures entry by entry (no text access) finds those positions exactly.
__global__ __launch_bounds__(256) void k_validate(const uint8_t* __restrict__ text, const uint8_t* __restrict__ flags, const uint2* __restrict__ info_old,
                                                   const uint32_t* __restrict__ sorted_old, const uint2* __restrict__ info_new,
                                                   const uint32_t* __restrict__ sorted_new, uint32_t n, SegGeometry geo,
                                                   uint8_t* __restrict__ dirty, uint32_t* __restrict__ recheck_list,
                                                   uint32_t* __restrict__ recheck_count, uint32_t recheck_cap) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    if (p < geo.first_block_start) continue;
    const bool searched = (flags[p] & kFlagSearched) != 0;
    const uint32_t in_front = searched ? 0xffffffffu : chain_in_front_if_near_boundary(p, geo);
    if (!searched && in_front == 0xffffffffu) continue;
    const uint2 a = info_old[p], b = info_new[p];
    const uint32_t na = min(a.y & 0xffffu, geo.block_size), nb = min(b.y & 0xffffu, geo.block_size);
    bool same = na == nb;
    // four entries per trip, all eight loads in flight (a row is one or two cache lines)
    for (uint32_t j = 0; same && j < na; j += 4) {
      uint32_t x[4], y[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        const bool in = j + i < na;
        x[i] = in ? sorted_old[a.x - 1 - j - i] : 0u;
        y[i] = in ? sorted_new[b.x - 1 - j - i] : 0u;
      }
      same = x[0] == y[0] && x[1] == y[1] && x[2] == y[2] && x[3] == y[3];
    }
    if (same) continue;
    if (br_row_change_matters(text, p, sorted_old + a.x - 1, na, sorted_new + b.x - 1, nb)) {
      if (searched) list_or_mark(p, geo, dirty, recheck_list, recheck_count, recheck_cap);
      else dirty[in_front] = 1;
    }
  }
}

void lz77_validate(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf_old, int rbuf_new, const SegGeometry& geo,
                   uint8_t* dirty_dev) {
  const uint32_t n = P.total_bytes;
  if (n == 0) return;
  uint32_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_validate, dim3(blocks), dim3(256), 0, BR_STREAM, B.text, B.flags[which], (const uint2*)B.info[rbuf_old], B.sorted[rbuf_old],
                     (const uint2*)B.info[rbuf_new], B.sorted[rbuf_new], n, geo, dirty_dev, B.recheck_list, B.recheck_count, B.recheck_cap);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ single searches again
struct RecheckArgs {
  Lz77Params P;
  ChainTables T;
  const uint32_t* list;
  const uint32_t* count;
  uint32_t cap;
  SegGeometry geo;
  uint8_t* dirty;
};
template <bool kH9, bool kDeep = false>
__global__ __launch_bounds__(64) void k_recheck_searches(RecheckArgs a) {
  __shared__ ChainScratchT<kH9, false, kDeep> scratch;
  uint32_t n = *a.count;
  if (n > a.cap) n = a.cap;
  for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
    const uint32_t p = a.list[item];
    // (input blocks are cut at prefix_bytes + k * block_bytes -- mark_dirty's geometry; the first one merely STARTS later on a catable
    // stream, whose first two bytes are stored raw: counting the blocks from first_block_start gave the first two positions of every
    // block the end of the block in front, i.e. a search of one or two bytes that "found the same as before" -- nothing -- whatever
    // the candidates had become.  Found by the API sweep in round 6, seed 62 case 213.)
    const uint32_t blk = (p - a.geo.prefix_bytes) / a.geo.block_bytes;
    const uint64_t end64 = (uint64_t)a.geo.prefix_bytes + (uint64_t)(blk + 1) * a.geo.block_bytes;
    const uint32_t blk_end = end64 < a.P.total_bytes ? (uint32_t)end64 : a.P.total_bytes;
    const bool same = br_recheck_search<kH9>(a.P, a.T, scratch, p, blk_end);
    if (!same && threadIdx.x == 0) mark_dirty(p, a.geo, a.dirty);
  }
}

void lz77_recheck_searches(const Lz77Params& P, const Lz77Buffers& B, int rbuf, const SegGeometry& geo, uint8_t* dirty_dev) {
  if (B.recheck_list == nullptr || B.search_log == nullptr || B.rows != nullptr) return;
  const DeviceTables& dt = dev_tables();
  RecheckArgs a;
  a.P = P;
  a.T.text = B.text;
  a.T.info = B.info[rbuf];
  a.T.sorted = B.sorted[rbuf];
  a.T.sorted_tag = B.stag ? B.sorted_tag[rbuf] : nullptr;
  a.T.rows = nullptr;
  a.T.run_end = B.run_end;
  a.T.search_log = B.search_log;
  a.T.flags_next = nullptr;
  a.T.cmds = nullptr;
  a.T.dict_hash = dt.dict_hash;
  a.T.dict_data = dt.dict_data;
  a.T.dict_offsets_by_length = dt.dict_offsets_by_length;
  a.T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  a.T.dist_postfix_bits = P.dist_postfix_bits;
  a.T.num_direct_distance_codes = P.num_direct_distance_codes;
  a.T.work = nullptr;
  a.list = B.recheck_list;
  a.count = B.recheck_count;
  a.cap = B.recheck_cap;
  a.geo = geo;
  a.dirty = dirty_dev;
  const uint32_t grid = B.recheck_cap < 16384u ? B.recheck_cap : 16384u;
  if (P.hasher_kind == 9) {
    hipLaunchKernelGGL((k_recheck_searches<true>), dim3(grid), dim3(64), 0, BR_STREAM, a);
  } else if (P.block_bits > 7) {
    hipLaunchKernelGGL((k_recheck_searches<false, true>), dim3(grid), dim3(64), 0, BR_STREAM, a);
  } else {
    hipLaunchKernelGGL((k_recheck_searches<false>), dim3(grid), dim3(64), 0, BR_STREAM, a);
  }
  HIP_CHECK(hipGetLastError());
}

#if defined(BR_CHAIN_PROFILE)
__device__ unsigned long long g_chain_prof[16];
#endif

void lz77_parse_timing(double* total_ms, uint32_t* launches, uint64_t* segments, uint64_t* work) {
  {
    ParseTiming& ptw = parse_timing();
    unsigned long long h[4] = {0, 0, 0, 0};
    if (ptw.work_dev) {
      HIP_CHECK(hipMemcpyAsync(h, ptw.work_dev, sizeof(h), hipMemcpyDeviceToHost, BR_STREAM));
      HIP_CHECK(hipStreamSynchronize(BR_STREAM));
      dev_memset(ptw.work_dev, 0, sizeof(h));
    }
    if (work)
      for (int i = 0; i < 3; ++i) work[i] = h[i];
  }
#if defined(BR_CHAIN_PROFILE)
  {
    unsigned long long h[16];
    HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chain_prof), sizeof(h)));
    fprintf(stderr, "chain profile: segments %llu total ticks %llu probe %llu (%llu calls; setup %llu, of which refill %llu in %llu refills) fold %llu (%llu calls) cmds %llu searches %llu\n",
            h[5], h[0], h[1], h[3], h[7], h[8], h[9], h[2], h[4], h[6], h[10]);
    unsigned long long z[16] = {0};
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_prof), z, sizeof(z)));
  }
#endif
  ParseTiming& pt = parse_timing();
  double ms = 0;
  static const bool show = getenv("BROTLI_MI355X_DEBUG_LAUNCH") != nullptr;
  size_t idx = 0;
  for (; idx < pt.used; ++idx) {
    const auto& ev = pt.events[idx];
    float t = 0;
    HIP_CHECK(hipEventSynchronize(ev.second));
    HIP_CHECK(hipEventElapsedTime(&t, ev.first, ev.second));
    if (show) fprintf(stderr, "  parse launch %zu: %u chains, %.3f ms\n", idx, pt.counts[idx], t);
    ms += t;
  }
  *total_ms = ms;
  *launches = (uint32_t)pt.used;
  pt.counts.clear();
  *segments = pt.segments;
  pt.used = 0;
  pt.segments = 0;
}

// ------------------------------------------------------------------------------------------ misc
// see lz77_check_cache: one wave per segment, one lane per position
__global__ __launch_bounds__(64) void k_check_cache(const uint8_t* __restrict__ text, const uint8_t* __restrict__ flags,
                                                     const Segment* __restrict__ segments, const CacheCheck* __restrict__ items,
                                                     uint32_t max_backward_limit, uint32_t ndist, uint8_t* __restrict__ ok) {
  const CacheCheck it = items[blockIdx.x];
  const Segment seg = segments[it.segment];
  // the candidate distances FindLongestMatch derives from the cache (adv_prepare_distance_cache, mod.rs:632-651):
  // 4 at qualities 5-6, 10 (last distance +-1..3) at 7-8
  int32_t dc[16];
  for (int i = 0; i < 4; ++i) dc[i] = it.cache[i];
  for (int i = 4; i < 16; ++i) dc[i] = 0;
  br_prepare_distance_cache(dc, ndist);
  bool hit = false;
  for (uint32_t p = seg.start + threadIdx.x; p < seg.end; p += 64) {
    if (!(flags[p] & kFlagSearched)) continue;
    const uint32_t max_backward = p < max_backward_limit ? p : max_backward_limit;
    const uint32_t cur = (uint32_t)text[p] | ((uint32_t)text[p + 1] << 8);
    for (uint32_t i = 0; i < ndist; ++i) {
      const int64_t d = (int64_t)dc[i];
      if (d <= 0 || d > (int64_t)max_backward) continue;
      const uint32_t q = p - (uint32_t)d;
      hit |= cur == ((uint32_t)text[q] | ((uint32_t)text[q + 1] << 8));
    }
  }
  const bool any = __any(hit);
  if (threadIdx.x == 0) ok[blockIdx.x] = any ? 0 : 1;
}

void lz77_check_cache(const Lz77Params& P, const Lz77Buffers& B, int which, const CacheCheck* items_dev, uint32_t count, uint8_t* ok_dev) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_check_cache, dim3(count), dim3(64), 0, BR_STREAM, B.text, B.flags[which], B.segments, items_dev, P.max_backward_limit,
                     P.ndist, ok_dev);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_sample_histogram(const uint8_t* __restrict__ text, uint32_t start, uint32_t samples,
                                                           uint32_t* __restrict__ histo) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < samples; i += gridDim.x * blockDim.x)
    atomicAdd(&h[text[start + i * 13u]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&histo[threadIdx.x], h[threadIdx.x]);
}

// Every-13th-byte histograms (should_compress, encode.rs:1325-1354) of several spans of the text at once: span r =
// {start, bytes} samples start, start + 13, ...; out[r * 256 + v] = number of samples with byte v.  One workgroup per
// kSampleChunk samples of a span (blockIdx.y = span); out is zeroed by the caller.
static constexpr uint32_t kSampleChunk = 8192;
__global__ __launch_bounds__(256) void k_sample_histograms(const uint8_t* __restrict__ text, const uint32_t* __restrict__ ranges,
                                                            uint32_t* __restrict__ out) {
  const uint32_t start = ranges[2 * blockIdx.y], bytes = ranges[2 * blockIdx.y + 1];
  const uint32_t samples = (bytes + 12) / 13;
  const uint32_t first = blockIdx.x * kSampleChunk;
  if (first >= samples) return;
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t last = min(first + kSampleChunk, samples);
  for (uint32_t i = first + threadIdx.x; i < last; i += 256) atomicAdd(&h[text[start + i * 13u]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&out[(size_t)blockIdx.y * 256 + threadIdx.x], h[threadIdx.x]);
}

// Runs on a stream of its own and waits for it: the caller is the host resolver in the middle of a pass, while the calling
// thread's stream is busy for milliseconds with the row update queued behind the parse (the text itself is long in place).
// Streams and their buffers are kept in a process-wide list (never destroyed: no HIP call from a thread's exit path).
namespace {
struct SideLane {
  hipStream_t stream = nullptr;
  uint32_t* dev = nullptr;
  uint32_t* host = nullptr;  // page-locked
  size_t words = 0;
  int device = -1;
};
std::mutex g_side_mu;
std::vector<SideLane*> g_side_free;
SideLane* side_acquire(size_t words) {
  int device = 0;
  HIP_CHECK(hipGetDevice(&device));
  SideLane* lane = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_free.size(); ++i)
      if (g_side_free[i]->device == device) {
        lane = g_side_free[i];
        g_side_free.erase(g_side_free.begin() + i);
        break;
      }
  }
  if (!lane) {
    lane = new SideLane;
    lane->device = device;
    HIP_CHECK(hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking));
  }
  if (lane->words < words) {
    if (lane->dev) HIP_CHECK(hipFree(lane->dev));
    if (lane->host) HIP_CHECK(hipHostFree(lane->host));
    lane->words = words + words / 2 + 1024;
    HIP_CHECK(hipMalloc((void**)&lane->dev, lane->words * 4));
    HIP_CHECK(hipHostMalloc((void**)&lane->host, lane->words * 4, hipHostMallocDefault));
  }
  return lane;
}
void side_release(SideLane* lane) {
  std::lock_guard<std::mutex> lock(g_side_mu);
  g_side_free.push_back(lane);
}
// a lane goes back to the list however the call ends (a HIP_CHECK that throws included)
struct SideLaneHold {
  SideLane* lane;
  explicit SideLaneHold(size_t words) : lane(side_acquire(words)) {}
  ~SideLaneHold() { side_release(lane); }
  SideLaneHold(const SideLaneHold&) = delete;
  SideLaneHold& operator=(const SideLaneHold&) = delete;
};
}  // namespace

void lz77_sample_histograms(const uint8_t* text, const uint32_t* ranges, uint32_t count, uint32_t* out) {
  if (count == 0) return;
  uint32_t most = 0;
  for (uint32_t r = 0; r < count; ++r) most = std::max(most, (ranges[2 * r + 1] + 12) / 13);
  const size_t range_words = (size_t)count * 2, out_words = (size_t)count * 256;
  SideLaneHold hold(range_words + out_words);
  SideLane* lane = hold.lane;
  uint32_t* ranges_dev = lane->dev;
  uint32_t* out_dev = lane->dev + range_words;
  memcpy(lane->host, ranges, range_words * 4);
  // the text was uploaded on the calling thread's stream: the side stream waits for what that stream holds NOW (every caller has
  // synchronised behind the upload long before; the event makes the lane correct without that knowledge)
  {
    static thread_local hipEvent_t text_ready = nullptr;
    if (!text_ready) HIP_CHECK(hipEventCreateWithFlags(&text_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(text_ready, BR_STREAM));
    HIP_CHECK(hipStreamWaitEvent(lane->stream, text_ready, 0));
  }
  HIP_CHECK(hipMemcpyAsync(ranges_dev, lane->host, range_words * 4, hipMemcpyHostToDevice, lane->stream));
  HIP_CHECK(hipMemsetAsync(out_dev, 0, out_words * 4, lane->stream));
  if (most != 0) {
    for (uint32_t r0 = 0; r0 < count; r0 += 32768) {  // (grid.y is limited to 65 535)
      const uint32_t nr = std::min(count - r0, 32768u);
      hipLaunchKernelGGL(k_sample_histograms, dim3((most + kSampleChunk - 1) / kSampleChunk, nr), dim3(256), 0, lane->stream, text, ranges_dev + 2 * (size_t)r0,
                         out_dev + (size_t)r0 * 256);
    }
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(lane->host + range_words, out_dev, out_words * 4, hipMemcpyDeviceToHost, lane->stream));
  HIP_CHECK(hipStreamSynchronize(lane->stream));
  memcpy(out, lane->host + range_words, out_words * 4);
}

void lz77_sample_histogram(const uint8_t* text, uint32_t start, uint32_t bytes, uint32_t* histo256_dev) {
  dev_memset(histo256_dev, 0, 256 * 4);
  const uint32_t samples = (bytes + 12) / 13;
  uint32_t blocks = (samples + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks == 0) return;
  hipLaunchKernelGGL(k_sample_histogram, dim3(blocks), dim3(256), 0, BR_STREAM, text, start, samples, histo256_dev);
  HIP_CHECK(hipGetLastError());
}

// compacts the per-segment slabs and turns the raw records of the chains into Commands (Command::init, command.rs:273-297)
__global__ __launch_bounds__(256) void k_gather_commands(const Command* __restrict__ slabs, const Segment* __restrict__ segments,
                                                          const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                          Command* __restrict__ out, uint32_t ndirect, uint32_t npostfix) {
  const uint32_t k = blockIdx.x;
  const uint32_t n = counts[k];
  const Command* src = slabs + (size_t)segments[k].cmd_base;
  Command* dst = out + offsets[k];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = br_finish_command(src[i], ndirect, npostfix);
}

void lz77_gather_commands(const Lz77Params& P, const Lz77Buffers& B, uint32_t num_segments, const uint32_t* offsets_dev,
                          const uint32_t* counts_dev, Command* out) {
  if (num_segments == 0) return;
  hipLaunchKernelGGL(k_gather_commands, dim3(num_segments), dim3(256), 0, BR_STREAM, B.cmds, B.segments, offsets_dev, counts_dev, out,
                     P.num_direct_distance_codes, P.dist_postfix_bits);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(64) void k_patch_commands(Command* cmds, const CmdPatch* patches, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) br_apply_patch(cmds, patches[i]);
}

void lz77_patch_commands(Command* cmds, const CmdPatch* patches_dev, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_patch_commands, dim3((n + 63) / 64), dim3(64), 0, BR_STREAM, cmds, patches_dev, n);
  HIP_CHECK(hipGetLastError());
}


// ------------------------------------------------------------------------------------------ live chains (lz77_live.h)
static LiveIndex live_index_of(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which) {
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

static ChainTables live_chain_tables(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int flags_out) {
  const DeviceTables& dt = dev_tables();
  ChainTables T;
  T.text = B.text;
  T.info = nullptr;
  T.sorted = nullptr;
  T.sorted_tag = nullptr;
  T.rows = nullptr;
  T.run_end = B.run_end;
  T.search_log = B.search_log;
  T.flags_next = B.flags[flags_out];
  T.cmds = B.cmds;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  T.dist_postfix_bits = P.dist_postfix_bits;
  T.num_direct_distance_codes = P.num_direct_distance_codes;
  T.work = nullptr;
  T.keys = B.keys;
  T.live_num = L.num;
  T.live_buckets = L.buckets;
  T.live_state = L.state;
  T.logs.logs_16 = dt.logs_16;
  T.logs.logs_8 = dt.logs_8;
  return T;
}

__global__ __launch_bounds__(256) void k_live_slots(const uint32_t* __restrict__ by_key, uint32_t n, uint32_t* __restrict__ slot_of) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slot_of[by_key[i]] = i;
}

void lz77_live_slots(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L) {
  const uint32_t n = P.total_bytes;
  if (n == 0) return;
  hipLaunchKernelGGL(k_live_slots, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, B.by_key, n, L.slot_of);
  HIP_CHECK(hipGetLastError());
}

// per slot: the stored / masked bits of its position (fbits) and the stored bit as a word for the prefix count
__global__ __launch_bounds__(256) void k_live_gather(const uint32_t* __restrict__ by_key, const uint8_t* __restrict__ flags, uint32_t n,
                                                      uint8_t* __restrict__ fbits, uint32_t* __restrict__ rank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  uint32_t f = 0;
  if (i < n) {
    f = flags[by_key[i]] & (kFlagStored | kFlagMasked);
    fbits[i] = (uint8_t)f;
  }
  rank[i] = f & 1u;
}
__global__ __launch_bounds__(256) void k_live_compact(const uint32_t* __restrict__ by_key, const uint8_t* __restrict__ fbits,
                                                       const uint32_t* __restrict__ rank, uint32_t n, uint32_t* __restrict__ entry) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = fbits[i];
  if (f & kFlagStored) entry[rank[i]] = (f & kFlagMasked) ? kLiveBreak : by_key[i];
}

void lz77_live_index(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which) {
  const uint32_t n = P.total_bytes;
  hipLaunchKernelGGL(k_live_gather, dim3((n + 256) / 256), dim3(256), 0, BR_STREAM, B.by_key, (const uint8_t*)B.flags[which], n, B.fbits, L.rank[which]);
  exclusive_scan_u32(L.rank[which], n + 1, (uint32_t*)B.sort_tmp);
  if (n) hipLaunchKernelGGL(k_live_compact, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, B.by_key, (const uint8_t*)B.fbits, (const uint32_t*)L.rank[which], n, L.entry[which]);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_live_materialise(LiveIndex ix, const uint32_t* __restrict__ first, const uint32_t* __restrict__ start,
                                                           uint32_t span_blocks, uint32_t bucket_bits, uint32_t block_bits, uint16_t* __restrict__ num,
                                                           uint32_t* __restrict__ buckets) {
  const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t K = (size_t)1 << bucket_bits;
  if (key >= K) return;
  const size_t t = first[blockIdx.y] / span_blocks;
  br_live_materialise_key(ix, key, start[blockIdx.y], block_bits, num + t * K, buckets + ((t * K) << block_bits));
}

void lz77_live_materialise(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which, const uint32_t* first_dev,
                           const uint32_t* start_dev, uint32_t count) {
  if (count == 0) return;
  const uint32_t K = 1u << P.bucket_bits;
  for (uint32_t done = 0; done < count; done += 32768) {  // (grid.y is limited to 65535)
    const uint32_t part = count - done < 32768u ? count - done : 32768u;
    hipLaunchKernelGGL(k_live_materialise, dim3((K + 255) / 256, part), dim3(256), 0, BR_STREAM, live_index_of(P, B, L, which), first_dev + done,
                       start_dev + done, L.span_blocks, P.bucket_bits, P.block_bits, L.num, L.buckets);
  }
  HIP_CHECK(hipGetLastError());
}

struct LiveParseArgs {
  Lz77Params P;
  ChainTables T;
  const Segment* segments;
  SegEntry* entries;
  SegExit* exits;
  const uint32_t* first;
  uint32_t count, span_blocks;
};
// (a launch has at most a few thousand of these chains, each of them bound by the latency of its own dependent loads:
// registers matter more than waves per SIMD)
template <bool kRows, bool kDeep = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_parse_live(LiveParseArgs a) {
  __shared__ ChainScratchT<false, kRows, kDeep> scratch;
  __shared__ uint32_t histo[256];
  const uint32_t item = blockIdx.x;
  if (item >= a.count) return;
  const uint32_t first = a.first[item];
  const uint32_t table = first / a.span_blocks;
  uint32_t last = (table + 1) * a.span_blocks;
  if (last > a.P.num_segments) last = a.P.num_segments;
  br_parse_live<kRows>(a.P, a.T, scratch, a.segments, a.entries, a.exits, first, last, table, histo);
}

void lz77_live_parse(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which, const uint32_t* first_dev, uint32_t count) {
  if (count == 0) return;
  LiveParseArgs a;
  a.P = P;
  a.T = live_chain_tables(P, B, L, which ^ 1);
  a.segments = B.segments;
  a.entries = B.entries;
  a.exits = B.exits;
  a.first = first_dev;
  a.count = count;
  a.span_blocks = L.span_blocks;
  if ((1u << P.block_bits) <= kRowEntries) {
    hipLaunchKernelGGL((k_parse_live<true>), dim3(count), dim3(64), 0, BR_STREAM, a);
  } else if (P.block_bits > 7) {
    hipLaunchKernelGGL((k_parse_live<false, true>), dim3(count), dim3(64), 0, BR_STREAM, a);
  } else {
    hipLaunchKernelGGL((k_parse_live<false>), dim3(count), dim3(64), 0, BR_STREAM, a);
  }
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_live_changed_keys(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ next, uint32_t n,
                                                            const uint16_t* __restrict__ keys, uint8_t* __restrict__ changed_key) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n && ((prev[q] ^ next[q]) & (kFlagStored | kFlagMasked))) changed_key[keys[q]] = 1;
}
// the searched positions that have to be looked at again
__global__ __launch_bounds__(256) void k_live_list(const uint8_t* __restrict__ flags, uint32_t n, uint32_t from, const uint16_t* __restrict__ keys,
                                                    const uint8_t* __restrict__ changed_key, const uint8_t* __restrict__ reparsed, uint32_t prefix_bytes,
                                                    uint32_t block_bytes, uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t cap) {
  const uint32_t p = from + blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false;
  if (p < n && (flags[p] & kFlagSearched)) take = changed_key == nullptr || reparsed[(p - prefix_bytes) / block_bytes] != 0 || changed_key[keys[p]] != 0;
  const unsigned long long m = __ballot(take);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl(base, (int)__ffsll((long long)m) - 1, 64);
  if (take) {
    const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (at < cap) list[at] = p;
  }
}

struct LiveVerifyArgs {
  Lz77Params P;
  ChainTables T;
  LiveIndex ix;
  const uint32_t* list;
  const uint32_t* count;
  uint32_t cap;
  uint32_t prefix_bytes, block_bytes;
  uint8_t* dirty;
};
template <bool kRows, bool kDeep = false>
__global__ __launch_bounds__(64) void k_live_verify(LiveVerifyArgs a) {
  __shared__ ChainScratchT<false, kRows, kDeep> scratch;
  uint32_t n = *a.count;
  if (n > a.cap) n = a.cap;
  for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
    const uint32_t p = a.list[item];
    const uint32_t blk = (p - a.prefix_bytes) / a.block_bytes;
    if (*(volatile uint8_t*)(a.dirty + blk)) continue;  // (already owed a re-parse)
    const uint64_t end64 = (uint64_t)a.prefix_bytes + (uint64_t)(blk + 1) * a.block_bytes;
    const uint32_t blk_end = end64 < a.P.total_bytes ? (uint32_t)end64 : a.P.total_bytes;
    const bool same = br_verify_search<kRows>(a.P, a.T, a.ix, scratch, p, blk_end);
    if (!same && threadIdx.x == 0) a.dirty[blk] = 1;
  }
}

void lz77_live_verify(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int prev, int next, const SegGeometry& geo,
                      const uint8_t* reparsed_dev, uint8_t* dirty_dev) {
  const uint32_t n = P.total_bytes;
  if (n <= geo.first_block_start) return;
  if (prev >= 0) {
    dev_memset(L.changed_key, 0, 65536);
    hipLaunchKernelGGL(k_live_changed_keys, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, (const uint8_t*)B.flags[prev], (const uint8_t*)B.flags[next], n,
                       (const uint16_t*)B.keys, L.changed_key);
  }
  dev_memset(B.recheck_count, 0, 4);
  const uint32_t span = n - geo.first_block_start;
  hipLaunchKernelGGL(k_live_list, dim3((span + 255) / 256), dim3(256), 0, BR_STREAM, (const uint8_t*)B.flags[next], n, geo.first_block_start,
                     (const uint16_t*)B.keys, prev >= 0 ? (const uint8_t*)L.changed_key : (const uint8_t*)nullptr, reparsed_dev, geo.prefix_bytes,
                     geo.block_bytes, B.recheck_list, B.recheck_count, B.recheck_cap);
  LiveVerifyArgs a;
  a.P = P;
  a.T = live_chain_tables(P, B, L, next);
  a.ix = live_index_of(P, B, L, next);
  a.list = B.recheck_list;
  a.count = B.recheck_count;
  a.cap = B.recheck_cap;
  a.prefix_bytes = geo.prefix_bytes;
  a.block_bytes = geo.block_bytes;
  a.dirty = dirty_dev;
  const uint32_t grid = B.recheck_cap < 32768u ? B.recheck_cap : 32768u;
  if ((1u << P.block_bits) <= kRowEntries) {
    hipLaunchKernelGGL((k_live_verify<true>), dim3(grid), dim3(64), 0, BR_STREAM, a);
  } else if (P.block_bits > 7) {
    hipLaunchKernelGGL((k_live_verify<false, true>), dim3(grid), dim3(64), 0, BR_STREAM, a);
  } else {
    hipLaunchKernelGGL((k_live_verify<false>), dim3(grid), dim3(64), 0, BR_STREAM, a);
  }
  HIP_CHECK(hipGetLastError());
}


// ------------------------------------------------------------------------------------------ qualities 10 / 11 (zopfli_device.h)
static ZopfliParams zopfli_params_of(const Lz77Params& P, const ZopfliJob& J) {
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
static ZopfliBuffers zopfli_buffers_of(const ZopfliJob& J) {
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
static ZopfliTables zopfli_tables() {
  const DeviceTables& dt = dev_tables();
  ZopfliTables T;
  T.lut_buckets = dt.dict_lut_buckets;
  T.lut_words = dt.dict_lut_words;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  T.dict_size_bits_by_length = dt.dict_size_bits_by_length;
  T.logs.logs_16 = dt.logs_16;
  T.logs.logs_8 = dt.logs_8;
  return T;
}
__global__ __launch_bounds__(256) void k_zopfli_fill(uint32_t* __restrict__ p, size_t n, uint32_t value) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = value;
}
void lz77_zopfli_init(const ZopfliJob& J) {
  const uint32_t window_mask = (1u << J.lgwin) - 1u;
  hipLaunchKernelGGL(k_zopfli_fill, dim3(256), dim3(256), 0, BR_STREAM, J.buckets, (size_t)1 << kZBucketBits, 0u - window_mask);
  hipLaunchKernelGGL(k_zopfli_fill, dim3(4096), dim3(256), 0, BR_STREAM, J.forest, (size_t)2 << J.lgwin, 0u);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_zopfli_import(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n, uint32_t delta, uint32_t invalid) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t e = src[i];
    dst[i] = (e != invalid && e >= delta) ? e - delta : invalid;
  }
}
void lz77_zopfli_import(const ZopfliJob& J, const uint32_t* buckets_src, const uint32_t* forest_src, uint32_t delta) {
  const uint32_t invalid = 0u - ((1u << J.lgwin) - 1u);
  hipLaunchKernelGGL(k_zopfli_import, dim3(256), dim3(256), 0, BR_STREAM, J.buckets, buckets_src, (size_t)1 << kZBucketBits, delta, invalid);
  hipLaunchKernelGGL(k_zopfli_import, dim3(4096), dim3(256), 0, BR_STREAM, J.forest, forest_src, (size_t)2 << J.lgwin, delta, invalid);
  HIP_CHECK(hipGetLastError());
}
// (first device slice: one lane per stream, see the header of zopfli_device.h)
__global__ __launch_bounds__(64) void k_zopfli_prepend(ZopfliParams Z, ZopfliBuffers ZB, const uint8_t* __restrict__ text, uint32_t dict_bytes) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const ZH10 h = z_hasher_of(Z, ZB);
  const uint32_t overlap = kZMaxTreeCompLength - 1;  // StoreLookahead() - 1
  for (uint32_t i = 0; i + overlap < dict_bytes; ++i) z_h10_store(h, Z, text, i);
}
void lz77_zopfli_prepend(const Lz77Params& P, const Lz77Buffers& B, const ZopfliJob& J, uint32_t dict_bytes) {
  hipLaunchKernelGGL(k_zopfli_prepend, dim3(1), dim3(64), 0, BR_STREAM, zopfli_params_of(P, J), zopfli_buffers_of(J), (const uint8_t*)B.text, dict_bytes);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(64) void k_zopfli_begin(ZopfliParams Z, ZopfliBuffers ZB, const uint8_t* __restrict__ text, const Segment* __restrict__ segments,
                                                     const SegEntry* __restrict__ entries, uint32_t block, ZBlockCtl* __restrict__ ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  br_zopfli_begin(Z, ZB, text, segments[block], entries[block], ctl);
}
// one lane per group of hash keys (the positions of a 16-bit key of the sort = two H10 keys): the groups' trees grow side by side
__global__ __launch_bounds__(64) void k_zopfli_matches(ZopfliParams Z, ZopfliTables T, ZopfliBuffers ZB, uint32_t* __restrict__ forest_new,
                                                       uint8_t* __restrict__ rerooted, const uint8_t* __restrict__ text, const uint32_t* __restrict__ by_key,
                                                       const uint32_t* __restrict__ key_first, const uint32_t* __restrict__ key_last, ZBlockCtl* ctl) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 65536u) return;
  const uint32_t lo = key_first[g], hi = key_last[g];
  if (lo >= hi) return;
  br_zopfli_matches_of_group(Z, T, ZB, forest_new, rerooted, text, by_key, lo, hi, ctl);
}
__global__ __launch_bounds__(256) void k_zopfli_merge(ZopfliParams Z, ZopfliBuffers ZB, const uint32_t* __restrict__ forest_new, const uint8_t* __restrict__ rerooted,
                                                      const ZBlockCtl* __restrict__ ctl, uint32_t block_bytes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < block_bytes) br_zopfli_merge_node(Z, ZB, forest_new, rerooted, ctl, i);
}
// The parse itself: one wavefront per stream.  All lanes run the (sequential) dynamic programme with identical scalar state; the
// lanes differ where a position's candidates and the lengths of a copy are spread over them (z_update_nodes).
__global__ __launch_bounds__(64) void k_zopfli_parse(ZopfliParams Z, ZopfliTables T, ZopfliBuffers ZB, const uint8_t* __restrict__ text,
                                                     const Segment* __restrict__ segments, const SegEntry* __restrict__ entries, Command* __restrict__ cmds,
                                                     SegExit* __restrict__ exits, uint32_t block, ZBlockCtl* ctl, uint32_t precomputed) {
  if (blockIdx.x != 0) return;
  __shared__ ZNode window[kZWin];   // (ZNodeView)
  __shared__ float lc_window[kZWin];
  __shared__ float cost_cmd[704], cost_dist[1200];  // (ZCostModel: every length of every candidate looks one of each up)
  __shared__ ZQueue queue;          // (indexed dynamically: in registers it would live in scratch memory)
  ZFast fast;
  fast.window = window;
  fast.lc_window = lc_window;
  fast.cost_cmd = cost_cmd;
  fast.cost_dist = Z.dist_alphabet_size <= 1136 ? cost_dist : (float*)nullptr;
  fast.queue = &queue;
  const Segment seg = segments[block];
  ctl->pad[0] = br_zopfli_parse(Z, T, ZB, text, seg, entries[block], ctl, precomputed != 0, cmds + seg.cmd_base, exits + block, fast);
}
bool lz77_zopfli_block(const Lz77Params& P, const Lz77Buffers& B, const ZopfliJob& J, uint32_t block) {
  static const bool sequential_only = getenv("BROTLI_MI355X_ZOPFLI_SEQUENTIAL") != nullptr;
  const ZopfliParams Z = zopfli_params_of(P, J);
  const ZopfliBuffers ZB = zopfli_buffers_of(J);
  const ZopfliTables T = zopfli_tables();
  ZBlockCtl* ctl = (ZBlockCtl*)J.ctl;
  const size_t forest_bytes = ((size_t)2 << J.lgwin) * 4, bucket_bytes = ((size_t)1 << kZBucketBits) * 4;
  hipLaunchKernelGGL(k_zopfli_begin, dim3(1), dim3(64), 0, BR_STREAM, Z, ZB, (const uint8_t*)B.text, (const Segment*)B.segments, (const SegEntry*)B.entries, block, ctl);
  bool redo = sequential_only;
  if (!sequential_only) {
    HIP_CHECK(hipMemcpyAsync(J.forest_bak, J.forest, forest_bytes, hipMemcpyDeviceToDevice, BR_STREAM));
    HIP_CHECK(hipMemcpyAsync(J.buckets_bak, J.buckets, bucket_bytes, hipMemcpyDeviceToDevice, BR_STREAM));
    hipLaunchKernelGGL(k_zopfli_matches, dim3(65536 / 64), dim3(64), 0, BR_STREAM, Z, T, ZB, J.forest_new, J.rerooted, (const uint8_t*)B.text, (const uint32_t*)B.by_key,
                       (const uint32_t*)B.key_first, (const uint32_t*)B.key_last, ctl);
    hipLaunchKernelGGL(k_zopfli_merge, dim3((J.block_bytes + 255) / 256), dim3(256), 0, BR_STREAM, Z, ZB, (const uint32_t*)J.forest_new, (const uint8_t*)J.rerooted,
                       (const ZBlockCtl*)ctl, J.block_bytes);
    hipLaunchKernelGGL(k_zopfli_parse, dim3(1), dim3(64), 0, BR_STREAM, Z, T, ZB, (const uint8_t*)B.text, (const Segment*)B.segments, (const SegEntry*)B.entries, B.cmds,
                       B.exits, block, ctl, 1u);
    HIP_CHECK(hipGetLastError());
    ZBlockCtl host;
    dev_d2h(&host, ctl, sizeof(host));
    redo = host.pad[0] == kZopfliRedo;
    if (redo) {  // a match beyond the Zopfli length: positions are skipped there, the trees differ -- back to the block's start
      HIP_CHECK(hipMemcpyAsync(J.forest, J.forest_bak, forest_bytes, hipMemcpyDeviceToDevice, BR_STREAM));
      HIP_CHECK(hipMemcpyAsync(J.buckets, J.buckets_bak, bucket_bytes, hipMemcpyDeviceToDevice, BR_STREAM));
    }
  }
  if (redo)
    hipLaunchKernelGGL(k_zopfli_parse, dim3(1), dim3(64), 0, BR_STREAM, Z, T, ZB, (const uint8_t*)B.text, (const Segment*)B.segments, (const SegEntry*)B.entries, B.cmds,
                       B.exits, block, ctl, 0u);
  HIP_CHECK(hipGetLastError());
  if (getenv("BROTLI_MI355X_DEBUG")) {  // (s_memtime counts at 100 MHz on gfx950)
    ZBlockCtl host;
    dev_d2h(&host, ctl, sizeof(host));
    fprintf(stderr, "zopfli block %u (%s): cost model %.1f ms, programme %.1f ms, commands %.1f ms, sequential matching %.1f ms\n", block, redo ? "sequential" : "matches side by side",
            host.ticks[0] / 1e5, host.ticks[1] / 1e5, host.ticks[2] / 1e5, host.ticks[3] / 1e5);
  }
  return redo;
}

}  // namespace brotli_mi355x
