// device_scan.h -- exclusive prefix sum over uint32 arrays (hierarchical, 1024 elements per workgroup),
// shared by the LZ77 and meta-block kernels.  Include from .hip files only.
#ifndef BROTLI_MI355X_DEVICE_SCAN_H_
#define BROTLI_MI355X_DEVICE_SCAN_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace brotli_mi355x {

void hip_check(hipError_t e, const char* what);
#define HIP_CHECK(x) hip_check((x), #x)

// ------------------------------------------------------------------------------------------ scan
// exclusive prefix sum of a uint32 array (in place), hierarchical: 1024 elements per workgroup
static constexpr uint32_t kScanTile = 1024;
static inline size_t scan_scratch_words(size_t n) { return n / kScanTile + n / (kScanTile * kScanTile) + 4096; }

static __global__ __launch_bounds__(256) void k_scan_tiles(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (base + j < n) ? data[base + j] : 0;
  const uint32_t local = v[0] + v[1] + v[2] + v[3];
  // inclusive scan across the wavefront (64 lanes) with DPP-free shuffles
  uint32_t x = local;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_sum[w] = x;
  __syncthreads();
  uint32_t wave_off = 0;
  for (int i = 0; i < w; ++i) wave_off += wave_sum[i];
  uint32_t excl = wave_off + x - local;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) data[base + j] = excl;
    excl += v[j];
  }
  if (threadIdx.x == 255 && tile_sums) tile_sums[blockIdx.x] = wave_off + x;
}

static __global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ tile_offsets) {
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
  const uint32_t add = tile_offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) data[base + j] += add;
}

// scratch must hold ceil(n/1024) + ceil(n/1024^2) + ... + 2 uint32
static void exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* scratch) {
  if (n == 0) return;
  const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, BR_STREAM, data, n, tiles > 1 ? scratch : (uint32_t*)nullptr);
  if (tiles > 1) {
    exclusive_scan_u32(scratch, tiles, scratch + tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(256), 0, BR_STREAM, data, n, scratch);
  }
}


}  // namespace brotli_mi355x
#endif
