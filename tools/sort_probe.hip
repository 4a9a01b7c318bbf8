// tools/sort_probe.hip -- (measured in round 6: 0.99 / 0.98 / 0.97 ms for the three variants below on 64 Mi pairs; the library's own sort carries the
// 16-bit tags as a third column, which a pair sort would need a pass in front and behind for: no gain)
// how long rocprim::radix_sort_pairs takes on the key sort of the headline step (67 M pairs of a 16-bit hash key and
// a 32-bit position; the library's own two-pass tile sort: k_radix_hist + k_radix_scatter, 1.45 ms per step).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint16_t* k, uint32_t* v, uint64_t* v64, uint32_t n) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t x = i * 2654435761u; x ^= x >> 15; x *= 0x85ebca6bu; x ^= x >> 13;
  k[i] = (uint16_t)(x >> 17);  // 15-bit keys, as bucket_bits 15
  v[i] = i;
  v64[i] = ((uint64_t)(x & 0xffff) << 32) | i;
}
int main() {
  const uint32_t n = 64u << 20;
  uint16_t *k0, *k1; uint32_t *v0, *v1; uint64_t *w0, *w1;
  CK(hipMalloc(&k0, n * 2)); CK(hipMalloc(&k1, n * 2)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4)); CK(hipMalloc(&w0, (size_t)n * 8)); CK(hipMalloc(&w1, (size_t)n * 8));
  fill<<<(n + 255) / 256, 256>>>(k0, v0, w0, n);
  size_t tmp_bytes = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, w0, w1, (size_t)n, 0u, 16u, 0));
  void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(a, 0));
      if (variant == 0) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, 16u, 0));
      if (variant == 1) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, 15u, 0));
      if (variant == 2) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, w0, w1, (size_t)n, 0u, 16u, 0));
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 3) printf("%s: %.3f ms (temporary storage %.1f MiB)\n", variant == 0 ? "u16 key / u32 value, 16 bits" : variant == 1 ? "u16 key / u32 value, 15 bits" : "u16 key / u64 value (position + tag), 16 bits", ms, tmp_bytes / 1048576.0);
    }
  }
  return 0;
}
