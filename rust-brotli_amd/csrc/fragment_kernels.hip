// fragment_kernels.hip -- gfx950 kernels of qualities 0 and 1 (fragment_api.h, fragment_device.h): one wavefront per fragment of a
// stream, code tables / histograms / Huffman nodes in workgroup memory.
#include <hip/hip_runtime.h>

#include "fragment_device.h"
#include "device_api.h"
#include "device_scan.h"

namespace brotli_mi355x {

__global__ __launch_bounds__(256) void k_fragment_clear(uint32_t* __restrict__ p, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

__global__ __launch_bounds__(64) void k_fragment(int quality, EntropyTables et, const uint8_t* __restrict__ input, uint32_t input_size, uint32_t is_last, uint32_t table_bits,
                                                 FragmentBuffers B, uint8_t* __restrict__ out) {
  if (blockIdx.x != 0) return;
  __shared__ FragmentScratch S;
  __shared__ uint64_t cmd_code_words[kTreeBitsWords];
  br_fragment(quality, et, input, input_size, is_last != 0, table_bits, B, out, S, cmd_code_words);
}

void frag_compress(int quality, const uint8_t* input, uint32_t input_size, bool is_last, uint32_t table_bits, const FragmentBuffers& B, uint8_t* out) {
  const DeviceTables& dt = dev_tables();
  EntropyTables et;
  et.logs_16 = dt.logs_16;
  et.logs_8 = dt.logs_8;
  const uint32_t words = 1u << table_bits;
  hipLaunchKernelGGL(k_fragment_clear, dim3((words + 255) / 256 < 64 ? (words + 255) / 256 : 64), dim3(256), 0, BR_STREAM, B.table, words);
  hipLaunchKernelGGL(k_fragment, dim3(1), dim3(64), 0, BR_STREAM, quality, et, input, input_size, is_last ? 1u : 0u, table_bits, B, out);
  HIP_CHECK(hipGetLastError());
}

}  // namespace brotli_mi355x
