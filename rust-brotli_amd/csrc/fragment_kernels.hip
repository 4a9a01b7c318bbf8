// fragment_kernels.hip -- gfx950 kernels of qualities 0 and 1 (fragment_api.h, fragment_device.h): one wavefront per fragment,
// all fragments of a batch side by side (grid = number of fragments), code tables / histograms / Huffman nodes in workgroup
// memory; and the join of the fragments' slots into one stream.
#include <hip/hip_runtime.h>

#include "fragment_device.h"
#include "device_api.h"
#include "device_scan.h"

namespace brotli_mi355x {

// the hash tables of the fragments of a batch: fragment j's words [0, 1 << table_bits) of slab j
__global__ __launch_bounds__(256) void k_fragment_clear(uint32_t* __restrict__ tables, size_t stride, const FragmentJob* __restrict__ jobs) {
  const FragmentJob job = jobs[blockIdx.y];
  uint4* p = (uint4*)(tables + (size_t)blockIdx.y * stride);
  const uint32_t n = (1u << job.table_bits) / 4u;  // (table_bits >= 8)
  const uint4 zero = {0u, 0u, 0u, 0u};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = zero;
}

__global__ __launch_bounds__(64) void k_fragment(int quality, EntropyTables et, const uint8_t* __restrict__ input, const FragmentJob* __restrict__ jobs,
                                                 FragmentBuffers B, const FragmentState* __restrict__ states_in, FragmentState* __restrict__ states_out,
                                                 FragmentResult* __restrict__ results, uint8_t* __restrict__ out) {
  __shared__ FragmentScratch S;
  __shared__ uint64_t cmd_code_words[kTreeBitsWords];
  const uint32_t j = blockIdx.x;
  FragmentJob job = jobs[j];
  job.in_offset = BR_UNIFORM(job.in_offset);
  job.in_size = BR_UNIFORM(job.in_size);
  job.is_last = BR_UNIFORM(job.is_last);
  job.table_bits = BR_UNIFORM(job.table_bits);
  job.start_bits = BR_UNIFORM(job.start_bits);
  job.state_in = BR_UNIFORM(job.state_in);
  br_fragment_job(quality, et, input, job, j, B, states_in, states_out, results, out, S, cmd_code_words);
}

void frag_compress_batch(int quality, const uint8_t* input, const FragmentJob* jobs_dev, uint32_t n, const FragmentBuffers& B,
                         const FragmentState* states_in_dev, FragmentState* states_out_dev, FragmentResult* results_dev, uint8_t* out) {
  if (n == 0) return;
  const DeviceTables& dt = dev_tables();
  EntropyTables et;
  et.logs_16 = dt.logs_16;
  et.logs_8 = dt.logs_8;
  hipLaunchKernelGGL(k_fragment_clear, dim3(32, n), dim3(256), 0, BR_STREAM, B.table, B.table_stride, jobs_dev);
  hipLaunchKernelGGL(k_fragment, dim3(n), dim3(64), 0, BR_STREAM, quality, et, input, jobs_dev, B, states_in_dev, states_out_dev, results_dev, out);
  HIP_CHECK(hipGetLastError());
}

// 64 bits of a piece per thread: read across two source words, OR into (at most) two destination words.  Pieces are disjoint in
// dst, so only their first and last words are shared -- every word goes through an atomic OR all the same (dst starts zeroed).
__global__ __launch_bounds__(256) void k_fragment_join(const uint64_t* __restrict__ src, const FragmentPiece* __restrict__ pieces, unsigned long long* __restrict__ dst) {
  const FragmentPiece pc = pieces[blockIdx.y];
  const uint64_t chunks = (pc.nbits + 63) >> 6;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = pc.src_bit + (c << 6);
    const uint64_t left = pc.nbits - (c << 6);
    const uint32_t sh = (uint32_t)(s & 63u);
    uint64_t v = src[s >> 6] >> sh;
    if (sh != 0 && left > 64u - sh) v |= src[(s >> 6) + 1] << (64u - sh);
    if (left < 64) v &= (1ull << left) - 1ull;
    if (v == 0) continue;
    const uint64_t d = pc.dst_bit + (c << 6);
    const uint32_t dh = (uint32_t)(d & 63u);
    atomicOr(dst + (d >> 6), (unsigned long long)(v << dh));
    if (dh != 0 && (v >> (64u - dh)) != 0) atomicOr(dst + (d >> 6) + 1, (unsigned long long)(v >> (64u - dh)));
  }
}

void frag_join(const uint8_t* src, const FragmentPiece* pieces_dev, uint32_t n, uint8_t* dst) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fragment_join, dim3(64, n), dim3(256), 0, BR_STREAM, (const uint64_t*)src, pieces_dev, (unsigned long long*)dst);
  HIP_CHECK(hipGetLastError());
}

}  // namespace brotli_mi355x
