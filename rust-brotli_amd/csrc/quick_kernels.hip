// quick_kernels.hip -- gfx950 kernels of qualities 2..4 (quick_api.h, quick_device.h): the BasicHasher family under the
// greedy / lazy parse, one wavefront per stream walking the reference's own hash table block by block.
#include <hip/hip_runtime.h>

#include "quick_device.h"
#include "device_scan.h"

namespace brotli_mi355x {

static QuickTables quick_tables() {
  const DeviceTables& dt = dev_tables();
  QuickTables T;
  T.dict_hash = dt.dict_hash;
  T.dict_data = dt.dict_data;
  T.dict_offsets_by_length = dt.dict_offsets_by_length;
  return T;
}

__global__ __launch_bounds__(256) void k_quick_fill(uint32_t* __restrict__ p, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}
void lz77_quick_init(const QuickJob& J) {
  hipLaunchKernelGGL(k_quick_fill, dim3(256), dim3(256), 0, BR_STREAM, J.table, quick_table_words(J));
  HIP_CHECK(hipGetLastError());
}

// slots: positions moved down by delta (0 where they fall in front of the new text); books: as they are
__global__ __launch_bounds__(256) void k_quick_import(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t slots, uint32_t words, uint32_t delta) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) {
    const uint32_t v = src[i];
    dst[i] = i < slots ? (v >= delta ? v - delta : 0u) : v;
  }
}
void lz77_quick_import(const QuickJob& J, const uint32_t* table_src, uint32_t delta) {
  hipLaunchKernelGGL(k_quick_import, dim3(256), dim3(256), 0, BR_STREAM, J.table, table_src, quick_slots(J), quick_table_words(J), delta);
  HIP_CHECK(hipGetLastError());
}

// HasherPrependCustomDictionary files the dictionary positions in ascending order into a table that holds nothing else yet
// (lz77_quick_init comes first): what a slot ends up with is the LARGEST position filed under it -- one thread per position and
// an atomic maximum give the table of the sequential loop (br_quick_prepend, which the emulation runs).  A shard of
// BrotliEncoderCompressMulti is primed with up to a window of text in front of it, several times its own size.
__global__ __launch_bounds__(256) void k_quick_prepend(QuickJob J, const uint8_t* __restrict__ text, uint32_t count) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint64_t v = (br_load64(text + i) << (64u - 8u * J.hash_len)) * kQuickHashMul64;
    const uint32_t key = (uint32_t)(v >> (64u - J.bucket_bits));
    atomicMax(J.table + key + ((i >> 3) & (J.sweep - 1u)), i);
  }
}
void lz77_quick_prepend(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, uint32_t dict_bytes) {
  if (dict_bytes <= kQuickHtl - 1u) return;
  const uint32_t count = dict_bytes - (kQuickHtl - 1u);
  const uint32_t blocks = (count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096;
  hipLaunchKernelGGL(k_quick_prepend, dim3(blocks), dim3(256), 0, BR_STREAM, J, (const uint8_t*)B.text, count);
  HIP_CHECK(hipGetLastError());
}

// One wavefront per stream: all lanes run the sequential parse with identical scalar state (lane 0 writes).
__global__ __launch_bounds__(64) void k_quick_block(QuickJob J, Lz77Params P, QuickTables T, const uint8_t* __restrict__ text, const Segment* __restrict__ segments,
                                                    const SegEntry* __restrict__ entries, Command* __restrict__ cmds, SegExit* __restrict__ exits, uint32_t block) {
  if (blockIdx.x != 0) return;
  const Segment seg = segments[block];
  br_quick_block(J, P, T, text, seg, entries[block], cmds + seg.cmd_base, exits + block);
}
void lz77_quick_block(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, uint32_t block) {
  hipLaunchKernelGGL(k_quick_block, dim3(1), dim3(64), 0, BR_STREAM, J, P, quick_tables(), (const uint8_t*)B.text, (const Segment*)B.segments, (const SegEntry*)B.entries, B.cmds,
                     B.exits, block);
  HIP_CHECK(hipGetLastError());
}

}  // namespace brotli_mi355x
