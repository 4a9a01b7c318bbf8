// metablock_kernels.hip -- gfx950 kernels of the meta-block stage: per-granule histograms, greedy block
// splitting (one workgroup per splitter, strictly ordered f32 entropy sums), Huffman code construction
// (one thread per histogram), header serialisation (one thread per meta-block) and the parallel symbol
// emission (bit length per symbol -> prefix sums -> 64-bit atomic OR scatter).  Integer/byte work, no MFMA.
#include <hip/hip_runtime.h>

#include "device_api.h"
#include "device_scan.h"
#include "metablock_api.h"
#include "metablock_hq.h"

namespace brotli_mi355x {

size_t mb_scan_scratch_bytes(size_t n) { return scan_scratch_words(n) * 4 + 256; }

template <typename F>
__global__ __launch_bounds__(256) void k_for_each(uint32_t n, F f) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f(i);
}
template <typename F>
static void for_each(uint32_t n, F f) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_for_each<F>, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, n, f);
}

// out[m] = src[descs[m].cmd_offset] for every meta-block, out[n_mb] = src[n_cmds]: the per-meta-block boundaries of an
// exclusive scan over the commands, fetched with one copy instead of one per meta-block
void mb_gather_at_metablock_starts(const MbBuffers& B, const uint32_t* src, uint32_t* out_dev) {
  const MbBuffers b = B;
  for_each(b.n_mb + 1, [b, src, out_dev] __device__(uint32_t m) { out_dev[m] = src[m < b.n_mb ? b.descs[m].cmd_offset : b.n_cmds]; });
  HIP_CHECK(hipGetLastError());
}

void mb_gather_bytes(const uint8_t* text, const uint32_t* positions_dev, uint32_t n, uint8_t* out_dev) {
  for_each(n, [text, positions_dev, out_dev] __device__(uint32_t i) {
    const uint32_t p = positions_dev[i];
    out_dev[i] = p == 0xffffffffu ? (uint8_t)0 : text[p];
  });
  HIP_CHECK(hipGetLastError());
}

void mb_command_scans(const MbBuffers& B, void* scan_scratch) {
  const MbBuffers b = B;
  for_each(b.n_cmds, [b] __device__(uint32_t c) { mb_item_command_counts(b, c); });
  // element [K] = 0 so that the exclusive scan leaves the totals there
  dev_memset(b.cmd_lit_start + b.n_cmds, 0, 4);
  dev_memset(b.cmd_pos + b.n_cmds, 0, 4);
  dev_memset(b.cmd_dist_index + b.n_cmds, 0, 4);
  exclusive_scan_u32(b.cmd_lit_start, b.n_cmds + 1, (uint32_t*)scan_scratch);
  exclusive_scan_u32(b.cmd_pos, b.n_cmds + 1, (uint32_t*)scan_scratch);
  exclusive_scan_u32(b.cmd_dist_index, b.n_cmds + 1, (uint32_t*)scan_scratch);
  HIP_CHECK(hipGetLastError());
}

void mb_literal_map(const MbBuffers& B) {
  const MbBuffers b = B;
  for_each(b.n_lits, [b] __device__(uint32_t i) { mb_item_literal_map(b, i); });
  HIP_CHECK(hipGetLastError());
}

// sampled context statistics, one workgroup per meta-block (encode.rs:1802-1927)
__global__ __launch_bounds__(256) void k_context_stats(MbBuffers B, uint32_t* stats) {
  __shared__ uint32_t s[kContextStatsWords];
  const uint32_t m = blockIdx.x;
  const MbDesc d = B.descs[m];
  for (uint32_t j = threadIdx.x; j < kContextStatsWords; j += blockDim.x) s[j] = 0;
  __syncthreads();
  const uint32_t length = d.end - d.start;
  const uint32_t n_strides = length >= 64 ? (length - 64) / 4096 + 1 : 0;
  for (uint32_t t = threadIdx.x; t < n_strides; t += blockDim.x) {
    const uint32_t start_pos = d.start + t * 4096;
    const uint8_t* p = B.text + start_pos;
    // simple bigram-prefix histogram (encode.rs:1885-1918)
    {
      const int lut[4] = {0, 0, 1, 2};
      int prev = lut[p[0] >> 6] * 3;
      for (uint32_t k = 1; k < 64; ++k) {
        const uint8_t literal = p[k];
        atomicAdd(&s[prev + lut[literal >> 6]], 1u);
        prev = lut[literal >> 6] * 3;
      }
    }
    // complex context statistics (encode.rs:1820-1842)
    {
      uint8_t prev2 = p[0], prev1 = p[1];
      for (uint32_t k = 2; k < 64; ++k) {
        const uint8_t literal = p[k];
        const uint32_t context = br_static_context_map(3, br_context(B.utf8_lut, B.signed_lut, prev1, prev2, 2));
        atomicAdd(&s[480], 1u);
        atomicAdd(&s[16 + (literal >> 3)], 1u);
        atomicAdd(&s[48 + context * 32 + (literal >> 3)], 1u);
        prev2 = prev1;
        prev1 = literal;
      }
    }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < kContextStatsWords; j += blockDim.x) stats[(size_t)m * kContextStatsWords + j] = s[j];
}

void mb_context_stats(const MbBuffers& B, uint32_t* stats_dev) {
  if (B.n_mb == 0) return;
  hipLaunchKernelGGL(k_context_stats, dim3(B.n_mb), dim3(256), 0, BR_STREAM, B, stats_dev);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_granule_histograms(MbBuffers B, uint32_t kind) {
  __shared__ uint32_t lds[kMaxStaticContexts * 256];
  mb_item_granule_histogram(B, kind, blockIdx.x, lds);
}

void mb_granule_histograms(const MbBuffers& B) {
  const MbBuffers b = B;
  if (b.n_granules[kSplitLiteral]) hipLaunchKernelGGL(k_granule_histograms, dim3(b.n_granules[kSplitLiteral]), dim3(256), 0, BR_STREAM, b, (uint32_t)kSplitLiteral);
  if (b.n_granules[kSplitCommand]) hipLaunchKernelGGL(k_granule_histograms, dim3(b.n_granules[kSplitCommand]), dim3(256), 0, BR_STREAM, b, (uint32_t)kSplitCommand);
  if (b.n_granules[kSplitDistance]) {
    dev_memset(b.gran_hist[kSplitDistance], 0, (size_t)b.n_granules[kSplitDistance] * kNumDistanceHistoSymbols * 2);
    for_each(b.n_cmds, [b] __device__(uint32_t c) { mb_item_distance_count(b, c); });
  }
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(1024) void k_split_chains(MbBuffers B) {
  __shared__ SplitScratch S;
  mb_item_split_chain(B, blockIdx.x / 3, blockIdx.x % 3, S);
}

// `wide`: some meta-block models its literals with the 13-context map (39 histogram rows per decision instead of 3-9):
// 16 waves per chain take them side by side; otherwise 4 waves are enough and synchronise faster.
void mb_split_chains(const MbBuffers& B, bool wide) {
  if (B.n_mb == 0) return;
  hipLaunchKernelGGL(k_split_chains, dim3(B.n_mb * 3), dim3(wide ? 1024 : 256), 0, BR_STREAM, B);
  HIP_CHECK(hipGetLastError());
}

#if defined(BR_CODES_PROFILE)
__device__ unsigned long long g_codes_prof[32];
#endif
// The Huffman construction is sequential, data-dependent control flow: 64 different histograms in the lanes of one
// wavefront would serialise on every divergent branch.  One histogram per wavefront (lane 0 works) puts the jobs
// on different SIMDs instead, where they really run concurrently.
__global__ __launch_bounds__(64) void k_build_codes(MbBuffers B, const CodeJob* jobs, uint32_t n_jobs) {
  // everything the sequential builder touches is staged in LDS (its loads/stores are dependent, so their latency is
  // what the job costs); the 64 lanes only help with the copies
  __shared__ HuffmanScratch sc;
  __shared__ uint32_t h[704];
  __shared__ uint16_t bits[704];
  __shared__ uint8_t depth[704];
  __shared__ uint64_t words[kTreeBitsWords];
  __shared__ uint32_t nbits;
  const uint32_t i = blockIdx.x;
  if (i >= n_jobs) return;
  const CodeJob j = jobs[i];
  const uint32_t row = kRowLen[j.kind];
  uint32_t* gh = B.histo[j.kind] + (size_t)j.row_index * row;
  for (uint32_t k = threadIdx.x; k < row; k += 64) h[k] = gh[k];
  __syncthreads();
  {
    const uint32_t nb = mb_build_code_core(j.kind, j.num_distance_symbols, h, depth, bits, words, &sc, true, j.mode);
    if (threadIdx.x == 0) nbits = nb;
  }
  __syncthreads();
  uint8_t* gd = B.depth[j.kind] + (size_t)j.row_index * row;
  uint16_t* gb = B.bits[j.kind] + (size_t)j.row_index * row;
  uint64_t* gw = B.tree_bits[j.kind] + (size_t)j.row_index * kTreeBitsWords;
  for (uint32_t k = threadIdx.x; k < row; k += 64) {
    gh[k] = h[k];
    gd[k] = depth[k];
    gb[k] = bits[k];
  }
  for (uint32_t k = threadIdx.x; k < kTreeBitsWords; k += 64) gw[k] = words[k];
  if (threadIdx.x == 0) B.tree_nbits[j.kind][j.row_index] = nbits;
}

void mb_build_codes(const MbBuffers& B, const CodeJob* jobs_dev, uint32_t n_jobs) {
  if (n_jobs == 0) return;
  hipLaunchKernelGGL(k_build_codes, dim3(n_jobs), dim3(64), 0, BR_STREAM, B, jobs_dev, n_jobs);
  HIP_CHECK(hipGetLastError());
#if defined(BR_CODES_PROFILE)
  {
    unsigned long long h[32];
    HIP_CHECK(hipStreamSynchronize(BR_STREAM));
    HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_codes_prof), sizeof(h)));
    fprintf(stderr, "codes profile (%u jobs), cycles sum / slowest job:", n_jobs);
    for (int i = 0; i < 16; ++i) fprintf(stderr, " [%d] %llu / %llu", i, h[i], h[16 + i]);
    fprintf(stderr, "\n");
    unsigned long long z[32] = {0};
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_codes_prof), z, sizeof(z)));
  }
#endif
}

// The header of a meta-block is a sequential bit string (block-split codes, context map, then the serialised trees that
// k_build_codes prepared): one lane composes it -- in LDS, where a read-modify-write of the word under the cursor costs
// a few cycles instead of a trip to L2 -- and the wave copies the finished words out.  Headers that might not fit the
// LDS buffer (hundreds of literal histograms) are composed in global memory as before.
__global__ __launch_bounds__(64) void k_write_headers(MbBuffers B) {
  constexpr uint32_t kLdsWords = 5120;            // 40 KiB
  constexpr uint64_t kOtherBits = 160u * 1024u;   // generous bound for everything but the trees (a 16 Ki-entry context map)
  __shared__ uint64_t lw[kLdsWords];
  __shared__ HuffmanScratch sc_lds;  // (the scratch of the small trees -- block-split codes, context map -- as well)
  __shared__ MbSplitPrepared prep;
  __shared__ uint32_t s_bits;
  const uint32_t m = blockIdx.x;
  if (m >= B.n_mb) return;
  const MbDesc d = B.descs[m];
  uint64_t* global_words = B.header_words + (size_t)m * B.header_stride;
  if (d.uncompressed) {
    if (threadIdx.x == 0) mb_item_write_header(B, m, B.huff_scratch + m, global_words);
    return;
  }
  uint32_t trees = 0;
  for (uint32_t kind = 0; kind < 3; ++kind)
    for (uint32_t i = threadIdx.x; i < B.results[m].num_histos[kind]; i += 64) trees += B.tree_nbits[kind][d.histo_base[kind] + i];
  for (int off = 32; off > 0; off >>= 1) trees += __shfl_down(trees, off, 64);
  trees = (uint32_t)__shfl((int)trees, 0, 64);
  const bool in_lds = (uint64_t)trees + kOtherBits <= (uint64_t)kLdsWords * 64;
  uint64_t* stage = in_lds ? lw : global_words;
  const uint32_t clear_words = in_lds ? kLdsWords : B.header_stride;
  for (uint32_t i = threadIdx.x; i < clear_words; i += 64) stage[i] = 0;
  // The block-split codes (BuildAndStoreBlockSplitCode) walk all blocks of a kind twice -- the histograms of the type and length
  // codes, then the switch command of every block: a block per lane in front of and behind the one lane that composes the header
  // (a meta-block of a mix has thousands of blocks: 4.8 ms of a 256 MiB Silesia-like call were this kernel).
  for (uint32_t i = threadIdx.x; i < 3 * (258 + 26); i += 64) (&prep.histograms[0][0])[i] = 0;
  __syncthreads();
  for (uint32_t kind = 0; kind < 3; ++kind) {
    const uint8_t* types = B.block_types[kind] + d.block_base[kind];
    const uint32_t* lengths = B.block_lengths[kind] + d.block_base[kind];
    const uint32_t nb = B.results[m].num_blocks[kind];
    for (uint32_t i = threadIdx.x; i < nb; i += 64) {
      if (i != 0) atomicAdd(&prep.histograms[kind][br_block_type_code_at(types, i)], 1u);
      atomicAdd(&prep.histograms[kind][258 + br_block_length_prefix_code(lengths[i])], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mb_item_write_header(B, m, &sc_lds, stage, &prep);
    s_bits = B.results[m].header_bits;
  }
  __syncthreads();
  for (uint32_t kind = 0; kind < 3; ++kind) {
    if (B.results[m].num_types[kind] <= 1) continue;
    const uint8_t* types = B.block_types[kind] + d.block_base[kind];
    const uint32_t* lengths = B.block_lengths[kind] + d.block_base[kind];
    uint64_t* switch_bits = B.switch_bits[kind] + d.block_base[kind];
    uint8_t* switch_nbits = B.switch_nbits[kind] + d.block_base[kind];
    const uint32_t nb = B.results[m].num_blocks[kind];
    for (uint32_t i = 1 + threadIdx.x; i < nb; i += 64) {
      uint32_t nbits;
      switch_bits[i] = br_block_switch_bits(prep.code[kind], br_block_type_code_at(types, i), lengths[i], false, &nbits);
      switch_nbits[i] = (uint8_t)nbits;
    }
  }
  if (in_lds) {
    const uint32_t words = (s_bits + 63) / 64 + 1;
    for (uint32_t i = threadIdx.x; i < words && i < kLdsWords; i += 64) global_words[i] = lw[i];
  }
}

void mb_write_headers(const MbBuffers& B) {
  if (B.n_mb == 0) return;
  hipLaunchKernelGGL(k_write_headers, dim3(B.n_mb), dim3(64), 0, BR_STREAM, B);
  HIP_CHECK(hipGetLastError());
}

void mb_symbol_bits(const MbBuffers& B, void* scan_scratch) {
  const MbBuffers b = B;
  for_each(b.n_lits, [b] __device__(uint32_t i) { mb_item_literal_nbits(b, i); });
  dev_memset(b.lit_nbits + b.n_lits, 0, 4);
  exclusive_scan_u32(b.lit_nbits, b.n_lits + 1, (uint32_t*)scan_scratch);
  for_each(b.n_cmds, [b] __device__(uint32_t c) { mb_item_command_nbits(b, c); });
  dev_memset(b.cmd_nbits + b.n_cmds, 0, 4);
  exclusive_scan_u32(b.cmd_nbits, b.n_cmds + 1, (uint32_t*)scan_scratch);
  HIP_CHECK(hipGetLastError());
}

void mb_emit(const MbBuffers& B) {
  const MbBuffers b = B;
  for_each(b.n_cmds, [b] __device__(uint32_t c) { mb_item_emit_command(b, c); });
  for_each(b.n_lits, [b] __device__(uint32_t i) { mb_item_emit_literal(b, i); });
  HIP_CHECK(hipGetLastError());
}

// ---- quality >= 10: one lane per sequential job (metablock_hq.h says why), all lanes for the per-symbol passes
__global__ __launch_bounds__(256) void k_hq_utf8_census(MbBuffers B) {
  __shared__ HqCensusScratch S;
  if (!B.descs[blockIdx.x].uncompressed) hq_item_utf8_census(B, blockIdx.x, S);
}
void mb_hq_utf8_census(const MbBuffers& B) {
  if (B.n_mb == 0) return;
  hipLaunchKernelGGL(k_hq_utf8_census, dim3(B.n_mb), dim3(256), 0, BR_STREAM, B);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(64) void k_hq_distance_params(MbBuffers B) {
  __shared__ HqWaveScratch S;
  hq_item_distance_params(B, blockIdx.x, S);
}
void mb_hq_distance_params(const MbBuffers& B) {
  if (B.n_mb == 0) return;
  hipLaunchKernelGGL(k_hq_distance_params, dim3(B.n_mb), dim3(64), 0, BR_STREAM, B);
  HIP_CHECK(hipGetLastError());
}
void mb_hq_gather_symbols(const MbBuffers& B) {
  const MbBuffers b = B;
  for_each(b.n_lits, [b] __device__(uint32_t i) { hq_item_literal_symbol(b, i); });
  for_each(b.n_cmds, [b] __device__(uint32_t c) { hq_item_command_symbols(b, c); });
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(64) void k_hq_find_blocks(EntropyTables et, HqSplitJob* jobs) {
  __shared__ HqWaveScratch S;
  hq_item_find_blocks(et, jobs[blockIdx.x], S);
}
void mb_hq_find_blocks(const MbBuffers& B, HqSplitJob* jobs_dev, uint32_t n_jobs) {
  if (n_jobs == 0) return;
  hipLaunchKernelGGL(k_hq_find_blocks, dim3(n_jobs), dim3(64), 0, BR_STREAM, B.et, jobs_dev);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(64) void k_hq_blocks_prep(const HqSplitJob* jobs) { hq_item_blocks_prep(jobs[blockIdx.x]); }
__global__ __launch_bounds__(64) void k_hq_cluster_blocks_batch(EntropyTables et, const HqSplitJob* jobs, const HqBatchRef* batches) {
  __shared__ HqWaveScratch S;
  __shared__ HqBatchPairs P;
  const HqBatchRef ref = batches[blockIdx.x];
  hq_item_cluster_blocks_batch(et, jobs[ref.job], ref.batch, S, P.pairs);
}
__global__ __launch_bounds__(64) void k_hq_cluster_blocks(MbBuffers B, const HqSplitJob* jobs) {
  __shared__ HqWaveScratch S;
  hq_item_cluster_blocks(B, jobs[blockIdx.x], S);
}
void mb_hq_cluster_blocks(const MbBuffers& B, const HqSplitJob* jobs_dev, uint32_t n_jobs, const HqBatchRef* batches_dev, uint32_t n_batches) {
  if (n_jobs == 0) return;
  hipLaunchKernelGGL(k_hq_blocks_prep, dim3(n_jobs), dim3(64), 0, BR_STREAM, jobs_dev);
  if (n_batches) hipLaunchKernelGGL(k_hq_cluster_blocks_batch, dim3(n_batches), dim3(64), 0, BR_STREAM, B.et, jobs_dev, batches_dev);
  hipLaunchKernelGGL(k_hq_cluster_blocks, dim3(n_jobs), dim3(64), 0, BR_STREAM, B, jobs_dev);
  HIP_CHECK(hipGetLastError());
}
void mb_hq_context_histograms(const MbBuffers& B) {
  const MbBuffers b = B;
  for_each(b.n_lits, [b] __device__(uint32_t i) { hq_item_literal_context_count(b, i); });
  for_each(b.n_cmds, [b] __device__(uint32_t c) { hq_item_command_context_count(b, c); });
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(64) void k_hq_cluster_histograms_batch(EntropyTables et, const HqClusterJob* jobs, const HqBatchRef* batches) {
  __shared__ HqWaveScratch S;
  __shared__ HqBatchPairs P;
  const HqBatchRef ref = batches[blockIdx.x];
  hq_item_cluster_histograms_batch(et, jobs[ref.job], ref.batch, S, P.pairs);
}
__global__ __launch_bounds__(64) void k_hq_cluster_histograms(MbBuffers B, const HqClusterJob* jobs) {
  __shared__ HqWaveScratch S;
  hq_item_cluster_histograms(B, jobs[blockIdx.x], S);
}
void mb_hq_cluster_histograms(const MbBuffers& B, const HqClusterJob* jobs_dev, uint32_t n_jobs, const HqBatchRef* batches_dev,
                              uint32_t n_batches) {
  if (n_jobs == 0) return;
  if (n_batches) hipLaunchKernelGGL(k_hq_cluster_histograms_batch, dim3(n_batches), dim3(64), 0, BR_STREAM, B.et, jobs_dev, batches_dev);
  hipLaunchKernelGGL(k_hq_cluster_histograms, dim3(n_jobs), dim3(64), 0, BR_STREAM, B, jobs_dev);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_copy_bits_batch(unsigned long long* __restrict__ out, const uint64_t* __restrict__ src_words,
                                                          const MbBitCopy* __restrict__ items) {
  const MbBitCopy it = items[blockIdx.y];
  const uint64_t* src = src_words + it.src_word;
  const uint64_t chunks = (it.nbits + 63) >> 6;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = src[c];
    const uint64_t left = it.nbits - (c << 6);
    if (left < 64) v &= (1ull << left) - 1ull;
    if (v == 0) continue;
    const uint64_t d = it.dst_bit + (c << 6);
    const uint32_t dh = (uint32_t)(d & 63u);
    atomicOr(out + (d >> 6), (unsigned long long)(v << dh));
    if (dh != 0 && (v >> (64u - dh)) != 0) atomicOr(out + (d >> 6) + 1, (unsigned long long)(v >> (64u - dh)));
  }
}
void mb_copy_bits_batch(uint64_t* out, const uint64_t* src_words, const MbBitCopy* items_dev, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_copy_bits_batch, dim3(8, n), dim3(256), 0, BR_STREAM, (unsigned long long*)out, src_words, items_dev);
  HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_place_pieces(unsigned long long* __restrict__ out, const MbBitPiece* __restrict__ pieces, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const MbBitPiece pc = pieces[i];
  uint64_t v = pc.bits;
  if (pc.nbits < 64) v &= (1ull << pc.nbits) - 1ull;
  if (v == 0) return;
  const uint32_t dh = (uint32_t)(pc.pos & 63u);
  atomicOr(out + (pc.pos >> 6), (unsigned long long)(v << dh));
  if (dh != 0 && (v >> (64u - dh)) != 0) atomicOr(out + (pc.pos >> 6) + 1, (unsigned long long)(v >> (64u - dh)));
}
void mb_place_pieces(uint64_t* out, const MbBitPiece* pieces_dev, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_place_pieces, dim3((n + 255) / 256), dim3(256), 0, BR_STREAM, (unsigned long long*)out, pieces_dev, n);
  HIP_CHECK(hipGetLastError());
}
// (byte-aligned destinations: a stored meta-block starts behind a jump to the byte boundary; 16 bytes per thread where both sides
// allow it, single bytes at the ragged ends)
__global__ __launch_bounds__(256) void k_raw_copies(uint8_t* __restrict__ out, const uint8_t* __restrict__ text, const MbRawCopy* __restrict__ items) {
  const MbRawCopy it = items[blockIdx.y];
  uint8_t* dst = out + it.dst_byte;
  const uint8_t* src = text + it.src_pos;
  // head up to the first 16-byte boundary of the destination
  uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
  if (head > it.bytes) head = it.bytes;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (uint32_t i = t; i < head; i += nt) dst[i] = src[i];
  const uint32_t body = (it.bytes - head) / 16u;
  typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
  typedef uint32_t u32x4_a __attribute__((ext_vector_type(4)));
  for (uint32_t i = t; i < body; i += nt) *(u32x4_a*)(dst + head + 16u * i) = *(const u32x4_u*)(src + head + 16u * i);
  for (uint32_t i = head + 16u * body + t; i < it.bytes; i += nt) dst[i] = src[i];
}
void mb_raw_copies(uint8_t* out_bytes, const uint8_t* text, const MbRawCopy* items_dev, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_raw_copies, dim3(64, n), dim3(256), 0, BR_STREAM, out_bytes, text, items_dev);
  HIP_CHECK(hipGetLastError());
}

void mb_copy_bits(uint64_t* out, uint64_t dst_bit, const uint64_t* src, uint64_t nbits) {
  const uint32_t words = (uint32_t)((nbits + 63) / 64);
  for_each(words, [=] __device__(uint32_t w) { mb_item_copy_bits_word(out, dst_bit, src, nbits, w); });
  HIP_CHECK(hipGetLastError());
}

}  // namespace brotli_mi355x
