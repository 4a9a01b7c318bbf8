// metablock_hq.h -- the quality >= 10 meta-block builder as device work items.
//
// What the reference does behind the LZ77 stage at quality 10 / 11 (BrotliBuildMetaBlock, metablock.rs:133-307):
//   * a search over the distance parameters (npostfix, ndirect) by the cost of the distance symbols (:88-131, 160-230),
//   * BrotliSplitBlock (block_splitter.rs:840-929): for literals, commands and distances an iterated "assign every symbol
//     to the cheapest of up to 100 entropy codes with a block-switch penalty" (FindBlocks, a Viterbi pass, :232-350) seeded
//     from random samples (:139-222), then ClusterBlocks (:402-688),
//   * context histograms (histogram.rs:465-534) and BrotliClusterHistograms (cluster.rs:353-465) for the literal and
//     distance context maps, all priced with BrotliPopulationCost (bit_cost.rs:76-211),
// everything in f32, left to right.  The items below restate that arithmetic operation by operation (tolerance zero: the
// stream must equal the oracle's byte for byte).  How the work is spread: one 64-lane workgroup per
// (meta-block, kind) job for the sequential algorithms, with the lanes sharing what is order-free or element-wise (integer
// histograms by atomics, rows strided, the costs of FindBlocks two entropy codes per lane, BrotliPopulationCost with its
// non-zero terms compacted in index order); the batches of 64 histograms of the two clustering passes side by side, one
// workgroup each; one thread per symbol for the gather / count passes.  The CPU emulation (tests/emu) compiles the same
// items for one lane; where the device form differs in shape (the wave minimum, ballots) the scalar twin sits beside it
// under BROTLI_HOST_EMU, and the GPU tests (tests/test_quality_9_5.py) are what vouches for the device form.
#ifndef BROTLI_MI355X_METABLOCK_HQ_H_
#define BROTLI_MI355X_METABLOCK_HQ_H_

#include "metablock_items.h"

#if !defined(BROTLI_HOST_EMU)
// wavefront-wide minimum of the device library (DPP row operations; the result is returned to every lane)
extern "C" __device__ __attribute__((const)) float __ockl_wfred_min_f32(float);
#endif

namespace brotli_mi355x {

static constexpr uint32_t kHqMaxAlphabet = 704;
static constexpr uint32_t kHqBatch = 64;            // HISTOGRAMS_PER_BATCH (block_splitter.rs) / max_input_histograms (cluster.rs)
static constexpr uint32_t kHqBatchPairs = 2048;     // 64 * 64 / 2

// a set of histograms: rows of `len` counters, the total and the cached bit cost of each
struct HqHistos {
  uint32_t* data;
  uint32_t* total;
  float* cost;
  uint32_t len;
  BR_DEV uint32_t* row(uint32_t i) const { return data + (size_t)i * len; }
};

struct HqPair {  // HistogramPair, cluster.rs:16-31
  uint32_t idx1, idx2;
  float cost_combo, cost_diff;
};
struct HqBatchRef {  // one batch of 64 histograms of one job
  uint32_t job, batch;
};
struct HqBatchPairs {  // the pair queue of a batch: workgroup memory
  HqPair pairs[2048 + 1];
};

// One (meta-block, kind) splitting job.  Host fills the sizes and the scratch pointers; the device fills num_blocks
// (after FindBlocks) and, through MbBuffers, the final split.
struct HqSplitJob {
  uint32_t m, kind;
  uint32_t length;          // symbols
  uint32_t alphabet;        // 256 / 704 / 544
  uint32_t num_histograms;  // length / symbols_per_histogram + 1, capped (block_splitter.rs:700-708)
  uint32_t stride;          // sampling stride (70 literals, 40 commands / distances)
  uint32_t iters;           // 3 below quality 11, else 10
  float block_switch_cost;
  const uint16_t* data;
  uint8_t* block_ids;       // [length]
  uint32_t* histo_data;     // [num_histograms + 1][alphabet] (the last row is the sample of RefineEntropyCodes)
  uint32_t* histo_total;    // [num_histograms + 1]
  float* insert_cost;       // [alphabet][num_histograms]
  float* cost;              // [bitmaplen * 8]
  uint8_t* switch_signal;   // [length][bitmaplen]
  uint16_t* new_id;         // [num_histograms]
  uint32_t num_blocks;      // out of phase 1
  uint32_t pad;
  // ---- phase 2 (ClusterBlocks), sized by num_blocks
  uint32_t* histogram_symbols;  // [num_blocks]
  uint32_t* block_lengths;      // [num_blocks]
  uint32_t* block_pos;          // [num_blocks + 1] first symbol of every block
  // the batches of 64 blocks are clustered side by side (one workgroup each), every batch in its own rows / slots:
  uint32_t* batch_data;         // [num_blocks][alphabet]
  uint32_t* batch_total;        // [num_blocks]
  float* batch_cost;            // [num_blocks]
  uint32_t* batch_sizes;        // [num_blocks] cluster sizes inside the batch
  uint32_t* batch_clusters;     // [num_blocks] surviving clusters of the batch (indices 0..63), batch_count[b] of them
  uint32_t* batch_symbols;      // [num_blocks] cluster (0..63) of every block inside its batch
  uint32_t* batch_count;        // [ceil(num_blocks / 64)]
  uint32_t* all_data;           // [num_blocks][alphabet]
  uint32_t* all_total;          // [num_blocks]
  float* all_cost;              // [num_blocks]
  uint32_t* cluster_size;       // [num_blocks]
  uint32_t* clusters;           // [num_blocks]
  uint32_t* new_index;          // [num_blocks]
  HqPair* pairs;                // [max(2048, min(64 n, n / 2 * n)) + 1]
};

// One context-map clustering job (literal or distance contexts of one meta-block), cluster.rs:353-465
struct HqClusterJob {
  uint32_t m, kind;        // kind: kSplitLiteral or kSplitDistance
  uint32_t in_size;        // num_types << 6 (literal) or << 2 (distance); num_types when context modelling is disabled
  uint32_t len;            // 256 / 544
  uint32_t expand64;       // literal context modelling disabled: the map of `num_types` entries is spread over 64 contexts each
  uint32_t pad;
  const uint32_t* in_data;  // [in_size][len]
  uint32_t* in_total;       // [in_size]
  uint32_t* out_data;       // [in_size + 2][len] (two scratch rows at the end)
  uint32_t* out_total;      // [in_size + 2]
  float* out_cost;          // [in_size + 2]
  uint32_t* cluster_size;   // [in_size]
  uint32_t* clusters;       // [in_size]
  uint32_t* symbols;        // [in_size]
  uint32_t* new_index;      // [in_size]
  uint32_t* batch_count;    // [ceil(in_size / 64)] clusters left of every batch of 64 inputs (clustered side by side)
  uint32_t* reindex_data;   // [min(in_size, 256)][len]
  uint32_t* reindex_total;  // [min(in_size, 256)]
  float* reindex_cost;      // [min(in_size, 256)]
  HqPair* pairs;            // [max(2048, 64 * in_size) + 1]
};

// ------------------------------------------------------------------------------------------------ execution model
// Every item below that works on one (meta-block, kind) job is entered by ALL lanes of a 64-lane workgroup (one lane in
// the emulation, where BR_TID == 0, BR_NT == 1 and BR_SYNC() is nothing).  The control flow is uniform: every lane keeps
// the job's scalar state in its own registers and redoes the scalar steps (stores of identical values to one address are
// harmless on SIMT hardware); loops over a histogram row or over the symbols are strided across the lanes and end with a
// barrier; f32 sums whose order matters are made by lane 0 from workgroup memory and broadcast.
#if defined(BROTLI_HOST_EMU)
#define BR_ATOMIC_MIN_U32(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define HQ_LD32(p) (*(p))
#else
#define BR_ATOMIC_MIN_U32(p, v) atomicMin((p), (v))
// counters that other lanes bump with atomic adds (performed in the L2) are read with device-scope loads: a plain load could
// be served from a line the vector L1 fetched before the adds
#define HQ_LD32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

struct HqWaveScratch {  // workgroup shared memory
  uint32_t first_pos[256];
  uint32_t ctl[8];
  float f[4];
  uint32_t tmp[kHqMaxAlphabet];  // the histogram being priced (sum of two rows, or one block's)
  uint32_t blk[kHqMaxAlphabet];  // the histogram of the block / input that is compared with the candidates
  // BrotliPopulationCost across the lanes: the non-zero bins compacted in index order
  uint32_t cval[kHqMaxAlphabet];
  float cterm[kHqMaxAlphabet];
  uint32_t cn3[kHqMaxAlphabet];
  uint32_t depth[18];
};

// ------------------------------------------------------------------------------------------------ histogram helpers
BR_DEV void hq_clear(const HqHistos& h, uint32_t i) {  // HistogramClear, histogram.rs:391-399
  uint32_t* r = h.row(i);
  for (uint32_t k = BR_TID; k < h.len; k += BR_NT) r[k] = 0;
  if (BR_TID == 0) {
    h.total[i] = 0;
    h.cost[i] = 3.402e+38f;
  }
  BR_SYNC();
}
BR_DEV void hq_copy(const HqHistos& d, uint32_t di, const HqHistos& s, uint32_t si) {
  uint32_t* dr = d.row(di);
  const uint32_t* sr = s.row(si);
  for (uint32_t k = BR_TID; k < d.len; k += BR_NT) dr[k] = sr[k];
  if (BR_TID == 0) {
    d.total[di] = s.total[si];
    d.cost[di] = s.cost[si];
  }
  BR_SYNC();
}
BR_DEV void hq_add(const HqHistos& d, uint32_t di, const HqHistos& s, uint32_t si) {  // HistogramAddHistogram
  uint32_t* dr = d.row(di);
  const uint32_t* sr = s.row(si);
  for (uint32_t k = BR_TID; k < d.len; k += BR_NT) dr[k] += sr[k];
  if (BR_TID == 0) d.total[di] += s.total[si];
  BR_SYNC();
}

// BrotliPopulationCost, bit_cost.rs:76-211 (the build without "vector_scratch_space").  Sequential: called by one lane.
BR_DEV float hq_population_cost(const EntropyTables& et, const uint32_t* data, uint32_t data_size, uint32_t total_count) {
  const float kOneSymbolHistogramCost = 12.0f, kTwoSymbolHistogramCost = 20.0f, kThreeSymbolHistogramCost = 28.0f,
              kFourSymbolHistogramCost = 37.0f;
  if (total_count == 0) return kOneSymbolHistogramCost;
  uint32_t count = 0;
  uint32_t s[5] = {0, 0, 0, 0, 0};
  for (uint32_t i = 0; i < data_size; ++i) {
    if (data[i] > 0) {
      s[count] = i;
      count++;
      if (count > 4) break;
    }
  }
  if (count == 1) return kOneSymbolHistogramCost;
  if (count == 2) return kTwoSymbolHistogramCost + (float)total_count;
  if (count == 3) {
    const uint32_t h0 = data[s[0]], h1 = data[s[1]], h2 = data[s[2]];
    const uint32_t m01 = h0 > h1 ? h0 : h1;
    const uint32_t hmax = m01 > h2 ? m01 : h2;
    return kThreeSymbolHistogramCost + (float)(2u * (h0 + h1 + h2)) - (float)hmax;
  }
  if (count == 4) {
    uint32_t h[4];
    for (int i = 0; i < 4; ++i) h[i] = data[s[i]];
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j)
        if (h[j] > h[i]) {
          const uint32_t t = h[j];
          h[j] = h[i];
          h[i] = t;
        }
    const uint32_t h23 = h[2] + h[3];
    const uint32_t hmax = h23 > h[0] ? h23 : h[0];
    return kFourSymbolHistogramCost + (float)(3u * h23) + (float)(2u * (h[0] + h[1])) - (float)hmax;
  }
  float bits = 0.0f;
  uint32_t max_depth = 1;
  uint32_t depth_histo[18];
  for (int i = 0; i < 18; ++i) depth_histo[i] = 0;
  const float log2total = br_fast_log2(et, total_count);
  uint32_t reps = 0;
  for (uint32_t i = 0; i < data_size; ++i) {
    const uint32_t histo = data[i];
    if (histo != 0) {
      if (reps != 0) {
        if (reps < 3) {
          depth_histo[0] += reps;
        } else {
          reps -= 2;
          while (reps > 0) {
            depth_histo[17] += 1;
            bits += 3.0f;
            reps >>= 3;
          }
        }
        reps = 0;
      }
      const float log2p = log2total - et.logs_16[histo & 0xffffu];
      const float dd = log2p + 0.5f;
      uint32_t depth = dd > 0.0f ? (uint32_t)dd : 0u;  // `as usize` saturates at 0
      bits += (float)histo * log2p;
      if (depth > 15) depth = 15;
      if (depth > max_depth) max_depth = depth;
      depth_histo[depth] += 1;
    } else {
      reps += 1;
    }
  }
  bits += (float)(18 + 2 * max_depth);
  bits += br_bits_entropy(et, depth_histo, 18);
  return bits;
}

#if !defined(BROTLI_HOST_EMU)
// BrotliPopulationCost of the histogram in S.tmp, by all 64 lanes.  The f32 sum of the reference runs over the bins in
// index order -- for every non-zero bin first one 3.0 per code-length-17 symbol that the run of zeros in front of it
// costs, then count * log2(total / count) -- so its ORDER is kept and only its LENGTH is cut: the lanes compute the terms
// of their bins, find the zero run in front of each (ballot masks, the last non-zero bin of the earlier 64-bin groups
// carried along) and compact the non-zero bins in index order; lane 0 adds up that list (a few dozen entries instead of
// 256 / 544 / 704 dependent steps); the depth histogram and the maximal depth are integers (atomics, any order).
BR_DEV float hq_population_cost_wave(const EntropyTables& et, HqWaveScratch& S, uint32_t len, uint32_t total) {
  const uint32_t lane = threadIdx.x;
  if (total == 0) return 12.0f;
  if (lane < 18) S.depth[lane] = 0;
  if (lane == 0) S.ctl[4] = 1;  // max_depth
  __syncthreads();
  const float log2total = br_fast_log2(et, total);
  uint32_t base = 0;   // non-zero bins so far
  int32_t carry = -1;  // index of the last non-zero bin of the groups in front
  for (uint32_t c0 = 0; c0 < len; c0 += 64) {
    const uint32_t i = c0 + lane;
    const uint32_t histo = i < len ? S.tmp[i] : 0u;
    const bool nz = histo != 0;
    const unsigned long long mask = __ballot(nz);
    if (nz) {
      const unsigned long long below = mask & ((1ull << lane) - 1ull);
      const int32_t prev = below != 0 ? (int32_t)(c0 + 63u - (uint32_t)__clzll((long long)below)) : carry;
      uint32_t reps = (uint32_t)((int32_t)i - prev - 1);
      uint32_t n3 = 0;
      if (reps != 0) {
        if (reps < 3) {
          atomicAdd(&S.depth[0], reps);
        } else {
          reps -= 2;
          while (reps > 0) {
            n3++;
            reps >>= 3;
          }
          atomicAdd(&S.depth[17], n3);
        }
      }
      const float log2p = log2total - et.logs_16[histo & 0xffffu];
      const float dd = log2p + 0.5f;
      uint32_t depth = dd > 0.0f ? (uint32_t)dd : 0u;
      if (depth > 15) depth = 15;
      atomicMax(&S.ctl[4], depth);
      atomicAdd(&S.depth[depth], 1u);
      const uint32_t k = base + (uint32_t)__popcll(below);
      S.cval[k] = histo;
      S.cterm[k] = (float)histo * log2p;
      S.cn3[k] = n3;
    }
    if (mask != 0) carry = (int32_t)(c0 + 63u - (uint32_t)__clzll((long long)mask));
    base += (uint32_t)__popcll(mask);
  }
  __syncthreads();
  if (lane == 0) {
    const uint32_t count = base;
    float bits = 0.0f;
    if (count == 1) {
      bits = 12.0f;
    } else if (count == 2) {
      bits = 20.0f + (float)total;
    } else if (count == 3) {
      const uint32_t h0 = S.cval[0], h1 = S.cval[1], h2 = S.cval[2];
      const uint32_t m01 = h0 > h1 ? h0 : h1;
      const uint32_t hmax = m01 > h2 ? m01 : h2;
      bits = 28.0f + (float)(2u * (h0 + h1 + h2)) - (float)hmax;
    } else if (count == 4) {
      uint32_t h[4];
      for (int i = 0; i < 4; ++i) h[i] = S.cval[i];
      for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
          if (h[j] > h[i]) {
            const uint32_t t = h[j];
            h[j] = h[i];
            h[i] = t;
          }
      const uint32_t h23 = h[2] + h[3];
      const uint32_t hmax = h23 > h[0] ? h23 : h[0];
      bits = 37.0f + (float)(3u * h23) + (float)(2u * (h[0] + h[1])) - (float)hmax;
    } else {
      for (uint32_t k = 0; k < count; ++k) {
        for (uint32_t t = S.cn3[k]; t != 0; --t) bits += 3.0f;
        bits += S.cterm[k];
      }
      bits += (float)(18 + 2 * S.ctl[4]);
      bits += br_bits_entropy(et, S.depth, 18);
    }
    S.f[0] = bits;
  }
  __syncthreads();
  return S.f[0];
}
#endif

// cost of the histogram in S.tmp (complete and visible: the caller has passed a barrier); all lanes call, all get the value
BR_DEV float hq_cost_in_tmp(const EntropyTables& et, HqWaveScratch& S, uint32_t len, uint32_t total) {
#if defined(BROTLI_HOST_EMU)
  return hq_population_cost(et, S.tmp, len, total);
#else
  return hq_population_cost_wave(et, S, len, total);
#endif
}

// cost of the histogram a (+ b): the sum is formed in workgroup memory by all lanes
BR_DEV float hq_cost_of_sum(const EntropyTables& et, HqWaveScratch& S, const uint32_t* a, const uint32_t* b, uint32_t len, uint32_t total) {
  for (uint32_t k = BR_TID; k < len; k += BR_NT) S.tmp[k] = a[k] + (b != nullptr ? b[k] : 0u);
  BR_SYNC();
  return hq_cost_in_tmp(et, S, len, total);
}

// ------------------------------------------------------------------------------------------------ cluster.rs
BR_DEV float hq_cluster_cost_diff(const EntropyTables& et, uint32_t size_a, uint32_t size_b) {  // :33-39
  const uint32_t size_c = size_a + size_b;
  return (float)size_a * br_fast_log2(et, size_a) + (float)size_b * br_fast_log2(et, size_b) - (float)size_c * br_fast_log2(et, size_c);
}
BR_DEV bool hq_pair_is_less(const HqPair& p1, const HqPair& p2) {  // :41-48
  if (p1.cost_diff != p2.cost_diff) return p1.cost_diff > p2.cost_diff;
  return (uint32_t)(p1.idx2 - p1.idx1) > (uint32_t)(p2.idx2 - p2.idx1);
}

// BrotliCompareAndPushToQueue, cluster.rs:52-121
BR_DEV void hq_compare_and_push(const EntropyTables& et, HqWaveScratch& S, const HqHistos& out, const uint32_t* cluster_size, uint32_t idx1,
                                uint32_t idx2, uint32_t max_num_pairs, HqPair* pairs, uint32_t* num_pairs) {
  if (idx1 == idx2) return;
  if (idx2 < idx1) {
    const uint32_t t = idx2;
    idx2 = idx1;
    idx1 = t;
  }
  bool is_good_pair = false;
  HqPair p;
  p.idx1 = idx1;
  p.idx2 = idx2;
  p.cost_combo = 0.0f;
  p.cost_diff = 0.5f * hq_cluster_cost_diff(et, cluster_size[idx1], cluster_size[idx2]);
  p.cost_diff -= out.cost[idx1];
  p.cost_diff -= out.cost[idx2];
  if (out.total[idx1] == 0) {
    p.cost_combo = out.cost[idx2];
    is_good_pair = true;
  } else if (out.total[idx2] == 0) {
    p.cost_combo = out.cost[idx1];
    is_good_pair = true;
  } else {
    float threshold = 1e38f;
    if (*num_pairs != 0) threshold = pairs[0].cost_diff > 0.0f ? pairs[0].cost_diff : 0.0f;
    const float cost_combo = hq_cost_of_sum(et, S, out.row(idx1), out.row(idx2), out.len, out.total[idx1] + out.total[idx2]);
    if (cost_combo < threshold - p.cost_diff) {
      p.cost_combo = cost_combo;
      is_good_pair = true;
    }
  }
  if (is_good_pair) {
    p.cost_diff += p.cost_combo;
    if (*num_pairs > 0 && hq_pair_is_less(pairs[0], p)) {
      if (*num_pairs < max_num_pairs) {
        pairs[*num_pairs] = pairs[0];
        ++*num_pairs;
      }
      pairs[0] = p;
    } else if (*num_pairs < max_num_pairs) {
      pairs[*num_pairs] = p;
      ++*num_pairs;
    }
  }
}

// BrotliHistogramCombine, cluster.rs:123-236
BR_DEV uint32_t hq_histogram_combine(const EntropyTables& et, HqWaveScratch& S, const HqHistos& out, uint32_t* cluster_size, uint32_t* symbols,
                                     uint32_t* clusters, HqPair* pairs, uint32_t num_clusters, uint32_t symbols_size, uint32_t max_clusters,
                                     uint32_t max_num_pairs) {
  float cost_diff_threshold = 0.0f;
  uint32_t min_cluster_size = 1;
  uint32_t num_pairs = 0;
  for (uint32_t i1 = 0; i1 < num_clusters; ++i1)
    for (uint32_t i2 = i1 + 1; i2 < num_clusters; ++i2)
      hq_compare_and_push(et, S, out, cluster_size, clusters[i1], clusters[i2], max_num_pairs, pairs, &num_pairs);
  while (num_clusters > min_cluster_size) {
    if (pairs[0].cost_diff >= cost_diff_threshold) {
      cost_diff_threshold = 1e38f;
      min_cluster_size = max_clusters;
      continue;
    }
    const uint32_t best_idx1 = pairs[0].idx1, best_idx2 = pairs[0].idx2;
    const float best_combo = pairs[0].cost_combo;
    hq_add(out, best_idx1, out, best_idx2);
    out.cost[best_idx1] = best_combo;
    cluster_size[best_idx1] += cluster_size[best_idx2];
    BR_SYNC();  // (every lane has read the old cluster size before any lane's store can land ... and reads the new one below)
    for (uint32_t i = BR_TID; i < symbols_size; i += BR_NT)
      if (symbols[i] == best_idx2) symbols[i] = best_idx1;
    BR_SYNC();
    for (uint32_t i = 0; i < num_clusters; ++i) {
      if (clusters[i] == best_idx2) {
        for (uint32_t k = i; k + 1 < num_clusters; ++k) clusters[k] = clusters[k + 1];
        break;
      }
    }
    --num_clusters;
    {
      uint32_t copy_to_idx = 0;
      for (uint32_t i = 0; i < num_pairs; ++i) {
        const HqPair p = pairs[i];
        if (p.idx1 == best_idx1 || p.idx2 == best_idx1 || p.idx1 == best_idx2 || p.idx2 == best_idx2) continue;
        if (hq_pair_is_less(pairs[0], p)) {
          const HqPair front = pairs[0];
          pairs[0] = p;
          pairs[copy_to_idx] = front;
        } else {
          pairs[copy_to_idx] = p;
        }
        ++copy_to_idx;
      }
      num_pairs = copy_to_idx;
    }
    for (uint32_t i = 0; i < num_clusters; ++i)
      hq_compare_and_push(et, S, out, cluster_size, best_idx1, clusters[i], max_num_pairs, pairs, &num_pairs);
  }
  return num_clusters;
}

// BrotliHistogramBitCostDistance, cluster.rs:238-254, of the histogram in S.blk (total `blk_total`) to candidate ci
BR_DEV float hq_bit_cost_distance(const EntropyTables& et, HqWaveScratch& S, uint32_t blk_total, const HqHistos& cand_set, uint32_t ci) {
  if (blk_total == 0) return 0.0f;
  return hq_cost_of_sum(et, S, S.blk, cand_set.row(ci), cand_set.len, blk_total + cand_set.total[ci]) - cand_set.cost[ci];
}

// ------------------------------------------------------------------------------------------------ block_splitter.rs
// ------------------------------------------------------------------------------------------------ block_splitter.rs
BR_DEV uint32_t hq_my_rand(uint32_t* seed) {  // :131-137
  *seed = *seed * 16807u;
  if (*seed == 0) *seed = 1;
  return *seed;
}
BR_DEV float hq_bit_cost(const EntropyTables& et, uint32_t count) { return count == 0 ? -2.0f : br_fast_log2(et, count); }  // :224-230


// ---- phase 1 of SplitByteVector, cooperatively: the item is entered by all lanes of a 64-lane workgroup (one lane in the
// emulation).  Histogram counters are bumped by atomic adds (integers: any order gives the same histogram), the costs of
// FindBlocks live in registers, two entropy codes per lane.
// InitialEntropyCodes + RefineEntropyCodes, block_splitter.rs:139-222.  The sample positions come from a 32-bit
// multiplicative generator that every lane steps for itself (uniform); a sample of `stride` symbols is one atomic add per lane.
BR_DEV void hq_seed_entropy_codes(const HqSplitJob& J, const HqHistos& H) {
  const uint16_t* data = J.data;
  const uint32_t length = J.length, stride = J.stride, nh = J.num_histograms;
  for (uint32_t i = BR_TID; i < nh * H.len; i += BR_NT) H.data[i] = 0;
  for (uint32_t i = BR_TID; i < nh; i += BR_NT) H.total[i] = 0;
  BR_SYNC();
  {
    uint32_t seed = 7;
    const uint32_t block_length = length / nh;
    for (uint32_t i = 0; i < nh; ++i) {
      uint32_t pos = (uint32_t)((uint64_t)length * i / nh);
      if (i != 0) pos += hq_my_rand(&seed) % block_length;
      if (pos + stride >= length) pos = length - stride - 1;
      if (BR_TID == 0) H.total[i] += stride;
      uint32_t* r = H.row(i);
      for (uint32_t k = BR_TID; k < stride; k += BR_NT) BR_ATOMIC_ADD_U32(&r[data[pos + k]], 1u);
    }
  }
  BR_SYNC();
  {
    uint32_t iters = (uint32_t)(2ull * length / stride) + 100u;  // kIterMulForRefining, kMinItersForRefining
    uint32_t seed = 7;
    iters = (iters + nh - 1) / nh * nh;
    for (uint32_t iter = 0; iter < iters; ++iter) {
      // RandomSample (:167-188) straight into the target histogram (sample + add = add)
      uint32_t pos, n = stride;
      if (stride >= length) {
        pos = 0;
        n = length;
      } else {
        pos = hq_my_rand(&seed) % (length - stride + 1);
      }
      const uint32_t t = iter % nh;
      if (BR_TID == 0) H.total[t] += n;
      uint32_t* r = H.row(t);
      for (uint32_t k = BR_TID; k < n; k += BR_NT) BR_ATOMIC_ADD_U32(&r[data[pos + k]], 1u);
    }
  }
  BR_SYNC();
}

// FindBlocks, block_splitter.rs:232-350 (with update_cost_and_signal :46-82).  Per symbol: cost[k] += insert_cost[sym][k];
// the minimum over k (ties to the lowest k); cost[k] = min(cost[k] - min, switch cost), and one "would switch" bit per k.
// The reference also steps the lanes that pad the cost vector to a multiple of eight; they feed neither the minimum nor
// the trace-back (which reads bit cur_id < num_histograms only) and are left out here.
BR_DEV uint32_t hq_find_blocks(const EntropyTables& et, const HqSplitJob& J, const HqHistos& H, uint32_t num_histograms) {
  const uint16_t* data = J.data;
  const uint32_t length = J.length, data_size = J.alphabet;
  uint8_t* block_id = J.block_ids;
  const uint32_t bitmaplen = (num_histograms + 7) >> 3;
  if (num_histograms == 0) return 0;
  if (num_histograms <= 1) {
    for (uint32_t i = BR_TID; i < length; i += BR_NT) block_id[i] = 0;
    BR_SYNC();
    return 1;
  }
  float* insert_cost = J.insert_cost;
  uint8_t* switch_signal = J.switch_signal;
  // insert_cost[i][j] = log2(total_j) - bit_cost(histogram_j[i])  (the reference keeps log2(total_j) in row 0 until that row
  // is overwritten last, :262-283)
  for (uint32_t idx = BR_TID; idx < data_size * num_histograms; idx += BR_NT) {
    const uint32_t i = idx / num_histograms, j = idx - i * num_histograms;
    insert_cost[idx] = br_fast_log2(et, HQ_LD32(&H.total[j])) - hq_bit_cost(et, HQ_LD32(&H.row(j)[i]));
  }
  BR_SYNC();
#if defined(BROTLI_HOST_EMU)
  float* cost = J.cost;
  for (uint32_t i = 0; i < num_histograms; ++i) cost[i] = 0.0f;
  for (uint32_t byte_ix = 0; byte_ix < length; ++byte_ix) {
    const size_t ix = (size_t)byte_ix * bitmaplen;
    const float* ic = insert_cost + (size_t)data[byte_ix] * num_histograms;
    float min_cost = 1e38f;
    float block_switch_cost = J.block_switch_cost;
    uint8_t best = block_id[byte_ix];
    for (uint32_t k = 0; k < num_histograms; ++k) {
      const float c = cost[k] + ic[k];
      cost[k] = c;
      if (c < min_cost) {
        min_cost = c;
        best = (uint8_t)k;
      }
    }
    block_id[byte_ix] = best;
    if (byte_ix < 2000) block_switch_cost *= (0.77f + 0.07f * (float)byte_ix / 2000.0f);
    for (uint32_t j = 0; j < bitmaplen; ++j) switch_signal[ix + j] = 0;
    for (uint32_t k = 0; k < num_histograms; ++k) {
      const float d = cost[k] - min_cost;
      if (d >= block_switch_cost) switch_signal[ix + (k >> 3)] |= (uint8_t)(1u << (k & 7));
      cost[k] = d < block_switch_cost ? d : block_switch_cost;
    }
  }
#else
  {
    const uint32_t lane = threadIdx.x;
    const bool has0 = lane < num_histograms, has1 = lane + 64 < num_histograms;
    const uint32_t k0 = has0 ? lane : 0u, k1 = has1 ? lane + 64 : 0u;
    float c0 = 0.0f, c1 = 0.0f;
    constexpr uint32_t kAhead = 8;  // symbols whose insert-cost rows are fetched together, ahead of the dependent chain
    for (uint32_t base = 0; base < length; base += kAhead) {
      float a0[kAhead], a1[kAhead];
#pragma unroll
      for (uint32_t u = 0; u < kAhead; ++u) {
        const uint32_t bx = base + u < length ? base + u : length - 1;
        const float* ic = insert_cost + (size_t)data[bx] * num_histograms;
        a0[u] = ic[k0];
        a1[u] = ic[k1];
      }
#pragma unroll
      for (uint32_t u = 0; u < kAhead; ++u) {
        const uint32_t byte_ix = base + u;
        if (byte_ix < length) {  // (uniform)
          c0 += a0[u];
          c1 += a1[u];
          // every cost is >= 0 (insert costs are, and the carried part lies in [0, switch cost]): absent lanes hold a huge value
          float v = has0 ? c0 : 3.0e38f;
          if (has1 && c1 < v) v = c1;
          const float min_cost = __ockl_wfred_min_f32(v);
          const unsigned long long e0 = __ballot(has0 && c0 == min_cost), e1 = __ballot(has1 && c1 == min_cost);
          const uint32_t best = e0 != 0 ? (uint32_t)__ffsll((long long)e0) - 1u : 64u + (uint32_t)__ffsll((long long)e1) - 1u;
          float block_switch_cost = J.block_switch_cost;
          if (byte_ix < 2000) block_switch_cost *= (0.77f + 0.07f * (float)byte_ix / 2000.0f);
          const float d0 = c0 - min_cost, d1 = c1 - min_cost;
          const unsigned long long m0 = __ballot(has0 && d0 >= block_switch_cost), m1 = __ballot(has1 && d1 >= block_switch_cost);
          c0 = d0 < block_switch_cost ? d0 : block_switch_cost;
          c1 = d1 < block_switch_cost ? d1 : block_switch_cost;
          if (lane < bitmaplen)
            switch_signal[(size_t)byte_ix * bitmaplen + lane] = (uint8_t)(lane < 8 ? (m0 >> (8 * lane)) : (m1 >> (8 * (lane - 8))));
          if (lane == 0) block_id[byte_ix] = (uint8_t)best;
        }
      }
    }
  }
#endif
  BR_SYNC();
  // trace back (:318-348): the id in force changes only where a switch was signalled for it
#if defined(BROTLI_HOST_EMU)
  {
    uint32_t num_blocks = 1;
    uint32_t byte_ix = length - 1;
    size_t ix = (size_t)byte_ix * bitmaplen;
    uint8_t cur_id = block_id[byte_ix];
    while (byte_ix > 0) {
      const uint8_t mask = (uint8_t)(1u << (cur_id & 7));
      --byte_ix;
      ix -= bitmaplen;
      if ((switch_signal[ix + (cur_id >> 3)] & mask) != 0 && cur_id != block_id[byte_ix]) {
        cur_id = block_id[byte_ix];
        ++num_blocks;
      }
      block_id[byte_ix] = cur_id;
    }
    const_cast<HqSplitJob&>(J).num_blocks = num_blocks;
  }
#else
  {
    // 64 positions per step: every lane tests "a switch to my own id was signalled here" for the id in force; the nearest hit
    // (ballot) ends the stretch, everything in front of it takes the id in force
    const uint32_t lane = threadIdx.x;
    uint32_t num_blocks = 1;
    uint32_t p = length - 1;  // positions below p are still to do
    uint32_t cur = block_id[p];
    while (p > 0) {
      const bool valid = lane < p;
      const uint32_t q = valid ? p - 1u - lane : 0u;
      const uint32_t bid = valid ? (uint32_t)block_id[q] : cur;
      const uint32_t sig = valid ? ((uint32_t)switch_signal[(size_t)q * bitmaplen + (cur >> 3)] >> (cur & 7u)) & 1u : 0u;
      const unsigned long long m = __ballot(valid && sig != 0 && bid != cur);
      if (m == 0) {
        if (valid) block_id[q] = (uint8_t)cur;
        p = p > 64u ? p - 64u : 0u;
      } else {
        const uint32_t f = (uint32_t)__ffsll((long long)m) - 1u;
        if (valid && lane < f) block_id[q] = (uint8_t)cur;
        cur = (uint32_t)__shfl((int)bid, (int)f, 64);
        ++num_blocks;
        p = p - 1u - f;  // the position of the hit: it keeps its own id, which is the id in force from here on
      }
    }
    if (lane == 0) const_cast<HqSplitJob&>(J).num_blocks = num_blocks;
  }
#endif
  BR_SYNC();
  return 0;
}

// phase 1 of SplitByteVector (block_splitter.rs:690-838): seeds + the FindBlocks / RemapBlockIds / BuildBlockHistograms
// iterations.  Leaves block_ids and J.num_blocks; 0 blocks = "no symbols", 1 with length < 128 = "one block, no search".
BR_DEV void hq_item_find_blocks(const EntropyTables& et, HqSplitJob& J, HqWaveScratch& S) {
  if (J.length == 0) {
    if (BR_TID == 0) J.num_blocks = 0;
    return;
  }
  if (J.length < 128) {  // kMinLengthForBlockSplitting
    if (BR_TID == 0) J.num_blocks = 1;
    for (uint32_t i = BR_TID; i < J.length; i += BR_NT) J.block_ids[i] = 0;
    return;
  }
  HqHistos H;
  H.data = J.histo_data;
  H.total = J.histo_total;
  H.cost = J.insert_cost;  // (the bit-cost slot is never used for the block histograms)
  H.len = J.alphabet;
  hq_seed_entropy_codes(J, H);
  uint32_t num_histograms = J.num_histograms;
  for (uint32_t it = 0; it < J.iters; ++it) {
    if (num_histograms <= 1) {
      for (uint32_t i = BR_TID; i < J.length; i += BR_NT) J.block_ids[i] = 0;
      if (BR_TID == 0) J.num_blocks = 1;
      BR_SYNC();
    } else {
      hq_find_blocks(et, J, H, num_histograms);
    }
    // RemapBlockIds (:352-378): new ids in the order of first appearance
    for (uint32_t i = BR_TID; i < 256; i += BR_NT) S.first_pos[i] = 0xffffffffu;
    BR_SYNC();
    for (uint32_t i = BR_TID; i < J.length; i += BR_NT) BR_ATOMIC_MIN_U32(&S.first_pos[J.block_ids[i]], i);
    BR_SYNC();
    if (BR_TID == 0) {
      uint32_t next_id = 0;
      for (;;) {  // (at most 100 ids: selection by smallest first position)
        uint32_t best = 0xffffffffu, best_id = 0;
        for (uint32_t id = 0; id < num_histograms; ++id)
          if (S.first_pos[id] < best) {
            best = S.first_pos[id];
            best_id = id;
          }
        if (best == 0xffffffffu) break;
        J.new_id[best_id] = (uint16_t)next_id++;
        S.first_pos[best_id] = 0xffffffffu;
      }
      S.ctl[0] = next_id;
    }
    BR_SYNC();
    num_histograms = S.ctl[0];
    // BuildBlockHistograms (:380-400)
    for (uint32_t i = BR_TID; i < num_histograms * H.len; i += BR_NT) H.data[i] = 0;
    for (uint32_t i = BR_TID; i < num_histograms; i += BR_NT) H.total[i] = 0;
    BR_SYNC();
    for (uint32_t i = BR_TID; i < J.length; i += BR_NT) {
      const uint32_t b = J.new_id[J.block_ids[i]];
      J.block_ids[i] = (uint8_t)b;
      BR_ATOMIC_ADD_U32(&H.row(b)[J.data[i]], 1u);
      BR_ATOMIC_ADD_U32(&H.total[b], 1u);
    }
    BR_SYNC();
  }
}

// histogram of `n` symbols into workgroup memory (S.tmp or S.blk), by atomic adds
BR_DEV void hq_block_histogram(uint32_t* dst, uint32_t len, const uint16_t* data, uint32_t n) {
  for (uint32_t k = BR_TID; k < len; k += BR_NT) dst[k] = 0;
  BR_SYNC();
  for (uint32_t k = BR_TID; k < n; k += BR_NT) BR_ATOMIC_ADD_U32(&dst[data[k]], 1u);
  BR_SYNC();
}

// phase 2: ClusterBlocks, block_splitter.rs:402-688, in three launches: the block lengths (one pass over the ids), the
// batches of 64 blocks side by side (their clusterings do not depend on each other), then the rest -- the clusters of all
// batches against each other, every block to its closest cluster, the split of (m, kind): types, lengths, starts, counts.
BR_DEV void hq_item_blocks_prep(const HqSplitJob& J) {
  if (J.num_blocks == 0 || J.length < 128) return;
#if defined(BROTLI_HOST_EMU)
  {
    uint32_t block_idx = 0, run = 0, start = 0;
    for (uint32_t i = 0; i < J.length; ++i) {
      run++;
      if (i + 1 == J.length || J.block_ids[i] != J.block_ids[i + 1]) {
        J.block_lengths[block_idx] = run;
        J.block_pos[block_idx] = start;
        start += run;
        block_idx++;
        run = 0;
      }
    }
    J.block_pos[block_idx] = start;
  }
#else
  {
    // the ends of the runs of equal ids, 64 positions per step (ballot + rank among the ends), then the lengths as differences
    const uint32_t lane = threadIdx.x;
    uint32_t base = 0;
    if (lane == 0) J.block_pos[0] = 0;
    for (uint32_t i0 = 0; i0 < J.length; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool end = i < J.length && (i + 1 == J.length || J.block_ids[i] != J.block_ids[i + 1]);
      const unsigned long long m = __ballot(end);
      if (end) J.block_pos[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) + 1u] = i + 1u;
      base += (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (uint32_t b = lane; b < base; b += 64) J.block_lengths[b] = J.block_pos[b + 1] - J.block_pos[b];
  }
#endif
}

BR_DEV void hq_item_cluster_blocks_batch(const EntropyTables& et, const HqSplitJob& J, uint32_t b, HqWaveScratch& S, HqPair* pairs) {
  if (J.num_blocks == 0 || J.length < 128) return;
  const uint32_t len = J.alphabet;
  const uint32_t i0 = b * kHqBatch;
  const uint32_t num_to_combine = J.num_blocks - i0 < kHqBatch ? J.num_blocks - i0 : kHqBatch;
  HqHistos batch;  // the batch's own 64 rows
  batch.data = J.batch_data + (size_t)i0 * len;
  batch.total = J.batch_total + i0;
  batch.cost = J.batch_cost + i0;
  batch.len = len;
  uint32_t* sizes = J.batch_sizes + i0;
  uint32_t* new_clusters = J.batch_clusters + i0;
  uint32_t* symbols = J.batch_symbols + i0;
  for (uint32_t j = 0; j < num_to_combine; ++j) {
    const uint32_t n = J.block_lengths[i0 + j];
    hq_block_histogram(S.tmp, len, J.data + J.block_pos[i0 + j], n);
    uint32_t* row = batch.row(j);
    for (uint32_t k = BR_TID; k < len; k += BR_NT) row[k] = S.tmp[k];
    const float cost_j = hq_cost_in_tmp(et, S, len, n);
    if (BR_TID == 0) {
      batch.total[j] = n;
      batch.cost[j] = cost_j;
      new_clusters[j] = j;
      symbols[j] = j;
      sizes[j] = 1;
    }
    BR_SYNC();
  }
  const uint32_t num_new_clusters =
      hq_histogram_combine(et, S, batch, sizes, symbols, new_clusters, pairs, num_to_combine, num_to_combine, kHqBatch, kHqBatchPairs);
  if (BR_TID == 0) J.batch_count[b] = num_new_clusters;
}

BR_DEV void hq_item_cluster_blocks(const MbBuffers& B, const HqSplitJob& J, HqWaveScratch& S) {
  const EntropyTables& et = B.et;
  const MbDesc& d = B.descs[J.m];
  MbResult& r = B.results[J.m];
  const uint32_t kind = J.kind;
  uint8_t* types = B.block_types[kind] + d.block_base[kind];
  uint32_t* lengths = B.block_lengths[kind] + d.block_base[kind];
  uint32_t* starts = B.block_start[kind] + d.block_base[kind];
  if (J.num_blocks == 0) {  // no symbols: one type, no blocks
    if (BR_TID == 0) {
      r.num_types[kind] = 1;
      r.num_blocks[kind] = 0;
      starts[0] = 0;
    }
    return;
  }
  if (J.length < 128) {
    if (BR_TID == 0) {
      r.num_types[kind] = 1;
      r.num_blocks[kind] = 1;
      types[0] = 0;
      lengths[0] = J.length;
      starts[0] = 0;
      starts[1] = J.length;
    }
    return;
  }
  const uint16_t* data = J.data;
  const uint32_t num_blocks = J.num_blocks, len = J.alphabet;
  uint32_t* histogram_symbols = J.histogram_symbols;
  const uint32_t* block_lengths = J.block_lengths;
  HqHistos all;
  all.data = J.all_data;
  all.total = J.all_total;
  all.cost = J.all_cost;
  all.len = len;
  uint32_t* cluster_size = J.cluster_size;
  uint32_t all_size = 0;
  uint32_t num_clusters = 0;
  uint32_t* remap = J.new_index;  // (free until the last pass)
  for (uint32_t i = 0; i < num_blocks; i += kHqBatch) {  // the clusters of the batches, in batch order (:480-560)
    const uint32_t num_to_combine = num_blocks - i < kHqBatch ? num_blocks - i : kHqBatch;
    const uint32_t num_new_clusters = J.batch_count[i / kHqBatch];
    HqHistos batch;
    batch.data = J.batch_data + (size_t)i * len;
    batch.total = J.batch_total + i;
    batch.cost = J.batch_cost + i;
    batch.len = len;
    const uint32_t* new_clusters = J.batch_clusters + i;
    for (uint32_t j = 0; j < num_new_clusters; ++j) {
      hq_copy(all, all_size, batch, new_clusters[j]);
      cluster_size[all_size] = J.batch_sizes[i + new_clusters[j]];
      all_size++;
      remap[new_clusters[j]] = j;
    }
    BR_SYNC();
    for (uint32_t j = 0; j < num_to_combine; ++j) histogram_symbols[i + j] = num_clusters + remap[J.batch_symbols[i + j]];
    num_clusters += num_new_clusters;
    BR_SYNC();
  }
  uint32_t max_num_pairs = 64u * num_clusters;
  {
    const uint64_t alt = (uint64_t)(num_clusters / 2) * num_clusters;
    if (alt < max_num_pairs) max_num_pairs = (uint32_t)alt;
  }
  uint32_t* clusters = J.clusters;
  for (uint32_t i = 0; i < num_clusters; ++i) clusters[i] = i;
  BR_SYNC();
  const uint32_t num_final_clusters =
      hq_histogram_combine(et, S, all, cluster_size, histogram_symbols, clusters, J.pairs, num_clusters, num_blocks, 256, max_num_pairs);
  uint32_t* new_index = J.new_index;
  const uint32_t kInvalidIndex = 0xffffffffu;
  for (uint32_t i = 0; i < num_clusters; ++i) new_index[i] = kInvalidIndex;
  BR_SYNC();
  {
    uint32_t next_index = 0;
    for (uint32_t i = 0; i < num_blocks; ++i) {
      const uint32_t n = block_lengths[i];
      hq_block_histogram(S.blk, len, data + J.block_pos[i], n);
      uint32_t best_out = i == 0 ? histogram_symbols[0] : histogram_symbols[i - 1];
      float best_bits = hq_bit_cost_distance(et, S, n, all, best_out);
      for (uint32_t j = 0; j < num_final_clusters; ++j) {
        const float cur_bits = hq_bit_cost_distance(et, S, n, all, clusters[j]);
        if (cur_bits < best_bits) {
          best_bits = cur_bits;
          best_out = clusters[j];
        }
      }
      histogram_symbols[i] = best_out;
      if (new_index[best_out] == kInvalidIndex) new_index[best_out] = next_index++;
    }
  }
  BR_SYNC();
  if (BR_TID == 0) {
    uint32_t cur_length = 0, block_idx = 0, start = 0;
    uint8_t max_type = 0;
    for (uint32_t i = 0; i < num_blocks; ++i) {
      cur_length += block_lengths[i];
      if (i + 1 == num_blocks || histogram_symbols[i] != histogram_symbols[i + 1]) {
        const uint8_t id = (uint8_t)new_index[histogram_symbols[i]];
        types[block_idx] = id;
        lengths[block_idx] = cur_length;
        starts[block_idx] = start;
        start += cur_length;
        if (id > max_type) max_type = id;
        cur_length = 0;
        ++block_idx;
      }
    }
    starts[block_idx] = start;
    r.num_blocks[kind] = block_idx;
    r.num_types[kind] = (uint32_t)max_type + 1;
  }
}

// ------------------------------------------------------------------------------------------------ metablock.rs
// CommandRestoreDistanceCode, command.rs:84-100
BR_DEV uint32_t hq_restore_distance_code(const Command& c, uint32_t ndirect, uint32_t npostfix) {
  const uint32_t dcode = c.dist_prefix_ & 0x3ffu;
  if (dcode < 16 + ndirect) return dcode;
  const uint32_t nbits = (uint32_t)(c.dist_prefix_ >> 10);
  const uint32_t extra = c.dist_extra_;
  const uint32_t postfix_mask = (1u << npostfix) - 1;
  const uint32_t hcode = (dcode - ndirect - 16u) >> npostfix;
  const uint32_t lcode = (dcode - ndirect - 16u) & postfix_mask;
  const uint32_t offset = ((2u + (hcode & 1)) << nbits) - 4u;
  return ((offset + extra) << npostfix) + lcode + ndirect + 16u;
}
// PrefixEncodeCopyDistance, command.rs:134-173
BR_DEV void hq_prefix_encode_distance(uint32_t distance_code, uint32_t ndirect, uint32_t npostfix, uint16_t* code, uint32_t* extra_bits) {
  if (distance_code < 16 + ndirect) {
    *code = (uint16_t)distance_code;
    *extra_bits = 0;
    return;
  }
  const uint64_t dist = (1ull << (npostfix + 2)) + ((uint64_t)distance_code - 16 - ndirect);
  const uint32_t bucket = (63u ^ (uint32_t)__builtin_clzll(dist)) - 1;
  const uint64_t postfix_mask = (1u << npostfix) - 1;
  const uint64_t postfix = dist & postfix_mask;
  const uint64_t prefix = (dist >> bucket) & 1;
  const uint64_t offset = (2 + prefix) << bucket;
  const uint64_t nbits = bucket - npostfix;
  *code = (uint16_t)((nbits << 10) | (16 + ndirect + ((2 * (nbits - 1) + prefix) << npostfix) + postfix));
  *extra_bits = (uint32_t)((dist - offset) >> npostfix);
}
// BrotliInitDistanceParams, metablock.rs:28-60 (large_window: MbDesc::hq bit 1)
BR_DEV uint32_t hq_distance_alphabet_size(uint32_t npostfix, uint32_t ndirect, bool large_window = false) {
  return 16 + ndirect + ((large_window ? 62u : 24u) << (npostfix + 1));
}
BR_DEV uint32_t hq_max_distance(uint32_t npostfix, uint32_t ndirect, bool large_window = false) {
  if (large_window) {
    // (no distance symbol in use may encode more than BROTLI_MAX_ALLOWED_DISTANCE with all its extra bits set)
    const uint32_t bound = npostfix == 0 ? 0u : (npostfix == 1 ? 4u : (npostfix == 2 ? 12u : 28u));
    const uint32_t postfix = 1u << npostfix;
    if (ndirect < bound) return 0x07fffffcu - (bound - ndirect);
    if (ndirect >= bound + postfix) return (3u << 29) - 4u + (ndirect - bound);
    return 0x07fffffcu;
  }
  return ndirect + (1u << (24 + npostfix + 2)) - (1u << (npostfix + 2));
}


// ComputeDistanceCost, metablock.rs:88-131: the histogram of the re-coded distance symbols in workgroup memory (atomic
// adds), their extra bits as an integer sum (the reference adds them up in f64: small integers, exact in any order).
BR_DEV bool hq_distance_cost(const MbBuffers& B, const MbDesc& d, HqWaveScratch& S, uint32_t new_npostfix, uint32_t new_ndirect, double* cost) {
  const bool equal_params = d.dist_postfix_bits == new_npostfix && d.num_direct_distance_codes == new_ndirect;
  const uint32_t new_max_distance = hq_max_distance(new_npostfix, new_ndirect, (d.hq & 2u) != 0);
  for (uint32_t i = BR_TID; i < kNumDistanceHistoSymbols; i += BR_NT) S.tmp[i] = 0;
  if (BR_TID == 0) S.ctl[0] = S.ctl[1] = S.ctl[2] = 0;  // too far, symbols, extra bits
  BR_SYNC();
  // (the symbol count and the extra bits are summed per thread and added once: three atomic adds per command, two of them by
  // every lane on the same word, were most of the 114 ms this search took for the one million commands of a lone 8 MiB meta-block)
  uint32_t my_far = 0, my_symbols = 0, my_extra = 0;
  for (uint32_t c = d.cmd_offset + BR_TID; c < d.cmd_offset + d.n_cmds; c += BR_NT) {
    const Command cmd = B.cmds[c];
    if (!br_command_has_distance(cmd)) continue;
    uint16_t dist_prefix;
    uint32_t dist_extra;
    if (equal_params) {
      dist_prefix = cmd.dist_prefix_;
    } else {
      const uint32_t distance = hq_restore_distance_code(cmd, d.num_direct_distance_codes, d.dist_postfix_bits);
      if (distance > new_max_distance) {
        my_far++;
        continue;
      }
      hq_prefix_encode_distance(distance, new_ndirect, new_npostfix, &dist_prefix, &dist_extra);
    }
    BR_ATOMIC_ADD_U32(&S.tmp[dist_prefix & 0x3ffu], 1u);
    my_symbols++;
    my_extra += (uint32_t)(dist_prefix >> 10);
  }
  if (my_far) BR_ATOMIC_ADD_U32(&S.ctl[0], my_far);
  if (my_symbols) BR_ATOMIC_ADD_U32(&S.ctl[1], my_symbols);
  if (my_extra) BR_ATOMIC_ADD_U32(&S.ctl[2], my_extra);
  BR_SYNC();
  const bool too_far = S.ctl[0] != 0;
  const uint32_t total = S.ctl[1];
  const double extra_bits = (double)S.ctl[2];
  BR_SYNC();  // (every lane has read S.ctl)
  if (too_far) return false;
  *cost = (double)hq_cost_in_tmp(B.et, S, kNumDistanceHistoSymbols, total) + extra_bits;
  return true;
}

// the distance-parameter search of BrotliBuildMetaBlock (metablock.rs:160-240) + RecomputeDistancePrefixes (:62-86) on the
// meta-block's own copy of the commands.  Result in MbResult::hq_postfix / hq_ndirect.
BR_DEV void hq_item_distance_params(const MbBuffers& B, uint32_t m, HqWaveScratch& S) {
  const MbDesc& d = B.descs[m];
  MbResult& r = B.results[m];
  uint32_t best_npostfix = d.dist_postfix_bits, best_ndirect = d.num_direct_distance_codes;
  if (d.uncompressed) {
    if (BR_TID == 0) {
      r.hq_postfix = best_npostfix;
      r.hq_ndirect = best_ndirect;
    }
    return;
  }
  uint32_t ndirect_msb = 0;
  bool check_orig = true;
  double best_dist_cost = 1e99;
  for (uint32_t npostfix = 0; npostfix <= 3; ++npostfix) {
    while (ndirect_msb < 16) {
      const uint32_t ndirect = ndirect_msb << npostfix;
      double dist_cost = 0.0;
      if (npostfix == d.dist_postfix_bits && ndirect == d.num_direct_distance_codes) check_orig = false;
      const bool skip = !hq_distance_cost(B, d, S, npostfix, ndirect, &dist_cost);
      if (skip || dist_cost > best_dist_cost) break;
      best_dist_cost = dist_cost;
      best_npostfix = npostfix;
      best_ndirect = ndirect;
      ndirect_msb += 1;
    }
    ndirect_msb = ndirect_msb > 0 ? ndirect_msb - 1 : 0;
    ndirect_msb /= 2;
  }
  if (check_orig) {
    double dist_cost = 0.0;
    hq_distance_cost(B, d, S, d.dist_postfix_bits, d.num_direct_distance_codes, &dist_cost);
    if (dist_cost < best_dist_cost) {
      best_npostfix = d.dist_postfix_bits;
      best_ndirect = d.num_direct_distance_codes;
    }
  }
  if (BR_TID == 0) {
    r.hq_postfix = best_npostfix;
    r.hq_ndirect = best_ndirect;
  }
  if (best_npostfix != d.dist_postfix_bits || best_ndirect != d.num_direct_distance_codes) {
    for (uint32_t c = d.cmd_offset + BR_TID; c < d.cmd_offset + d.n_cmds; c += BR_NT) {
      Command cmd = B.cmds_rw[c];
      if (!br_command_has_distance(cmd)) continue;
      const uint32_t code = hq_restore_distance_code(cmd, d.num_direct_distance_codes, d.dist_postfix_bits);
      hq_prefix_encode_distance(code, best_ndirect, best_npostfix, &cmd.dist_prefix_, &cmd.dist_extra_);
      B.cmds_rw[c] = cmd;
    }
  }
}

// BrotliIsMostlyUTF8 (utf8_util.rs:3-62) over the bytes ChooseContextMode really looks at (encode.rs:1357-1377 is handed
// the ring buffer ALLOCATION, whose data start two bytes in: the census runs over the stream shifted by two; positions in
// front of the stream read the zeroed slack).  Result: MbResult::hq_mostly_utf8.
BR_DEV uint32_t hq_census_byte(const MbBuffers& B, const MbDesc& d, uint32_t i) {
  const uint32_t p = d.start + i;  // the reference reads data_mo[p] = ring[p - 2] (a custom dictionary lies in the ring too)
  return p >= 2 ? B.text[p - 2] : 0u;
}
// BrotliParseAsUTF8 (utf8_util.rs:3-43) at offset i of a census of `length` bytes: bytes consumed, and whether they count
// as UTF-8 (symbol < 0x110000).  Returns bytes | valid << 3.
BR_DEV uint32_t hq_parse_utf8(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t size) {
  uint32_t bytes_read = 0;
  int32_t symbol = 0;
  if ((b0 & 0x80) == 0 && b0 > 0) {
    symbol = (int32_t)b0;
    bytes_read = 1;
  }
  if (bytes_read == 0 && size > 1 && (b0 & 0xe0) == 0xc0 && (b1 & 0xc0) == 0x80) {
    symbol = (int32_t)(((b0 & 0x1f) << 6) | (b1 & 0x3f));
    if (symbol > 0x7f) bytes_read = 2;
  }
  if (bytes_read == 0 && size > 2 && (b0 & 0xf0) == 0xe0 && (b1 & 0xc0) == 0x80 && (b2 & 0xc0) == 0x80) {
    symbol = (int32_t)(((b0 & 0x0f) << 12) | ((b1 & 0x3f) << 6) | (b2 & 0x3f));
    if (symbol > 0x7ff) bytes_read = 3;
  }
  if (bytes_read == 0 && size > 3 && (b0 & 0xf8) == 0xf0 && (b1 & 0xc0) == 0x80 && (b2 & 0xc0) == 0x80 && (b3 & 0xc0) == 0x80) {
    symbol = (int32_t)(((b0 & 0x07) << 18) | ((b1 & 0x3f) << 12) | ((b2 & 0x3f) << 6) | (b3 & 0x3f));
    if (symbol > 0xffff && symbol <= 0x10ffff) bytes_read = 4;
  }
  if (bytes_read == 0) {
    symbol = (int32_t)(0x110000u | b0);
    bytes_read = 1;
  }
  return bytes_read | (symbol < 0x110000 ? 8u : 0u);
}

// BrotliIsMostlyUTF8 (utf8_util.rs:45-69) walks the census from offset 0, one symbol after the other -- a chain of 8 M dependent
// steps for a lone 8 MiB meta-block, 485 ms on one lane (round 4: 37 % of such a call).  The walk is not needed.  What a parse
// that starts at offset o consumes depends on the (at most four) bytes there only, and it only ever SKIPS bytes it has checked to
// be continuation bytes (10xxxxxx) of a valid sequence: every other byte -- ASCII, a lead byte, a stray 0xf8..0xff, a zero -- is
// visited whatever came before it.  A continuation byte that IS visited parses as an invalid single byte and adds nothing.  So
// the sum over the visited offsets of "bytes consumed where the symbol is valid" equals the same sum over ALL offsets: the terms
// of unvisited offsets are those of continuation bytes, which are zero.  (Checked against the walk on 3 000 random and damaged
// UTF-8 strings before the change; integer sum, exact in any order.)
struct HqCensusScratch {
  uint32_t ctl[4];
};
BR_DEV void hq_item_utf8_census(const MbBuffers& B, uint32_t m, HqCensusScratch& S) {
  const MbDesc& d = B.descs[m];
  const uint32_t length = d.end - d.start;
  if (BR_TID == 0) S.ctl[0] = 0;
  BR_SYNC();
  uint32_t mine = 0;
  for (uint32_t o = BR_TID; o < length; o += BR_NT) {
    const uint32_t size = length - o;
    const uint32_t b0 = hq_census_byte(B, d, o);
    if ((b0 & 0xc0u) == 0x80u) continue;  // (a continuation byte: an invalid single byte if visited at all)
    const uint32_t b1 = size > 1 ? hq_census_byte(B, d, o + 1) : 0u;
    const uint32_t b2 = size > 2 ? hq_census_byte(B, d, o + 2) : 0u;
    const uint32_t b3 = size > 3 ? hq_census_byte(B, d, o + 3) : 0u;
    const uint32_t s = hq_parse_utf8(b0, b1, b2, b3, size);
    if (s & 8u) mine += s & 7u;
  }
  if (mine != 0) BR_ATOMIC_ADD_U32(&S.ctl[0], mine);
  BR_SYNC();
  if (BR_TID == 0) B.results[m].hq_mostly_utf8 = (float)S.ctl[0] > 0.75f * (float)length ? 1u : 0u;
}

// ---- symbol streams of the three splitters (CopyLiteralsToByteArray etc., block_splitter.rs:97-129, 860-927)
BR_DEV void hq_item_literal_symbol(const MbBuffers& B, uint32_t i) { B.hq_sym[kSplitLiteral][i] = B.text[B.lit_pos[i]]; }
BR_DEV void hq_item_command_symbols(const MbBuffers& B, uint32_t c) {
  const Command cmd = B.cmds[c];
  B.hq_sym[kSplitCommand][c] = cmd.cmd_prefix_;
  if (br_command_has_distance(cmd)) B.hq_sym[kSplitDistance][B.cmd_dist_index[c]] = cmd.dist_prefix_ & 0x3ffu;
}

// ---- BrotliBuildHistogramsWithContext (histogram.rs:465-534), one symbol per item, counters by atomic add
BR_DEV void hq_item_literal_context_count(const MbBuffers& B, uint32_t i) {
  const uint32_t m = mb_find_by_lit(B, i);
  const MbDesc& d = B.descs[m];
  if (d.uncompressed) return;
  const uint32_t local = i - d.lit_base;
  const uint32_t blk = hq_block_of(B.block_start[kSplitLiteral] + d.block_base[kSplitLiteral], B.results[m].num_blocks[kSplitLiteral], local);
  const uint32_t type = B.block_types[kSplitLiteral][d.block_base[kSplitLiteral] + blk];
  const uint32_t pos = B.lit_pos[i];
  const uint32_t row = d.hq_no_context ? type : (type << 6) + mb_literal_context(B, d, pos);
  BR_ATOMIC_ADD_U32(B.hq_ctx_histo[0] + ((size_t)d.hq_ctx_row_base[0] + row) * 256 + B.text[pos], 1u);
}
BR_DEV void hq_item_command_context_count(const MbBuffers& B, uint32_t c) {
  const uint32_t m = mb_find_by_cmd(B, c);
  const MbDesc& d = B.descs[m];
  if (d.uncompressed) return;
  const Command cmd = B.cmds[c];
  {
    const uint32_t local = c - d.cmd_offset;
    const uint32_t blk = hq_block_of(B.block_start[kSplitCommand] + d.block_base[kSplitCommand], B.results[m].num_blocks[kSplitCommand], local);
    const uint32_t type = B.block_types[kSplitCommand][d.block_base[kSplitCommand] + blk];
    BR_ATOMIC_ADD_U32(B.histo[kSplitCommand] + ((size_t)d.histo_base[kSplitCommand] + type) * kNumCommandSymbols + cmd.cmd_prefix_, 1u);
  }
  if (br_command_has_distance(cmd)) {
    const uint32_t local = B.cmd_dist_index[c] - d.dist_base;
    const uint32_t blk = hq_block_of(B.block_start[kSplitDistance] + d.block_base[kSplitDistance], B.results[m].num_blocks[kSplitDistance], local);
    const uint32_t type = B.block_types[kSplitDistance][d.block_base[kSplitDistance] + blk];
    const uint32_t row = (type << 2) + br_distance_context(cmd);
    BR_ATOMIC_ADD_U32(B.hq_ctx_histo[1] + ((size_t)d.hq_ctx_row_base[1] + row) * kNumDistanceHistoSymbols + (cmd.dist_prefix_ & 0x3ffu), 1u);
  }
}

// ---- BrotliClusterHistograms, cluster.rs:353-465 (+ HistogramRemap :261-297, HistogramReindex :310-351), and the copy of
// the result into the meta-block's histogram rows and context map.  Two launches: the batches of 64 input histograms side by
// side (one workgroup each, the pair queue in workgroup memory), then the rest per job.
BR_DEV void hq_item_cluster_histograms_batch(const EntropyTables& et, const HqClusterJob& J, uint32_t b, HqWaveScratch& S, HqPair* pairs) {
  const uint32_t in_size = J.in_size, len = J.len;
  const uint32_t i0 = b * kHqBatch;
  const uint32_t num_to_combine = in_size - i0 < kHqBatch ? in_size - i0 : kHqBatch;
  HqHistos inp;
  inp.data = const_cast<uint32_t*>(J.in_data);
  inp.total = J.in_total;
  inp.cost = J.out_cost;  // (not used for the inputs)
  inp.len = len;
  HqHistos out;
  out.data = J.out_data;
  out.total = J.out_total;
  out.cost = J.out_cost;
  out.len = len;
  for (uint32_t i = i0; i < i0 + num_to_combine; ++i) {
    // copy + total (an integer sum: order-free) + cost
    const uint32_t* s = inp.row(i);
    uint32_t* o = out.row(i);
    uint32_t part = 0;
    for (uint32_t k = BR_TID; k < len; k += BR_NT) {
      const uint32_t v = s[k];
      o[k] = v;
      S.tmp[k] = v;
      part += v;
    }
    if (BR_TID == 0) S.ctl[0] = 0;
    BR_SYNC();
    BR_ATOMIC_ADD_U32(&S.ctl[0], part);
    BR_SYNC();
    const uint32_t t = S.ctl[0];
    const float cost_i = hq_cost_in_tmp(et, S, len, t);
    if (BR_TID == 0) {
      J.in_total[i] = t;
      out.total[i] = t;
      out.cost[i] = cost_i;
      J.cluster_size[i] = 1;
      J.symbols[i] = i;
      J.clusters[i] = i;  // the batch's cluster list lives in its own stretch of the array until the batches are gathered
    }
    BR_SYNC();
  }
  const uint32_t num_new_clusters = hq_histogram_combine(et, S, out, J.cluster_size, J.symbols + i0, J.clusters + i0, pairs, num_to_combine,
                                                         num_to_combine, 256, kHqBatchPairs);
  if (BR_TID == 0) J.batch_count[b] = num_new_clusters;
}

BR_DEV void hq_item_cluster_histograms(const MbBuffers& B, const HqClusterJob& J, HqWaveScratch& S) {
  const EntropyTables& et = B.et;
  const MbDesc& d = B.descs[J.m];
  MbResult& r = B.results[J.m];
  const uint32_t which = J.kind == kSplitLiteral ? 0u : 1u;
  const uint32_t in_size = J.in_size, len = J.len;
  uint32_t* map = B.hq_ctx_map[which] + d.hq_ctx_map_base[which];
  HqHistos inp;
  inp.data = const_cast<uint32_t*>(J.in_data);
  inp.total = J.in_total;
  inp.cost = J.out_cost;  // (not used for the inputs)
  inp.len = len;
  HqHistos out;
  out.data = J.out_data;
  out.total = J.out_total;
  out.cost = J.out_cost;
  out.len = len;
  uint32_t* cluster_size = J.cluster_size;
  uint32_t* clusters = J.clusters;
  uint32_t* symbols = J.symbols;
  uint32_t num_clusters = 0;
  for (uint32_t i = 0; i < in_size; i += kHqBatch) {  // gather the batches' cluster lists (compact, in batch order)
    const uint32_t cnt = J.batch_count[i / kHqBatch];
    for (uint32_t j = 0; j < cnt; ++j) clusters[num_clusters + j] = clusters[i + j];
    num_clusters += cnt;
  }
  BR_SYNC();
  {
    uint32_t max_num_pairs = 64u * num_clusters;
    const uint64_t alt = (uint64_t)(num_clusters / 2) * num_clusters;
    if (alt < max_num_pairs) max_num_pairs = (uint32_t)alt;
    num_clusters = hq_histogram_combine(et, S, out, cluster_size, symbols, clusters, J.pairs, num_clusters, in_size, 256, max_num_pairs);
  }
  // BrotliHistogramRemap
  for (uint32_t i = 0; i < in_size; ++i) {
    const uint32_t* s = inp.row(i);
    for (uint32_t k = BR_TID; k < len; k += BR_NT) S.blk[k] = s[k];
    BR_SYNC();
    const uint32_t t = J.in_total[i];
    uint32_t best_out = i == 0 ? symbols[0] : symbols[i - 1];
    float best_bits = hq_bit_cost_distance(et, S, t, out, best_out);
    for (uint32_t j = 0; j < num_clusters; ++j) {
      const float cur_bits = hq_bit_cost_distance(et, S, t, out, clusters[j]);
      if (cur_bits < best_bits) {
        best_bits = cur_bits;
        best_out = clusters[j];
      }
    }
    symbols[i] = best_out;
    BR_SYNC();
  }
  for (uint32_t i = 0; i < num_clusters; ++i) hq_clear(out, clusters[i]);
  for (uint32_t i = 0; i < in_size; ++i) hq_add(out, symbols[i], inp, i);
  // BrotliHistogramReindex
  uint32_t next_index = 0;
  {
    const uint32_t kInvalidIndex = 0xffffffffu;
    uint32_t* new_index = J.new_index;
    for (uint32_t i = 0; i < in_size; ++i) new_index[i] = kInvalidIndex;
    for (uint32_t i = 0; i < in_size; ++i) {
      if (new_index[symbols[i]] == kInvalidIndex) {
        new_index[symbols[i]] = next_index;
        ++next_index;
      }
    }
    BR_SYNC();
    HqHistos re;
    re.data = J.reindex_data;
    re.total = J.reindex_total;
    re.cost = J.reindex_cost;
    re.len = len;
    next_index = 0;
    for (uint32_t i = 0; i < in_size; ++i) {
      const uint32_t sym = symbols[i];
      if (new_index[sym] == next_index) {
        hq_copy(re, next_index, out, sym);
        ++next_index;
      }
      BR_SYNC();  // (every lane has read symbols[i] before it is rewritten)
      symbols[i] = new_index[sym];
    }
    BR_SYNC();
    // into the meta-block's histogram rows
    uint32_t* H = B.histo[J.kind] + (size_t)d.histo_base[J.kind] * len;
    for (uint32_t i = BR_TID; i < next_index * len; i += BR_NT) H[i] = re.data[i];
  }
  if (BR_TID == 0) r.num_histos[J.kind] = next_index;
  if (J.expand64) {
    for (uint32_t i = BR_TID; i < in_size * 64; i += BR_NT) map[i] = symbols[i >> 6];
  } else {
    for (uint32_t i = BR_TID; i < in_size; i += BR_NT) map[i] = symbols[i];
  }
}

}  // namespace brotli_mi355x
#endif
