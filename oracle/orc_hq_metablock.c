/* oracle/orc_hq_metablock.c -- CPU restatement of rust-brotli's quality >= 10 meta-block builder.
 * TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows:
 *   src/enc/bit_cost.rs:76-211      BrotliPopulationCost (the build without "vector_scratch_space")
 *   src/enc/cluster.rs:33-467       histogram pairs, BrotliHistogramCombine / Remap / Reindex, BrotliClusterHistograms
 *   src/enc/block_splitter.rs       InitialEntropyCodes, RefineEntropyCodes, FindBlocks, ClusterBlocks,
 *                                   SplitByteVector, BrotliSplitBlock
 *   src/enc/histogram.rs:465-534    BrotliBuildHistogramsWithContext
 *   src/enc/metablock.rs:28-307     distance-parameter search + BrotliBuildMetaBlock
 * floatX = f32 (the default build: no "float64" feature), evaluated left to right; compile with -ffp-contract=off.
 *
 * Pins (tests/test_oracle.py): src/bin/integration_tests.rs:397-428 -- random_then_unicode through this builder behind
 * the greedy LZ77 stage ("quality 9.5": q10 + q9_5 -> 130 036 bytes, q11 + q9_5 -> 129 715 bytes) and, with the Zopfli
 * stage of orc_zopfli.c in front, alice29 at q10 / q11 -> 47 488 / 46 493 bytes.
 */
#include <math.h>
#include <assert.h>
#include <stdio.h>

#include "orc_internal.h"

#define HQ_MAX_SYMBOLS 704

typedef struct {
  uint32_t data_[HQ_MAX_SYMBOLS];
  size_t total_count_;
  float bit_cost_;
} Histo;

static void histo_clear(Histo* h, size_t len) { /* histogram.rs:391-399 */
  memset(h->data_, 0, len * sizeof(uint32_t));
  h->total_count_ = 0;
  h->bit_cost_ = 3.402e+38f;
}
static void histo_copy(Histo* dst, const Histo* src, size_t len) {
  memcpy(dst->data_, src->data_, len * sizeof(uint32_t));
  dst->total_count_ = src->total_count_;
  dst->bit_cost_ = src->bit_cost_;
}
static void histo_add_histo(Histo* dst, const Histo* src, size_t len) { /* histogram.rs HistogramAddHistogram */
  dst->total_count_ += src->total_count_;
  for (size_t i = 0; i < len; ++i) dst->data_[i] += src->data_[i];
}
static inline void histo_add(Histo* h, size_t sym) {
  h->data_[sym]++;
  h->total_count_++;
}
static void histo_add_vector(Histo* h, const uint16_t* p, size_t n) { /* histogram.rs:370-388 */
  h->total_count_ += n;
  for (size_t i = 0; i < n; ++i) h->data_[p[i]]++;
}

/* bit_cost.rs:76-211 */
float orc_population_cost(const uint32_t* data, size_t data_size, size_t total_count) {
  const float kOneSymbolHistogramCost = 12.0f, kTwoSymbolHistogramCost = 20.0f, kThreeSymbolHistogramCost = 28.0f,
              kFourSymbolHistogramCost = 37.0f;
  const float* l16 = orc_logs_16();
  size_t count = 0;
  size_t s[5] = {0, 0, 0, 0, 0};
  float bits = 0.0f;
  if (total_count == 0) return kOneSymbolHistogramCost;
  for (size_t i = 0; i < data_size; ++i) {
    if (data[i] > 0) {
      s[count] = i;
      count++;
      if (count > 4) break;
    }
  }
  switch (count) {
    case 1: return kOneSymbolHistogramCost;
    case 2: return kTwoSymbolHistogramCost + (float)total_count;
    case 3: {
      uint32_t histo0 = data[s[0]], histo1 = data[s[1]], histo2 = data[s[2]];
      uint32_t histomax = ORC_MAX(histo0, ORC_MAX(histo1, histo2));
      return kThreeSymbolHistogramCost + (float)(2u * (histo0 + histo1 + histo2)) - (float)histomax;
    }
    case 4: {
      uint32_t histo[4];
      for (int i = 0; i < 4; ++i) histo[i] = data[s[i]];
      for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
          if (histo[j] > histo[i]) {
            uint32_t t = histo[j];
            histo[j] = histo[i];
            histo[i] = t;
          }
      uint32_t h23 = histo[2] + histo[3];
      uint32_t histomax = ORC_MAX(h23, histo[0]);
      return kFourSymbolHistogramCost + (float)(3u * h23) + (float)(2u * (histo[0] + histo[1])) - (float)histomax;
    }
    default: break;
  }
  {
    size_t max_depth = 1;
    uint32_t depth_histo[18];
    float log2total = orc_fast_log2(total_count);
    uint32_t reps = 0;
    memset(depth_histo, 0, sizeof(depth_histo));
    for (size_t i = 0; i < data_size; ++i) {
      uint32_t histo = data[i];
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
        float log2p = log2total - l16[(uint16_t)histo];
        float d = log2p + 0.5f;
        size_t depth = (d > 0.0f) ? (size_t)d : 0; /* `as usize` saturates */
        bits += (float)histo * log2p;
        depth = ORC_MIN(depth, (size_t)15);
        max_depth = ORC_MAX(depth, max_depth);
        depth_histo[depth] += 1;
      } else {
        reps += 1;
      }
    }
    bits += (float)(18 + 2 * max_depth);
    bits += orc_bits_entropy_impl(depth_histo, 18);
  }
  return bits;
}
static float population_cost(const Histo* h, size_t len) { return orc_population_cost(h->data_, len, h->total_count_); }

/* ------------------------------------------------------------------ cluster.rs */
typedef struct {
  uint32_t idx1, idx2;
  float cost_combo, cost_diff;
} HistogramPair;

/* cluster.rs:33-39 */
static float cluster_cost_diff(size_t size_a, size_t size_b) {
  size_t size_c = size_a + size_b;
  return (float)size_a * orc_fast_log2(size_a) + (float)size_b * orc_fast_log2(size_b) -
         (float)size_c * orc_fast_log2(size_c);
}
/* cluster.rs:41-48 */
static int histogram_pair_is_less(const HistogramPair* p1, const HistogramPair* p2) {
  if (p1->cost_diff != p2->cost_diff) return p1->cost_diff > p2->cost_diff;
  return (uint32_t)(p1->idx2 - p1->idx1) > (uint32_t)(p2->idx2 - p2->idx1);
}

/* cluster.rs:52-121 */
static void compare_and_push_to_queue(const Histo* out, size_t len, const uint32_t* cluster_size, uint32_t idx1,
                                      uint32_t idx2, size_t max_num_pairs, HistogramPair* pairs, size_t* num_pairs,
                                      Histo* combo) {
  int is_good_pair = 0;
  HistogramPair p = {0, 0, 0.0f, 0.0f};
  if (idx1 == idx2) return;
  if (idx2 < idx1) {
    uint32_t t = idx2;
    idx2 = idx1;
    idx1 = t;
  }
  p.idx1 = idx1;
  p.idx2 = idx2;
  p.cost_diff = 0.5f * cluster_cost_diff(cluster_size[idx1], cluster_size[idx2]);
  p.cost_diff -= out[idx1].bit_cost_;
  p.cost_diff -= out[idx2].bit_cost_;
  if (out[idx1].total_count_ == 0) {
    p.cost_combo = out[idx2].bit_cost_;
    is_good_pair = 1;
  } else if (out[idx2].total_count_ == 0) {
    p.cost_combo = out[idx1].bit_cost_;
    is_good_pair = 1;
  } else {
    float threshold = (*num_pairs == 0) ? 1e38f : fmaxf(pairs[0].cost_diff, 0.0f);
    histo_copy(combo, &out[idx1], len);
    histo_add_histo(combo, &out[idx2], len);
    float cost_combo = population_cost(combo, len);
    if (cost_combo < threshold - p.cost_diff) {
      p.cost_combo = cost_combo;
      is_good_pair = 1;
    }
  }
  if (is_good_pair) {
    p.cost_diff += p.cost_combo;
    if (*num_pairs > 0 && histogram_pair_is_less(&pairs[0], &p)) {
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

/* cluster.rs:123-236 */
static size_t histogram_combine(Histo* out, size_t len, uint32_t* cluster_size, uint32_t* symbols, uint32_t* clusters,
                                HistogramPair* pairs, size_t num_clusters, size_t symbols_size, size_t max_clusters,
                                size_t max_num_pairs, Histo* scratch) {
  float cost_diff_threshold = 0.0f;
  size_t min_cluster_size = 1;
  size_t num_pairs = 0;
  for (size_t idx1 = 0; idx1 < num_clusters; ++idx1)
    for (size_t idx2 = idx1 + 1; idx2 < num_clusters; ++idx2)
      compare_and_push_to_queue(out, len, cluster_size, clusters[idx1], clusters[idx2], max_num_pairs, pairs, &num_pairs,
                                scratch);
  while (num_clusters > min_cluster_size) {
    if (pairs[0].cost_diff >= cost_diff_threshold) {
      cost_diff_threshold = 1e38f;
      min_cluster_size = max_clusters;
      continue;
    }
    uint32_t best_idx1 = pairs[0].idx1, best_idx2 = pairs[0].idx2;
    histo_add_histo(&out[best_idx1], &out[best_idx2], len);
    out[best_idx1].bit_cost_ = pairs[0].cost_combo;
    cluster_size[best_idx1] += cluster_size[best_idx2];
    for (size_t i = 0; i < symbols_size; ++i)
      if (symbols[i] == best_idx2) symbols[i] = best_idx1;
    for (size_t i = 0; i < num_clusters; ++i) {
      if (clusters[i] == best_idx2) {
        memmove(&clusters[i], &clusters[i + 1], (num_clusters - i - 1) * sizeof(uint32_t));
        break;
      }
    }
    --num_clusters;
    {
      size_t copy_to_idx = 0;
      for (size_t i = 0; i < num_pairs; ++i) {
        HistogramPair p = pairs[i];
        if (p.idx1 == best_idx1 || p.idx2 == best_idx1 || p.idx1 == best_idx2 || p.idx2 == best_idx2) continue;
        if (histogram_pair_is_less(&pairs[0], &p)) {
          HistogramPair front = pairs[0];
          pairs[0] = p;
          pairs[copy_to_idx] = front;
        } else {
          pairs[copy_to_idx] = p;
        }
        ++copy_to_idx;
      }
      num_pairs = copy_to_idx;
    }
    for (size_t i = 0; i < num_clusters; ++i)
      compare_and_push_to_queue(out, len, cluster_size, best_idx1, clusters[i], max_num_pairs, pairs, &num_pairs,
                                scratch);
  }
  return num_clusters;
}

/* cluster.rs:238-254 */
static float histogram_bit_cost_distance(const Histo* histogram, const Histo* candidate, size_t len, Histo* tmp) {
  if (histogram->total_count_ == 0) return 0.0f;
  histo_copy(tmp, histogram, len);
  histo_add_histo(tmp, candidate, len);
  return population_cost(tmp, len) - candidate->bit_cost_;
}

/* cluster.rs:261-297 */
static void histogram_remap(const Histo* inp, size_t in_size, const uint32_t* clusters, size_t num_clusters, size_t len,
                            Histo* out, uint32_t* symbols, Histo* tmp) {
  for (size_t i = 0; i < in_size; ++i) {
    uint32_t best_out = (i == 0) ? symbols[0] : symbols[i - 1];
    float best_bits = histogram_bit_cost_distance(&inp[i], &out[best_out], len, tmp);
    for (size_t j = 0; j < num_clusters; ++j) {
      float cur_bits = histogram_bit_cost_distance(&inp[i], &out[clusters[j]], len, tmp);
      if (cur_bits < best_bits) {
        best_bits = cur_bits;
        best_out = clusters[j];
      }
    }
    symbols[i] = best_out;
  }
  for (size_t i = 0; i < num_clusters; ++i) histo_clear(&out[clusters[i]], len);
  for (size_t i = 0; i < in_size; ++i) histo_add_histo(&out[symbols[i]], &inp[i], len);
}

/* cluster.rs:310-351 */
static size_t histogram_reindex(Histo* out, size_t len, uint32_t* symbols, size_t length) {
  const uint32_t kInvalidIndex = 0xffffffffu;
  uint32_t* new_index = (uint32_t*)malloc((length ? length : 1) * sizeof(uint32_t));
  uint32_t next_index = 0;
  for (size_t i = 0; i < length; ++i) new_index[i] = kInvalidIndex;
  for (size_t i = 0; i < length; ++i) {
    if (new_index[symbols[i]] == kInvalidIndex) {
      new_index[symbols[i]] = next_index;
      ++next_index;
    }
  }
  Histo* tmp = (Histo*)malloc((next_index ? next_index : 1) * sizeof(Histo));
  next_index = 0;
  for (size_t i = 0; i < length; ++i) {
    if (new_index[symbols[i]] == next_index) {
      histo_copy(&tmp[next_index], &out[symbols[i]], len);
      ++next_index;
    }
    symbols[i] = new_index[symbols[i]];
  }
  free(new_index);
  for (size_t i = 0; i < next_index; ++i) histo_copy(&out[i], &tmp[i], len);
  free(tmp);
  return next_index;
}

/* cluster.rs:353-465.  `out` has room for in_size histograms. */
static void cluster_histograms(const Histo* inp, size_t in_size, size_t max_histograms, size_t len, Histo* out,
                               size_t* out_size, uint32_t* histogram_symbols) {
  uint32_t* cluster_size = (uint32_t*)calloc(in_size ? in_size : 1, sizeof(uint32_t));
  uint32_t* clusters = (uint32_t*)calloc(in_size ? in_size : 1, sizeof(uint32_t));
  size_t num_clusters = 0;
  const size_t max_input_histograms = 64;
  size_t pairs_capacity = max_input_histograms * max_input_histograms / 2;
  HistogramPair* pairs = (HistogramPair*)calloc(pairs_capacity + 1, sizeof(HistogramPair));
  Histo* scratch = (Histo*)malloc(sizeof(Histo));
  for (size_t i = 0; i < in_size; ++i) cluster_size[i] = 1;
  for (size_t i = 0; i < in_size; ++i) {
    histo_copy(&out[i], &inp[i], len);
    out[i].bit_cost_ = population_cost(&inp[i], len);
    histogram_symbols[i] = (uint32_t)i;
  }
  for (size_t i = 0; i < in_size; i += max_input_histograms) {
    size_t num_to_combine = ORC_MIN(in_size - i, max_input_histograms);
    for (size_t j = 0; j < num_to_combine; ++j) clusters[num_clusters + j] = (uint32_t)(i + j);
    size_t num_new_clusters =
        histogram_combine(out, len, cluster_size, &histogram_symbols[i], &clusters[num_clusters], pairs, num_to_combine,
                          num_to_combine, max_histograms, pairs_capacity, scratch);
    num_clusters += num_new_clusters;
  }
  {
    size_t max_num_pairs = ORC_MIN(64 * num_clusters, (num_clusters / 2) * num_clusters);
    if (pairs_capacity < max_num_pairs + 1) {
      size_t new_size = pairs_capacity;
      while (new_size < max_num_pairs + 1) new_size *= 2;
      HistogramPair* np = (HistogramPair*)calloc(new_size, sizeof(HistogramPair));
      memcpy(np, pairs, pairs_capacity * sizeof(HistogramPair));
      free(pairs);
      pairs = np;
    }
    num_clusters = histogram_combine(out, len, cluster_size, histogram_symbols, clusters, pairs, num_clusters, in_size,
                                     max_histograms, max_num_pairs, scratch);
  }
  free(pairs);
  free(cluster_size);
  histogram_remap(inp, in_size, clusters, num_clusters, len, out, histogram_symbols, scratch);
  free(clusters);
  free(scratch);
  *out_size = histogram_reindex(out, len, histogram_symbols, in_size);
}

/* ------------------------------------------------------------------ block_splitter.rs */
static const size_t kMaxLiteralHistograms = 100, kMaxCommandHistograms = 50;
static const float kLiteralBlockSwitchCost = 28.1f, kCommandBlockSwitchCost = 13.5f, kDistanceBlockSwitchCost = 14.6f;
static const size_t kLiteralStrideLength = 70, kCommandStrideLength = 40;
static const size_t kSymbolsPerLiteralHistogram = 544, kSymbolsPerCommandHistogram = 530,
                    kSymbolsPerDistanceHistogram = 544;
static const size_t kMinLengthForBlockSplitting = 128, kIterMulForRefining = 2, kMinItersForRefining = 100;

/* block_splitter.rs:131-137 */
static uint32_t my_rand(uint32_t* seed) {
  *seed = *seed * 16807u;
  if (*seed == 0) *seed = 1;
  return *seed;
}

/* block_splitter.rs:139-165 */
static void initial_entropy_codes(const uint16_t* data, size_t length, size_t stride, size_t num_histograms,
                                  Histo* histograms, size_t len) {
  uint32_t seed = 7;
  size_t block_length = length / num_histograms;
  for (size_t i = 0; i < num_histograms; ++i) histo_clear(&histograms[i], len);
  for (size_t i = 0; i < num_histograms; ++i) {
    size_t pos = length * i / num_histograms;
    if (i != 0) pos += (size_t)my_rand(&seed) % block_length;
    if (pos + stride >= length) pos = length - stride - 1;
    histo_add_vector(&histograms[i], data + pos, stride);
  }
}

/* block_splitter.rs:167-188 */
static void random_sample(uint32_t* seed, const uint16_t* data, size_t length, size_t stride, Histo* sample) {
  size_t pos;
  if (stride >= length) {
    pos = 0;
    stride = length;
  } else {
    pos = (size_t)my_rand(seed) % (length - stride + 1);
  }
  histo_add_vector(sample, data + pos, stride);
}

/* block_splitter.rs:190-222 */
static void refine_entropy_codes(const uint16_t* data, size_t length, size_t stride, size_t num_histograms,
                                 Histo* histograms, size_t len) {
  size_t iters = kIterMulForRefining * length / stride + kMinItersForRefining;
  uint32_t seed = 7;
  Histo* sample = (Histo*)malloc(sizeof(Histo));
  iters = (iters + num_histograms - 1) / num_histograms * num_histograms;
  for (size_t iter = 0; iter < iters; ++iter) {
    histo_clear(sample, len);
    random_sample(&seed, data, length, stride, sample);
    histo_add_histo(&histograms[iter % num_histograms], sample, len);
  }
  free(sample);
}

/* block_splitter.rs:224-230 */
static float bit_cost(size_t count) { return count == 0 ? -2.0f : orc_fast_log2(count); }

/* block_splitter.rs:232-350 (+ update_cost_and_signal :46-82) */
static size_t find_blocks(const uint16_t* data, size_t length, float block_switch_bitcost, size_t num_histograms,
                          const Histo* histograms, size_t data_size, float* insert_cost, float* cost /* padded to 8 */,
                          uint8_t* switch_signal, uint8_t* block_id) {
  size_t bitmaplen = (num_histograms + 7) >> 3;
  size_t num_blocks = 1;
  size_t padded = bitmaplen << 3;
  if (num_histograms == 0) return 0;
  if (num_histograms <= 1) {
    for (size_t i = 0; i < length; ++i) block_id[i] = 0;
    return 1;
  }
  for (size_t i = 0; i < data_size * num_histograms; ++i) insert_cost[i] = 0.0f;
  for (size_t i = 0; i < num_histograms; ++i) insert_cost[i] = orc_fast_log2((uint64_t)(uint32_t)histograms[i].total_count_);
  for (size_t i = data_size; i != 0;) {
    --i;
    for (size_t j = 0; j < num_histograms; ++j)
      insert_cost[i * num_histograms + j] = insert_cost[j] - bit_cost(histograms[j].data_[i]);
  }
  for (size_t i = 0; i < padded; ++i) cost[i] = 0.0f;
  memset(switch_signal, 0, length * bitmaplen);
  for (size_t byte_ix = 0; byte_ix < length; ++byte_ix) {
    size_t ix = byte_ix * bitmaplen;
    size_t insert_cost_ix = (size_t)data[byte_ix] * num_histograms;
    float min_cost = 1e38f;
    float block_switch_cost = block_switch_bitcost;
    for (size_t k = 0; k < num_histograms; ++k) {
      cost[k] += insert_cost[insert_cost_ix + k];
      if (cost[k] < min_cost) {
        min_cost = cost[k];
        block_id[byte_ix] = (uint8_t)k;
      }
    }
    if (byte_ix < 2000) block_switch_cost *= (0.77f + 0.07f * (float)byte_ix / 2000.0f);
    /* update_cost_and_signal: every lane of the padded vectors (lanes >= num_histograms never feed block_id) */
    for (size_t k = 0; k < padded; ++k) {
      float d = cost[k] - min_cost;
      if (d >= block_switch_cost) switch_signal[ix + (k >> 3)] |= (uint8_t)(1u << (k & 7));
      cost[k] = (d < block_switch_cost) ? d : block_switch_cost; /* simd_min */
    }
  }
  {
    size_t byte_ix = length - 1;
    size_t ix = byte_ix * bitmaplen;
    uint8_t cur_id = block_id[byte_ix];
    while (byte_ix > 0) {
      uint8_t mask = (uint8_t)(1u << (cur_id & 7));
      --byte_ix;
      ix -= bitmaplen;
      if ((switch_signal[ix + (cur_id >> 3)] & mask) != 0 && cur_id != block_id[byte_ix]) {
        cur_id = block_id[byte_ix];
        ++num_blocks;
      }
      block_id[byte_ix] = cur_id;
    }
  }
  return num_blocks;
}

/* block_splitter.rs:352-378 */
static size_t remap_block_ids(uint8_t* block_ids, size_t length, uint16_t* new_id, size_t num_histograms) {
  const uint16_t kInvalidId = 256;
  uint16_t next_id = 0;
  for (size_t i = 0; i < num_histograms; ++i) new_id[i] = kInvalidId;
  for (size_t i = 0; i < length; ++i)
    if (new_id[block_ids[i]] == kInvalidId) new_id[block_ids[i]] = next_id++;
  for (size_t i = 0; i < length; ++i) block_ids[i] = (uint8_t)new_id[block_ids[i]];
  return next_id;
}

/* block_splitter.rs:380-400 */
static void build_block_histograms(const uint16_t* data, size_t length, const uint8_t* block_ids, size_t num_histograms,
                                   Histo* histograms, size_t len) {
  for (size_t i = 0; i < num_histograms; ++i) histo_clear(&histograms[i], len);
  for (size_t i = 0; i < length; ++i) histo_add(&histograms[block_ids[i]], data[i]);
}

static void split_reserve(BlockSplit* split, size_t n) {
  split->types = (uint8_t*)realloc(split->types, n ? n : 1);
  split->lengths = (uint32_t*)realloc(split->lengths, (n ? n : 1) * sizeof(uint32_t));
}

/* block_splitter.rs:402-688 */
static void cluster_blocks(const uint16_t* data, size_t length, size_t num_blocks, uint8_t* block_ids, size_t len,
                           BlockSplit* split) {
  uint32_t* histogram_symbols = (uint32_t*)calloc(num_blocks, sizeof(uint32_t));
  uint32_t* block_lengths = (uint32_t*)calloc(num_blocks, sizeof(uint32_t));
  size_t expected_num_clusters = 16 * (num_blocks + 64 - 1) / 64;
  size_t all_histograms_size = 0, all_histograms_capacity = expected_num_clusters;
  Histo* all_histograms = (Histo*)malloc((all_histograms_capacity ? all_histograms_capacity : 1) * sizeof(Histo));
  size_t cluster_size_size = 0, cluster_size_capacity = expected_num_clusters;
  uint32_t* cluster_size = (uint32_t*)calloc(cluster_size_capacity ? cluster_size_capacity : 1, sizeof(uint32_t));
  size_t num_clusters = 0;
  Histo* histograms = (Histo*)malloc(ORC_MIN(num_blocks, (size_t)64) * sizeof(Histo));
  size_t max_num_pairs = 64 * 64 / 2;
  size_t pairs_capacity = max_num_pairs + 1;
  HistogramPair* pairs = (HistogramPair*)calloc(pairs_capacity, sizeof(HistogramPair));
  Histo* scratch = (Histo*)malloc(sizeof(Histo));
  size_t pos = 0;
  const uint32_t kInvalidIndex = 0xffffffffu;
  uint32_t sizes[64], new_clusters[64], symbols[64], remap[64];
  memset(sizes, 0, sizeof(sizes));
  memset(new_clusters, 0, sizeof(new_clusters));
  memset(symbols, 0, sizeof(symbols));
  memset(remap, 0, sizeof(remap));
  {
    size_t block_idx = 0;
    for (size_t i = 0; i < length; ++i) {
      block_lengths[block_idx]++;
      if (i + 1 == length || block_ids[i] != block_ids[i + 1]) ++block_idx;
    }
  }
  for (size_t i = 0; i < num_blocks; i += 64) {
    size_t num_to_combine = ORC_MIN(num_blocks - i, (size_t)64);
    for (size_t j = 0; j < num_to_combine; ++j) {
      histo_clear(&histograms[j], len);
      for (size_t k = 0; k < block_lengths[i + j]; ++k) histo_add(&histograms[j], data[pos++]);
      histograms[j].bit_cost_ = population_cost(&histograms[j], len);
      new_clusters[j] = (uint32_t)j;
      symbols[j] = (uint32_t)j;
      sizes[j] = 1;
    }
    size_t num_new_clusters = histogram_combine(histograms, len, sizes, symbols, new_clusters, pairs, num_to_combine,
                                                num_to_combine, 64, max_num_pairs, scratch);
    if (all_histograms_capacity < all_histograms_size + num_new_clusters) {
      size_t ns = all_histograms_capacity == 0 ? all_histograms_size + num_new_clusters : all_histograms_capacity;
      while (ns < all_histograms_size + num_new_clusters) ns *= 2;
      all_histograms = (Histo*)realloc(all_histograms, ns * sizeof(Histo));
      all_histograms_capacity = ns;
    }
    if (cluster_size_capacity < cluster_size_size + num_new_clusters) {
      size_t ns = cluster_size_capacity == 0 ? cluster_size_size + num_new_clusters : cluster_size_capacity;
      while (ns < cluster_size_size + num_new_clusters) ns *= 2;
      cluster_size = (uint32_t*)realloc(cluster_size, ns * sizeof(uint32_t));
      cluster_size_capacity = ns;
    }
    for (size_t j = 0; j < num_new_clusters; ++j) {
      histo_copy(&all_histograms[all_histograms_size++], &histograms[new_clusters[j]], len);
      cluster_size[cluster_size_size++] = sizes[new_clusters[j]];
      remap[new_clusters[j]] = (uint32_t)j;
    }
    for (size_t j = 0; j < num_to_combine; ++j) histogram_symbols[i + j] = (uint32_t)num_clusters + remap[symbols[j]];
    num_clusters += num_new_clusters;
  }
  free(histograms);
  max_num_pairs = ORC_MIN(64 * num_clusters, (num_clusters / 2) * num_clusters);
  if (pairs_capacity < max_num_pairs + 1) {
    free(pairs);
    pairs = (HistogramPair*)calloc(max_num_pairs + 1, sizeof(HistogramPair));
  }
  uint32_t* clusters = (uint32_t*)malloc((num_clusters ? num_clusters : 1) * sizeof(uint32_t));
  for (size_t i = 0; i < num_clusters; ++i) clusters[i] = (uint32_t)i;
  size_t num_final_clusters = histogram_combine(all_histograms, len, cluster_size, histogram_symbols, clusters, pairs,
                                                num_clusters, num_blocks, 256, max_num_pairs, scratch);
  free(pairs);
  free(cluster_size);
  uint32_t* new_index = (uint32_t*)malloc((num_clusters ? num_clusters : 1) * sizeof(uint32_t));
  for (size_t i = 0; i < num_clusters; ++i) new_index[i] = kInvalidIndex;
  pos = 0;
  {
    uint32_t next_index = 0;
    Histo* histo = (Histo*)malloc(sizeof(Histo));
    for (size_t i = 0; i < num_blocks; ++i) {
      histo_clear(histo, len);
      for (size_t j = 0; j < block_lengths[i]; ++j) histo_add(histo, data[pos++]);
      uint32_t best_out = (i == 0) ? histogram_symbols[0] : histogram_symbols[i - 1];
      float best_bits = histogram_bit_cost_distance(histo, &all_histograms[best_out], len, scratch);
      for (size_t j = 0; j < num_final_clusters; ++j) {
        float cur_bits = histogram_bit_cost_distance(histo, &all_histograms[clusters[j]], len, scratch);
        if (cur_bits < best_bits) {
          best_bits = cur_bits;
          best_out = clusters[j];
        }
      }
      histogram_symbols[i] = best_out;
      if (new_index[best_out] == kInvalidIndex) new_index[best_out] = next_index++;
    }
    free(histo);
  }
  free(clusters);
  free(all_histograms);
  free(scratch);
  split_reserve(split, num_blocks);
  {
    uint32_t cur_length = 0;
    size_t block_idx = 0;
    uint8_t max_type = 0;
    for (size_t i = 0; i < num_blocks; ++i) {
      cur_length += block_lengths[i];
      if (i + 1 == num_blocks || histogram_symbols[i] != histogram_symbols[i + 1]) {
        uint8_t id = (uint8_t)new_index[histogram_symbols[i]];
        split->types[block_idx] = id;
        split->lengths[block_idx] = cur_length;
        max_type = ORC_MAX(max_type, id);
        cur_length = 0;
        ++block_idx;
      }
    }
    split->num_blocks = block_idx;
    split->num_types = (size_t)max_type + 1;
  }
  free(new_index);
  free(block_lengths);
  free(histogram_symbols);
}

/* block_splitter.rs:690-838 */
static void split_byte_vector(const uint16_t* data, size_t length, size_t literals_per_histogram, size_t max_histograms,
                              size_t sampling_stride_length, float block_switch_cost, int quality, size_t data_size,
                              BlockSplit* split) {
  size_t num_histograms = length / literals_per_histogram + 1;
  if (num_histograms > max_histograms) num_histograms = max_histograms;
  if (length == 0) {
    split->num_types = 1;
    return;
  } else if (length < kMinLengthForBlockSplitting) {
    split_reserve(split, split->num_blocks + 1);
    split->num_types = 1;
    split->types[split->num_blocks] = 0;
    split->lengths[split->num_blocks] = (uint32_t)length;
    split->num_blocks++;
    return;
  }
  Histo* histograms = (Histo*)malloc(num_histograms * sizeof(Histo));
  initial_entropy_codes(data, length, sampling_stride_length, num_histograms, histograms, data_size);
  refine_entropy_codes(data, length, sampling_stride_length, num_histograms, histograms, data_size);
  {
    uint8_t* block_ids = (uint8_t*)calloc(length, 1);
    size_t num_blocks = 0;
    size_t bitmaplen = (num_histograms + 7) >> 3;
    float* insert_cost = (float*)calloc(data_size * num_histograms, sizeof(float));
    float* cost = (float*)calloc(bitmaplen << 3, sizeof(float));
    uint8_t* switch_signal = (uint8_t*)calloc(length * bitmaplen, 1);
    uint16_t* new_id = (uint16_t*)calloc(num_histograms, sizeof(uint16_t));
    size_t iters = quality <= 11 ? 3 : 10;
    for (size_t i = 0; i < iters; ++i) {
      num_blocks = find_blocks(data, length, block_switch_cost, num_histograms, histograms, data_size, insert_cost, cost,
                               switch_signal, block_ids);
      num_histograms = remap_block_ids(block_ids, length, new_id, num_histograms);
      build_block_histograms(data, length, block_ids, num_histograms, histograms, data_size);
    }
    free(insert_cost);
    free(cost);
    free(switch_signal);
    free(new_id);
    free(histograms);
    cluster_blocks(data, length, num_blocks, block_ids, data_size, split);
    free(block_ids);
  }
}

/* block_splitter.rs:840-929 */
static void split_block(const Command* cmds, size_t num_commands, const uint8_t* data, size_t pos, size_t mask,
                        int quality, BlockSplit* literal_split, BlockSplit* insert_and_copy_split,
                        BlockSplit* dist_split) {
  {
    size_t literals_count = 0;
    for (size_t i = 0; i < num_commands; ++i) literals_count += cmds[i].insert_len_;
    uint16_t* literals = (uint16_t*)malloc((literals_count ? literals_count : 1) * sizeof(uint16_t));
    /* CopyLiteralsToByteArray :97-129 */
    size_t n = 0, from_pos = pos & mask;
    for (size_t i = 0; i < num_commands; ++i) {
      for (size_t j = 0; j < cmds[i].insert_len_; ++j) literals[n++] = data[(from_pos + j) & mask];
      from_pos = (from_pos + cmds[i].insert_len_ + orc_command_copy_len(&cmds[i])) & mask;
    }
    split_byte_vector(literals, literals_count, kSymbolsPerLiteralHistogram, kMaxLiteralHistograms, kLiteralStrideLength,
                      kLiteralBlockSwitchCost, quality, 256, literal_split);
    free(literals);
  }
  {
    uint16_t* codes = (uint16_t*)malloc((num_commands ? num_commands : 1) * sizeof(uint16_t));
    for (size_t i = 0; i < num_commands; ++i) codes[i] = cmds[i].cmd_prefix_;
    split_byte_vector(codes, num_commands, kSymbolsPerCommandHistogram, kMaxCommandHistograms, kCommandStrideLength,
                      kCommandBlockSwitchCost, quality, 704, insert_and_copy_split);
    free(codes);
  }
  {
    uint16_t* prefixes = (uint16_t*)malloc((num_commands ? num_commands : 1) * sizeof(uint16_t));
    size_t j = 0;
    for (size_t i = 0; i < num_commands; ++i) {
      const Command* cmd = &cmds[i];
      if (orc_command_copy_len(cmd) != 0 && cmd->cmd_prefix_ >= 128) prefixes[j++] = cmd->dist_prefix_ & 0x03ff;
    }
    split_byte_vector(prefixes, j, kSymbolsPerDistanceHistogram, kMaxCommandHistograms, kCommandStrideLength,
                      kDistanceBlockSwitchCost, quality, ORC_NUM_DISTANCE_HISTO_SYMBOLS, dist_split);
    free(prefixes);
  }
}

/* ------------------------------------------------------------------ metablock.rs */
typedef struct {
  const BlockSplit* split;
  size_t idx, type, length;
} SplitIterator;
static void split_iter_init(SplitIterator* it, const BlockSplit* split) { /* block_split.rs / histogram.rs:423-463 */
  it->split = split;
  it->idx = 0;
  it->type = 0;
  it->length = split->num_blocks != 0 ? split->lengths[0] : 0;
}
static void split_iter_next(SplitIterator* it) {
  if (it->length == 0) {
    it->idx++;
    it->type = it->split->types[it->idx];
    it->length = it->split->lengths[it->idx];
  }
  it->length--;
}

/* metablock.rs:88-131 */
static int compute_distance_cost(const Command* cmds, size_t num_commands, const DistanceParams* orig_params,
                                 const DistanceParams* new_params, double* cost) {
  int equal_params = 0;
  uint16_t dist_prefix = 0;
  uint32_t dist_extra = 0;
  double extra_bits = 0.0;
  Histo* histo = (Histo*)malloc(sizeof(Histo));
  histo_clear(histo, ORC_NUM_DISTANCE_HISTO_SYMBOLS);
  if (orig_params->distance_postfix_bits == new_params->distance_postfix_bits &&
      orig_params->num_direct_distance_codes == new_params->num_direct_distance_codes)
    equal_params = 1;
  for (size_t i = 0; i < num_commands; ++i) {
    const Command* cmd = &cmds[i];
    if (orc_command_copy_len(cmd) != 0 && cmd->cmd_prefix_ >= 128) {
      if (equal_params) {
        dist_prefix = cmd->dist_prefix_;
      } else {
        uint32_t distance = orc_command_restore_distance_code(cmd, orig_params);
        if (distance > (uint32_t)new_params->max_distance) {
          free(histo);
          return 0;
        }
        orc_prefix_encode_copy_distance(distance, new_params->num_direct_distance_codes,
                                        new_params->distance_postfix_bits, &dist_prefix, &dist_extra);
      }
      histo_add(histo, dist_prefix & 0x03ff);
      extra_bits += (double)(dist_prefix >> 10);
    }
  }
  *cost = (double)population_cost(histo, ORC_NUM_DISTANCE_HISTO_SYMBOLS) + extra_bits;
  free(histo);
  return 1;
}

/* metablock.rs:133-307.  `params` is the per-meta-block copy whose distance parameters the search may change. */
void orc_build_meta_block(const uint8_t* ringbuffer, size_t pos, size_t mask, EncoderParams* params, uint8_t prev_byte,
                          uint8_t prev_byte2, Command* cmds, size_t num_commands, int literal_context_mode,
                          MetaBlockSplit* mb) {
  const size_t kMaxNumberOfHistograms = 256;
  size_t literal_context_multiplier = 1;
  uint32_t ndirect_msb = 0;
  int check_orig = 1;
  memset(mb, 0, sizeof(*mb));
  { /* avoid_distance_prefix_search is false by default (encode.rs:329) and has no C-ABI parameter here */
    double best_dist_cost = 1e99;
    EncoderParams orig_params = *params;
    EncoderParams new_params = *params;
    for (uint32_t npostfix = 0; npostfix <= 3; ++npostfix) {
      while (ndirect_msb < 16) {
        uint32_t ndirect = ndirect_msb << npostfix;
        double dist_cost = 0.0;
        orc_init_distance_params(&new_params, npostfix, ndirect);
        if (npostfix == orig_params.dist.distance_postfix_bits && ndirect == orig_params.dist.num_direct_distance_codes)
          check_orig = 0;
        int skip = !compute_distance_cost(cmds, num_commands, &orig_params.dist, &new_params.dist, &dist_cost);
        if (skip || dist_cost > best_dist_cost) break;
        best_dist_cost = dist_cost;
        params->dist = new_params.dist;
        ndirect_msb += 1;
      }
      ndirect_msb = ndirect_msb > 0 ? ndirect_msb - 1 : 0;
      ndirect_msb /= 2;
    }
    if (check_orig) {
      double dist_cost = 0.0;
      compute_distance_cost(cmds, num_commands, &orig_params.dist, &orig_params.dist, &dist_cost);
      if (dist_cost < best_dist_cost) params->dist = orig_params.dist;
    }
    /* RecomputeDistancePrefixes :62-86 */
    if (!(orig_params.dist.distance_postfix_bits == params->dist.distance_postfix_bits &&
          orig_params.dist.num_direct_distance_codes == params->dist.num_direct_distance_codes)) {
      for (size_t i = 0; i < num_commands; ++i) {
        Command* cmd = &cmds[i];
        if (orc_command_copy_len(cmd) != 0 && cmd->cmd_prefix_ >= 128) {
          uint32_t ret = orc_command_restore_distance_code(cmd, &orig_params.dist);
          orc_prefix_encode_copy_distance(ret, params->dist.num_direct_distance_codes, params->dist.distance_postfix_bits,
                                          &cmd->dist_prefix_, &cmd->dist_extra_);
        }
      }
    }
  }
  split_block(cmds, num_commands, ringbuffer, pos, mask, params->quality, &mb->literal_split, &mb->command_split,
              &mb->distance_split);
  int have_modes = 0;
  if (params->disable_literal_context_modeling == 0) {
    literal_context_multiplier = 1 << 6;
    have_modes = 1; /* literal_context_modes = [literal_context_mode; num_types] */
  }
  size_t literal_histograms_size = mb->literal_split.num_types * literal_context_multiplier;
  Histo* literal_histograms = (Histo*)malloc((literal_histograms_size ? literal_histograms_size : 1) * sizeof(Histo));
  for (size_t i = 0; i < literal_histograms_size; ++i) histo_clear(&literal_histograms[i], 256);
  size_t distance_histograms_size = mb->distance_split.num_types << 2;
  Histo* distance_histograms = (Histo*)malloc((distance_histograms_size ? distance_histograms_size : 1) * sizeof(Histo));
  for (size_t i = 0; i < distance_histograms_size; ++i) histo_clear(&distance_histograms[i], ORC_NUM_DISTANCE_HISTO_SYMBOLS);
  mb->command_histograms_size = mb->command_split.num_types;
  Histo* command_histograms = (Histo*)malloc(mb->command_histograms_size * sizeof(Histo));
  for (size_t i = 0; i < mb->command_histograms_size; ++i) histo_clear(&command_histograms[i], 704);
  { /* BrotliBuildHistogramsWithContext, histogram.rs:465-534 */
    size_t p = pos;
    uint8_t pb = prev_byte, pb2 = prev_byte2;
    SplitIterator literal_it, insert_and_copy_it, dist_it;
    split_iter_init(&literal_it, &mb->literal_split);
    split_iter_init(&insert_and_copy_it, &mb->command_split);
    split_iter_init(&dist_it, &mb->distance_split);
    for (size_t i = 0; i < num_commands; ++i) {
      const Command* cmd = &cmds[i];
      split_iter_next(&insert_and_copy_it);
      histo_add(&command_histograms[insert_and_copy_it.type], cmd->cmd_prefix_);
      for (size_t j = cmd->insert_len_; j != 0; --j) {
        split_iter_next(&literal_it);
        size_t context = have_modes ? (literal_it.type << 6) + orc_context(pb, pb2, literal_context_mode) : literal_it.type;
        histo_add(&literal_histograms[context], ringbuffer[p & mask]);
        pb2 = pb;
        pb = ringbuffer[p & mask];
        ++p;
      }
      p += orc_command_copy_len(cmd);
      if (orc_command_copy_len(cmd) != 0) {
        pb2 = ringbuffer[(p - 2) & mask];
        pb = ringbuffer[(p - 1) & mask];
        if (cmd->cmd_prefix_ >= 128) {
          split_iter_next(&dist_it);
          size_t context = (dist_it.type << 2) + orc_command_distance_context(cmd);
          histo_add(&distance_histograms[context], cmd->dist_prefix_ & 0x3ff);
        }
      }
    }
  }
  mb->command_histograms = (uint32_t*)calloc(mb->command_histograms_size * 704, sizeof(uint32_t));
  for (size_t i = 0; i < mb->command_histograms_size; ++i)
    memcpy(mb->command_histograms + i * 704, command_histograms[i].data_, 704 * sizeof(uint32_t));
  free(command_histograms);

  mb->literal_context_map_size = mb->literal_split.num_types << 6;
  mb->literal_context_map = (uint32_t*)calloc(mb->literal_context_map_size, sizeof(uint32_t));
  {
    Histo* out = (Histo*)malloc((literal_histograms_size ? literal_histograms_size : 1) * sizeof(Histo));
    cluster_histograms(literal_histograms, literal_histograms_size, kMaxNumberOfHistograms, 256, out,
                       &mb->literal_histograms_size, mb->literal_context_map);
    mb->literal_histograms = (uint32_t*)calloc(mb->literal_histograms_size * 256, sizeof(uint32_t));
    for (size_t i = 0; i < mb->literal_histograms_size; ++i)
      memcpy(mb->literal_histograms + i * 256, out[i].data_, 256 * sizeof(uint32_t));
    free(out);
  }
  free(literal_histograms);
  if (params->disable_literal_context_modeling != 0) {
    for (size_t i = mb->literal_split.num_types; i != 0;) {
      --i;
      for (size_t j = 0; j < 64; ++j) {
        uint32_t val = mb->literal_context_map[i];
        mb->literal_context_map[(i << 6) + j] = val;
      }
    }
  }
  mb->distance_context_map_size = mb->distance_split.num_types << 2;
  mb->distance_context_map = (uint32_t*)calloc(mb->distance_context_map_size, sizeof(uint32_t));
  {
    Histo* out = (Histo*)malloc((distance_histograms_size ? distance_histograms_size : 1) * sizeof(Histo));
    cluster_histograms(distance_histograms, mb->distance_context_map_size, kMaxNumberOfHistograms,
                       ORC_NUM_DISTANCE_HISTO_SYMBOLS, out, &mb->distance_histograms_size, mb->distance_context_map);
    mb->distance_histograms =
        (uint32_t*)calloc(mb->distance_histograms_size * ORC_NUM_DISTANCE_HISTO_SYMBOLS, sizeof(uint32_t));
    for (size_t i = 0; i < mb->distance_histograms_size; ++i)
      memcpy(mb->distance_histograms + i * ORC_NUM_DISTANCE_HISTO_SYMBOLS, out[i].data_,
             ORC_NUM_DISTANCE_HISTO_SYMBOLS * sizeof(uint32_t));
    free(out);
  }
  free(distance_histograms);
}
