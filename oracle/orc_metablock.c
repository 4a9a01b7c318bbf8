/* oracle/orc_metablock.c -- CPU restatement of the meta-block half of rust-brotli's encoder hot path.
 * TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows:
 *   src/enc/bit_cost.rs:13-42, src/enc/util.rs:17-25          (f32 entropy, FastLog2)
 *   src/enc/metablock.rs:309-1108                             (greedy block splitters, optimize histograms)
 *   src/enc/histogram.rs:357-463                              (histogram ops, Context)
 *   src/enc/entropy_encode.rs                                 (Huffman tree, RLE, canonical codes)
 *   src/enc/brotli_bit_stream.rs:742-757,764-911,1272-2261,2743-2896 (bit writer and meta-block storage)
 * All floating point is f32 evaluated left to right; compile with -ffp-contract=off.
 */
#include <math.h>
#include <assert.h>

#include "orc_internal.h"

/* ------------------------------------------------------------------ f32 entropy */
/* util.rs:17-25: table below 256, otherwise libm log2f (std build of the reference). */
float orc_fast_log2(uint64_t v) {
  if (v < 256) return orc_logs_8()[v];
  return log2f((float)v);
}

/* TEST SWITCH (tests/test_oracle_vs_libbrotlienc.py only; 0 = the rust-brotli behaviour this oracle restates).
   rust-brotli indexes its 65 536-entry log table with `p as u16` (bit_cost.rs:22,27): a histogram count >= 65 536 is
   TRUNCATED before the lookup, and everything is summed in f32.  Google's C encoder 1.0.9 (c/enc/bit_cost.h) sums
   p * log2(p) in double with the true logarithm.  With this switch on the oracle follows the C arithmetic, so that the
   two can be compared on inputs whose meta-blocks hold symbols more frequent than 65 535. */
int orc_test_c109_entropy = 0;

/* bit_cost.rs:13-33 */
float orc_shannon_entropy(const uint32_t* population, size_t size, size_t* total) {
  const float* l16 = orc_logs_16();
  size_t sum = 0;
  float retval = 0.0f;
  if (orc_test_c109_entropy) {
    double r = 0.0;
    for (size_t i = 0; i < size; ++i) {
      size_t p = population[i];
      sum += p;
      r -= (double)p * (p < 256 ? (double)orc_logs_8()[p] : log2((double)p));
    }
    if (sum != 0) r += (double)sum * (sum < 256 ? (double)orc_logs_8()[sum] : log2((double)sum));
    *total = sum;
    return (float)r;
  }
  for (size_t i = 0; i < size; ++i) {
    size_t p = population[i];
    sum += p;
    retval -= (float)p * l16[(uint16_t)p];
  }
  if (sum != 0) retval += (float)sum * orc_fast_log2(sum);
  *total = sum;
  return retval;
}

/* bit_cost.rs:35-42 */
float orc_bits_entropy_impl(const uint32_t* population, size_t size) {
  size_t sum;
  float retval = orc_shannon_entropy(population, size, &sum);
  if (retval < (float)sum) retval = (float)sum;
  return retval;
}
float orc_bits_entropy(const uint32_t* population, size_t size) { return orc_bits_entropy_impl(population, size); }

/* glibc >= 2.27 log2f (sysdeps/ieee754/flt-32/e_log2f.c), restated so that the device version can
   be checked against libm bit for bit.  Table = __log2f_data (16 entries) + degree-4 polynomial. */
static const double kLog2fTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double kLog2fPoly[4] = {-0x1.712b6f70a7e4dp-2, 0x1.ecabf496832ep-2, -0x1.715479ffae3dep-1,
                                     0x1.715475f35c8b8p0};
float orc_log2f_restated(float x) {
  uint32_t ix;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  /* only positive normal inputs are needed here (x >= 256) */
  uint32_t tmp = ix - 0x3f330000u;
  int i = (int)((tmp >> (23 - 4)) % 16);
  uint32_t top = tmp & 0xff800000u;
  uint32_t iz = ix - top;
  int k = (int32_t)tmp >> 23;
  double invc = kLog2fTab[i][0], logc = kLog2fTab[i][1];
  float zf;
  memcpy(&zf, &iz, 4);
  double z = (double)zf;
  double r = z * invc - 1;
  double y0 = logc + (double)k;
  double r2 = r * r;
  double y = kLog2fPoly[1] * r + kLog2fPoly[2];
  y = kLog2fPoly[0] * r2 + y;
  double p = kLog2fPoly[3] * r + y0;
  y = y * r2 + p;
  return (float)y;
}

/* exhaustive check of the restatement against libm over every integer FastLog2 can be given
   (256 .. 2^24): returns the number of mismatches */
uint32_t orc_log2f_check_all(void) {
  uint32_t bad = 0;
  for (uint32_t v = 256; v <= (1u << 24); ++v)
    if (orc_log2f_restated((float)v) != log2f((float)v)) bad++;
  return bad;
}

/* histogram.rs:448-463 */
uint8_t orc_context(uint8_t p1, uint8_t p2, int mode) {
  switch (mode) {
    case ORC_CONTEXT_SIGNED:
      return (uint8_t)((orc_signed3_context_lookup()[p1] << 3) + orc_signed3_context_lookup()[p2]);
    case ORC_CONTEXT_UTF8:
      return (uint8_t)(orc_utf8_context_lookup()[p1] | orc_utf8_context_lookup()[256 + p2]);
    case ORC_CONTEXT_MSB6:
      return (uint8_t)(p1 >> 2);
    default:
      return (uint8_t)(p1 & 0x3f);
  }
}

/* ------------------------------------------------------------------ greedy block splitters */
typedef struct {
  size_t alphabet_size_;   /* entropy alphabet */
  size_t histo_len;        /* physical histogram width */
  size_t num_contexts_;    /* 1 for the plain splitter */
  size_t max_block_types_;
  size_t min_block_size_;
  float split_threshold_;
  size_t num_blocks_;
  size_t target_block_size_;
  size_t block_size_;
  size_t curr_histogram_ix_;
  size_t last_histogram_ix_[2];
  float last_entropy_[2 * 13];
  size_t merge_last_count_;
  BlockSplit* split;
  uint32_t* histograms; /* [histograms_size][histo_len] */
  size_t histograms_size;
  int is_context; /* ContextBlockSplitter vs BlockSplitter type limits */
} Splitter;

static void histo_clear(Splitter* s, size_t ix) { memset(s->histograms + ix * s->histo_len, 0, s->histo_len * 4); }

/* metablock.rs:385-470 (InitBlockSplitter) / :471-550 (InitContextBlockSplitter) */
static void splitter_init(Splitter* s, int is_context, size_t alphabet_size, size_t histo_len, size_t num_contexts,
                          size_t min_block_size, float split_threshold, size_t num_symbols, BlockSplit* split) {
  size_t max_num_blocks = num_symbols / min_block_size + 1;
  memset(s, 0, sizeof(*s));
  s->is_context = is_context;
  s->alphabet_size_ = alphabet_size;
  s->histo_len = histo_len;
  s->num_contexts_ = num_contexts;
  s->max_block_types_ = is_context ? 256 / num_contexts : 256;
  s->min_block_size_ = min_block_size;
  s->split_threshold_ = split_threshold;
  s->target_block_size_ = min_block_size;
  size_t max_num_types = ORC_MIN(max_num_blocks, s->max_block_types_ + 1);
  s->split = split;
  split->types = (uint8_t*)calloc(max_num_blocks, 1);
  split->lengths = (uint32_t*)calloc(max_num_blocks, 4);
  split->num_blocks = max_num_blocks;
  split->num_types = 0;
  s->histograms_size = max_num_types * num_contexts;
  s->histograms = (uint32_t*)calloc(s->histograms_size * histo_len, 4);
}

/* metablock.rs:551-670 (BlockSplitterFinishBlock) and :673-792 (ContextBlockSplitterFinishBlock);
   the plain splitter is the num_contexts == 1 instance of the same decision procedure, except that
   its type limit is the constant 256 (:588) instead of max_block_types_ (:733). */
static void splitter_finish_block(Splitter* s, int is_final) {
  BlockSplit* split = s->split;
  const size_t nc = s->num_contexts_;
  const size_t hl = s->histo_len;
  if (s->block_size_ < s->min_block_size_) s->block_size_ = s->min_block_size_;
  if (s->num_blocks_ == 0) {
    split->lengths[0] = (uint32_t)s->block_size_;
    split->types[0] = 0;
    for (size_t i = 0; i < nc; ++i) {
      s->last_entropy_[i] = orc_bits_entropy_impl(s->histograms + i * hl, s->alphabet_size_);
      s->last_entropy_[nc + i] = s->last_entropy_[i];
    }
    s->num_blocks_++;
    split->num_types++;
    s->curr_histogram_ix_ += nc;
    if (s->curr_histogram_ix_ < s->histograms_size)
      for (size_t i = 0; i < nc; ++i) histo_clear(s, s->curr_histogram_ix_ + i);
    s->block_size_ = 0;
  } else if (s->block_size_ > 0) {
    float entropy[13];
    float combined_entropy[2 * 13];
    float diff[2] = {0.0f, 0.0f};
    uint32_t* combined_histo = (uint32_t*)malloc(2 * nc * hl * 4);
    for (size_t i = 0; i < nc; ++i) {
      size_t curr_histo_ix = s->curr_histogram_ix_ + i;
      entropy[i] = orc_bits_entropy_impl(s->histograms + curr_histo_ix * hl, s->alphabet_size_);
      for (size_t j = 0; j < 2; ++j) {
        size_t jx = j * nc + i;
        size_t last_histogram_ix = s->last_histogram_ix_[j] + i;
        uint32_t* ch = combined_histo + jx * hl;
        const uint32_t* a = s->histograms + curr_histo_ix * hl;
        const uint32_t* b = s->histograms + last_histogram_ix * hl;
        for (size_t k = 0; k < hl; ++k) ch[k] = a[k] + b[k];
        combined_entropy[jx] = orc_bits_entropy_impl(ch, s->alphabet_size_);
        if (s->is_context) {
          diff[j] += combined_entropy[jx] - entropy[i] - s->last_entropy_[jx];
        } else {
          diff[j] = combined_entropy[jx] - entropy[i] - s->last_entropy_[jx];
        }
      }
    }
    if (split->num_types < s->max_block_types_ && diff[0] > s->split_threshold_ && diff[1] > s->split_threshold_) {
      split->lengths[s->num_blocks_] = (uint32_t)s->block_size_;
      split->types[s->num_blocks_] = (uint8_t)split->num_types;
      s->last_histogram_ix_[1] = s->last_histogram_ix_[0];
      s->last_histogram_ix_[0] = s->is_context ? split->num_types * nc : (size_t)(uint8_t)split->num_types;
      for (size_t i = 0; i < nc; ++i) {
        s->last_entropy_[nc + i] = s->last_entropy_[i];
        s->last_entropy_[i] = entropy[i];
      }
      s->num_blocks_++;
      split->num_types++;
      s->curr_histogram_ix_ += nc;
      if (s->curr_histogram_ix_ < s->histograms_size)
        for (size_t i = 0; i < nc; ++i) histo_clear(s, s->curr_histogram_ix_ + i);
      s->block_size_ = 0;
      s->merge_last_count_ = 0;
      s->target_block_size_ = s->min_block_size_;
    } else if (diff[1] < diff[0] - 20.0f) {
      split->lengths[s->num_blocks_] = (uint32_t)s->block_size_;
      split->types[s->num_blocks_] = split->types[s->num_blocks_ - 2];
      {
        size_t t = s->last_histogram_ix_[0];
        s->last_histogram_ix_[0] = s->last_histogram_ix_[1];
        s->last_histogram_ix_[1] = t;
      }
      for (size_t i = 0; i < nc; ++i) {
        memcpy(s->histograms + (s->last_histogram_ix_[0] + i) * hl, combined_histo + (nc + i) * hl, hl * 4);
        s->last_entropy_[nc + i] = s->last_entropy_[i];
        s->last_entropy_[i] = combined_entropy[nc + i];
        histo_clear(s, s->curr_histogram_ix_ + i);
      }
      s->num_blocks_++;
      s->block_size_ = 0;
      s->merge_last_count_ = 0;
      s->target_block_size_ = s->min_block_size_;
    } else {
      split->lengths[s->num_blocks_ - 1] += (uint32_t)s->block_size_;
      for (size_t i = 0; i < nc; ++i) {
        memcpy(s->histograms + (s->last_histogram_ix_[0] + i) * hl, combined_histo + i * hl, hl * 4);
        s->last_entropy_[i] = combined_entropy[i];
        if (split->num_types == 1) s->last_entropy_[nc + i] = s->last_entropy_[i];
        histo_clear(s, s->curr_histogram_ix_ + i);
      }
      s->block_size_ = 0;
      if (++s->merge_last_count_ > 1) s->target_block_size_ += s->min_block_size_;
    }
    free(combined_histo);
  }
  if (is_final) {
    s->histograms_size = split->num_types * nc;
    split->num_blocks = s->num_blocks_;
  }
}

static inline void splitter_add_symbol(Splitter* s, size_t symbol, size_t context) {
  s->histograms[(s->curr_histogram_ix_ + context) * s->histo_len + symbol]++;
  s->block_size_++;
  if (s->block_size_ == s->target_block_size_) splitter_finish_block(s, 0);
}

void orc_metablock_destroy(MetaBlockSplit* mb) {
  free(mb->literal_split.types);
  free(mb->literal_split.lengths);
  free(mb->command_split.types);
  free(mb->command_split.lengths);
  free(mb->distance_split.types);
  free(mb->distance_split.lengths);
  free(mb->literal_context_map);
  free(mb->distance_context_map);
  free(mb->literal_histograms);
  free(mb->command_histograms);
  free(mb->distance_histograms);
  memset(mb, 0, sizeof(*mb));
}

/* metablock.rs:858-1075 */
void orc_build_meta_block_greedy(const uint8_t* ringbuffer, size_t pos, size_t mask, uint8_t prev_byte,
                                 uint8_t prev_byte2, int literal_context_mode, size_t num_contexts,
                                 const uint32_t* static_context_map, const Command* commands,
                                 size_t n_commands, MetaBlockSplit* mb) {
  Splitter lit, cmd, dst;
  size_t num_literals = 0;
  memset(mb, 0, sizeof(*mb));
  for (size_t i = 0; i < n_commands; ++i) num_literals += commands[i].insert_len_;
  if (num_contexts == 1) {
    splitter_init(&lit, 0, 256, 256, 1, 512, 400.0f, num_literals, &mb->literal_split);
  } else {
    splitter_init(&lit, 1, 256, 256, num_contexts, 512, 400.0f, num_literals, &mb->literal_split);
  }
  splitter_init(&cmd, 0, 704, 704, 1, 1024, 500.0f, n_commands, &mb->command_split);
  splitter_init(&dst, 0, 64, ORC_NUM_DISTANCE_HISTO_SYMBOLS, 1, 512, 100.0f, n_commands, &mb->distance_split);
  for (size_t i = 0; i < n_commands; ++i) {
    const Command c = commands[i];
    splitter_add_symbol(&cmd, c.cmd_prefix_, 0);
    for (size_t j = c.insert_len_; j != 0; --j) {
      uint8_t literal = ringbuffer[pos & mask];
      if (num_contexts == 1) {
        splitter_add_symbol(&lit, literal, 0);
      } else {
        size_t context = orc_context(prev_byte, prev_byte2, literal_context_mode);
        splitter_add_symbol(&lit, literal, static_context_map[context]);
      }
      prev_byte2 = prev_byte;
      prev_byte = literal;
      pos++;
    }
    pos += orc_command_copy_len(&c);
    if (orc_command_copy_len(&c) != 0) {
      prev_byte2 = ringbuffer[(pos - 2) & mask];
      prev_byte = ringbuffer[(pos - 1) & mask];
      if (c.cmd_prefix_ >= 128) splitter_add_symbol(&dst, c.dist_prefix_ & 0x3ff, 0);
    }
  }
  splitter_finish_block(&lit, 1);
  splitter_finish_block(&cmd, 1);
  splitter_finish_block(&dst, 1);
  mb->literal_histograms = lit.histograms;
  mb->literal_histograms_size = lit.histograms_size;
  mb->command_histograms = cmd.histograms;
  mb->command_histograms_size = cmd.histograms_size;
  mb->distance_histograms = dst.histograms;
  mb->distance_histograms_size = dst.histograms_size;
  if (num_contexts > 1) {
    /* MapStaticContexts, metablock.rs:832-857 */
    mb->literal_context_map_size = mb->literal_split.num_types << 6;
    mb->literal_context_map = (uint32_t*)calloc(mb->literal_context_map_size, 4);
    for (size_t i = 0; i < mb->literal_split.num_types; ++i) {
      uint32_t offset = (uint32_t)(i * num_contexts);
      for (size_t j = 0; j < 64; ++j) mb->literal_context_map[(i << 6) + j] = offset + static_context_map[j];
    }
  }
}

/* ------------------------------------------------------------------ entropy_encode.rs */
typedef struct {
  uint32_t total_count_;
  int16_t index_left_;
  int16_t index_right_or_value_;
} HuffmanTree;

/* entropy_encode.rs:27-56 */
static int set_depth(int p0, HuffmanTree* pool, uint8_t* depth, int max_depth) {
  int stack[16];
  int level = 0;
  int p = p0;
  stack[0] = -1;
  for (;;) {
    if (pool[p].index_left_ >= 0) {
      level++;
      if (level > max_depth) return 0;
      stack[level] = pool[p].index_right_or_value_;
      p = pool[p].index_left_;
      continue;
    } else {
      depth[pool[p].index_right_or_value_] = (uint8_t)level;
    }
    while (level >= 0 && stack[level] == -1) level--;
    if (level < 0) return 1;
    p = stack[level];
    stack[level] = -1;
  }
}

/* entropy_encode.rs:61-69 */
static inline int sort_cmp(const HuffmanTree* v0, const HuffmanTree* v1) {
  if (v0->total_count_ != v1->total_count_) return v0->total_count_ < v1->total_count_;
  return v0->index_right_or_value_ > v1->index_right_or_value_;
}

/* entropy_encode.rs:71-116 */
static int sort_simple = 0; /* SimpleSortHuffmanTree (brotli_bit_stream.rs:917-923): by count only */
static inline int sort_cmp_sel(const HuffmanTree* v0, const HuffmanTree* v1) {
  return sort_simple ? v0->total_count_ < v1->total_count_ : sort_cmp(v0, v1);
}
static void sort_huffman_tree_items(HuffmanTree* items, size_t n) {
  static const size_t gaps[6] = {132, 57, 23, 10, 4, 1};
  if (n < 13) {
    for (size_t i = 1; i < n; ++i) {
      HuffmanTree tmp = items[i];
      size_t k = i;
      size_t j = i - 1;
      while (sort_cmp_sel(&tmp, &items[j])) {
        items[k] = items[j];
        k = j;
        if (j-- == 0) break;
      }
      items[k] = tmp;
    }
  } else {
    for (int g = n < 57 ? 2 : 0; g < 6; ++g) {
      size_t gap = gaps[g];
      for (size_t i = gap; i < n; ++i) {
        size_t j = i;
        HuffmanTree tmp = items[i];
        for (; j >= gap && sort_cmp_sel(&tmp, &items[j - gap]); j -= gap) items[j] = items[j - gap];
        items[j] = tmp;
      }
    }
  }
}

/* entropy_encode.rs:133-210 */
static void create_huffman_tree(const uint32_t* data, size_t length, int tree_limit, HuffmanTree* tree,
                                uint8_t* depth) {
  const HuffmanTree sentinel = {0xffffffffu, -1, -1};
  uint32_t count_limit;
  for (count_limit = 1;; count_limit *= 2) {
    size_t n = 0;
    for (size_t i = length; i != 0;) {
      --i;
      if (data[i] != 0) {
        uint32_t count = ORC_MAX(data[i], count_limit);
        tree[n].total_count_ = count;
        tree[n].index_left_ = -1;
        tree[n].index_right_or_value_ = (int16_t)i;
        n++;
      }
    }
    if (n == 1) {
      depth[tree[0].index_right_or_value_] = 1;
      break;
    }
    sort_huffman_tree_items(tree, n);
    tree[n] = sentinel;
    tree[n + 1] = sentinel;
    size_t i = 0, j = n + 1;
    for (size_t k = n - 1; k != 0; --k) {
      size_t left, right;
      if (tree[i].total_count_ <= tree[j].total_count_) {
        left = i++;
      } else {
        left = j++;
      }
      if (tree[i].total_count_ <= tree[j].total_count_) {
        right = i++;
      } else {
        right = j++;
      }
      size_t j_end = 2 * n - k;
      tree[j_end].total_count_ = tree[left].total_count_ + tree[right].total_count_;
      tree[j_end].index_left_ = (int16_t)left;
      tree[j_end].index_right_or_value_ = (int16_t)right;
      tree[j_end + 1] = sentinel;
    }
    if (set_depth((int)(2 * n - 1), tree, depth, tree_limit)) break;
  }
}

/* entropy_encode.rs:211-345 */
static void optimize_huffman_counts_for_rle(size_t length, uint32_t* counts, uint8_t* good_for_rle /*[704]*/) {
  size_t nonzero_count = 0, stride, limit, sum;
  const size_t streak_limit = 1240;
  for (size_t i = 0; i < length; ++i)
    if (counts[i] != 0) nonzero_count++;
  if (nonzero_count < 16) return;
  while (length != 0 && counts[length - 1] == 0) length--;
  if (length == 0) return;
  {
    size_t nonzeros = 0;
    uint32_t smallest_nonzero = 1u << 30;
    for (size_t i = 0; i < length; ++i) {
      if (counts[i] != 0) {
        nonzeros++;
        if (smallest_nonzero > counts[i]) smallest_nonzero = counts[i];
      }
    }
    if (nonzeros < 5) return;
    if (smallest_nonzero < 4) {
      size_t zeros = length - nonzeros;
      if (zeros < 6) {
        for (size_t i = 1; i < length - 1; ++i)
          if (counts[i - 1] != 0 && counts[i] == 0 && counts[i + 1] != 0) counts[i] = 1;
      }
    }
    if (nonzeros < 28) return;
  }
  memset(good_for_rle, 0, 704);
  {
    uint32_t symbol = counts[0];
    size_t step = 0;
    for (size_t i = 0; i <= length; ++i) {
      if (i == length || counts[i] != symbol) {
        if ((symbol == 0 && step >= 5) || (symbol != 0 && step >= 7)) {
          for (size_t k = 0; k < step; ++k) good_for_rle[i - k - 1] = 1;
        }
        step = 1;
        if (i != length) symbol = counts[i];
      } else {
        step++;
      }
    }
  }
  stride = 0;
  limit = (size_t)((uint32_t)(256u * (counts[0] + counts[1] + counts[2])) / 3u + 420u);
  sum = 0;
  for (size_t i = 0; i <= length; ++i) {
    if (i == length || good_for_rle[i] != 0 || (i != 0 && good_for_rle[i - 1] != 0) ||
        (size_t)(uint32_t)(256u * counts[i]) - limit + streak_limit >= 2 * streak_limit) {
      if (stride >= 4 || (stride >= 3 && sum == 0)) {
        size_t count = (sum + stride / 2) / stride;
        if (count == 0) count = 1;
        if (sum == 0) count = 0;
        for (size_t k = 0; k < stride; ++k) counts[i - k - 1] = (uint32_t)count;
      }
      stride = 0;
      sum = 0;
      if (i < length - 2) {
        limit = (size_t)((uint32_t)(256u * (counts[i] + counts[i + 1] + counts[i + 2])) / 3u + 420u);
      } else if (i < length) {
        limit = (size_t)(uint32_t)(256u * counts[i]);
      } else {
        limit = 0;
      }
    }
    stride++;
    if (i != length) {
      sum += counts[i];
      if (stride >= 4) limit = (256 * sum + stride / 2) / stride;
      if (stride == 4) limit += 120;
    }
  }
}

/* metablock.rs:1076-1108 */
void orc_optimize_histograms(size_t num_distance_codes, MetaBlockSplit* mb) {
  uint8_t good_for_rle[704];
  memset(good_for_rle, 0, sizeof(good_for_rle));
  for (size_t i = 0; i < mb->literal_histograms_size; ++i)
    optimize_huffman_counts_for_rle(256, mb->literal_histograms + i * 256, good_for_rle);
  for (size_t i = 0; i < mb->command_histograms_size; ++i)
    optimize_huffman_counts_for_rle(704, mb->command_histograms + i * 704, good_for_rle);
  for (size_t i = 0; i < mb->distance_histograms_size; ++i)
    optimize_huffman_counts_for_rle(num_distance_codes, mb->distance_histograms + i * ORC_NUM_DISTANCE_HISTO_SYMBOLS,
                                    good_for_rle);
}

/* entropy_encode.rs:347-378 */
static void decide_over_rle_use(const uint8_t* depth, size_t length, int* use_rle_for_non_zero,
                                int* use_rle_for_zero) {
  size_t total_reps_zero = 0, total_reps_non_zero = 0, count_reps_zero = 1, count_reps_non_zero = 1;
  for (size_t i = 0; i < length;) {
    uint8_t value = depth[i];
    size_t reps = 1;
    for (size_t k = i + 1; k < length && depth[k] == value; ++k) reps++;
    if (reps >= 3 && value == 0) {
      total_reps_zero += reps;
      count_reps_zero++;
    }
    if (reps >= 4 && value != 0) {
      total_reps_non_zero += reps;
      count_reps_non_zero++;
    }
    i += reps;
  }
  *use_rle_for_non_zero = total_reps_non_zero > count_reps_non_zero * 2;
  *use_rle_for_zero = total_reps_zero > count_reps_zero * 2;
}

static void reverse_u8(uint8_t* v, size_t start, size_t end) {
  end--;
  while (start < end) {
    uint8_t t = v[start];
    v[start] = v[end];
    v[end] = t;
    start++;
    end--;
  }
}

/* entropy_encode.rs:389-429 */
static void write_huffman_tree_repetitions(uint8_t previous_value, uint8_t value, size_t repetitions,
                                           size_t* tree_size, uint8_t* tree, uint8_t* extra_bits_data) {
  if (previous_value != value) {
    tree[*tree_size] = value;
    extra_bits_data[*tree_size] = 0;
    (*tree_size)++;
    repetitions--;
  }
  if (repetitions == 7) {
    tree[*tree_size] = value;
    extra_bits_data[*tree_size] = 0;
    (*tree_size)++;
    repetitions--;
  }
  if (repetitions < 3) {
    for (size_t i = 0; i < repetitions; ++i) {
      tree[*tree_size] = value;
      extra_bits_data[*tree_size] = 0;
      (*tree_size)++;
    }
  } else {
    size_t start = *tree_size;
    repetitions -= 3;
    for (;;) {
      tree[*tree_size] = 16;
      extra_bits_data[*tree_size] = (uint8_t)(repetitions & 0x3);
      (*tree_size)++;
      repetitions >>= 2;
      if (repetitions == 0) break;
      repetitions--;
    }
    reverse_u8(tree, start, *tree_size);
    reverse_u8(extra_bits_data, start, *tree_size);
  }
}

/* entropy_encode.rs:431-468 */
static void write_huffman_tree_repetitions_zeros(size_t repetitions, size_t* tree_size, uint8_t* tree,
                                                 uint8_t* extra_bits_data) {
  if (repetitions == 11) {
    tree[*tree_size] = 0;
    extra_bits_data[*tree_size] = 0;
    (*tree_size)++;
    repetitions--;
  }
  if (repetitions < 3) {
    for (size_t i = 0; i < repetitions; ++i) {
      tree[*tree_size] = 0;
      extra_bits_data[*tree_size] = 0;
      (*tree_size)++;
    }
  } else {
    size_t start = *tree_size;
    repetitions -= 3;
    for (;;) {
      tree[*tree_size] = 17;
      extra_bits_data[*tree_size] = (uint8_t)(repetitions & 0x7);
      (*tree_size)++;
      repetitions >>= 3;
      if (repetitions == 0) break;
      repetitions--;
    }
    reverse_u8(tree, start, *tree_size);
    reverse_u8(extra_bits_data, start, *tree_size);
  }
}

/* entropy_encode.rs:470-525 */
static void write_huffman_tree(const uint8_t* depth, size_t length, size_t* tree_size, uint8_t* tree,
                               uint8_t* extra_bits_data) {
  uint8_t previous_value = 8;
  int use_rle_for_non_zero = 0, use_rle_for_zero = 0;
  size_t new_length = length;
  for (size_t i = 0; i < length; ++i) {
    if (depth[length - i - 1] == 0) {
      new_length--;
    } else {
      break;
    }
  }
  if (length > 50) decide_over_rle_use(depth, new_length, &use_rle_for_non_zero, &use_rle_for_zero);
  for (size_t i = 0; i < new_length;) {
    uint8_t value = depth[i];
    size_t reps = 1;
    if ((value != 0 && use_rle_for_non_zero) || (value == 0 && use_rle_for_zero)) {
      for (size_t k = i + 1; k < new_length && depth[k] == value; ++k) reps++;
    }
    if (value == 0) {
      write_huffman_tree_repetitions_zeros(reps, tree_size, tree, extra_bits_data);
    } else {
      write_huffman_tree_repetitions(previous_value, value, reps, tree_size, tree, extra_bits_data);
      previous_value = value;
    }
    i += reps;
  }
}

/* entropy_encode.rs:527-544 */
static uint16_t reverse_bits(size_t num_bits, uint16_t bits) {
  static const size_t kLut[16] = {0x0, 0x8, 0x4, 0xc, 0x2, 0xa, 0x6, 0xe, 0x1, 0x9, 0x5, 0xd, 0x3, 0xb, 0x7, 0xf};
  size_t retval = kLut[bits & 0xf];
  for (size_t i = 4; i < num_bits; i += 4) {
    retval <<= 4;
    bits = (uint16_t)(bits >> 4);
    retval |= kLut[bits & 0xf];
  }
  retval >>= ((0 - num_bits) & 0x3);
  return (uint16_t)retval;
}

/* entropy_encode.rs:546-575 */
static void convert_bit_depths_to_symbols(const uint8_t* depth, size_t len, uint16_t* bits) {
  uint16_t bl_count[16] = {0};
  uint16_t next_code[16];
  int code = 0;
  for (size_t i = 0; i < len; ++i) bl_count[depth[i]]++;
  bl_count[0] = 0;
  next_code[0] = 0;
  for (size_t i = 1; i < 16; ++i) {
    code = (code + bl_count[i - 1]) << 1;
    next_code[i] = (uint16_t)code;
  }
  for (size_t i = 0; i < len; ++i) {
    if (depth[i] != 0) bits[i] = reverse_bits(depth[i], next_code[depth[i]]++);
  }
}

/* ------------------------------------------------------------------ bit stream */
/* brotli_bit_stream.rs:742-757 */
void orc_write_bits(unsigned n_bits, uint64_t bits, size_t* pos, uint8_t* array) {
  uint8_t* p = &array[*pos >> 3];
  uint64_t v = (uint64_t)*p;
  v |= bits << (*pos & 7);
  memcpy(p, &v, 8); /* little-endian host */
  *pos += n_bits;
}

static void jump_to_byte_boundary(size_t* storage_ix, uint8_t* storage) {
  *storage_ix = (*storage_ix + 7u) & ~(size_t)7u;
  storage[*storage_ix >> 3] = 0;
}

/* brotli_bit_stream.rs:764-803 */
static void store_huffman_tree_of_huffman_tree_to_bit_mask(int num_codes, const uint8_t* code_length_bitdepth,
                                                           size_t* storage_ix, uint8_t* storage) {
  static const uint8_t kStorageOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kSymbols[6] = {0, 7, 3, 2, 1, 15};
  static const uint8_t kBitLengths[6] = {2, 4, 3, 2, 2, 4};
  uint64_t skip_some = 0;
  uint64_t codes_to_store = 18;
  if (num_codes > 1) {
    for (; codes_to_store > 0; --codes_to_store) {
      if (code_length_bitdepth[kStorageOrder[codes_to_store - 1]] != 0) break;
    }
  }
  if (code_length_bitdepth[kStorageOrder[0]] == 0 && code_length_bitdepth[kStorageOrder[1]] == 0) {
    skip_some = 2;
    if (code_length_bitdepth[kStorageOrder[2]] == 0) skip_some = 3;
  }
  orc_write_bits(2, skip_some, storage_ix, storage);
  for (uint64_t i = skip_some; i < codes_to_store; ++i) {
    size_t l = code_length_bitdepth[kStorageOrder[i]];
    orc_write_bits(kBitLengths[l], kSymbols[l], storage_ix, storage);
  }
}

/* brotli_bit_stream.rs:835-911 */
static void store_huffman_tree(const uint8_t* depths, size_t num, HuffmanTree* tree, size_t* storage_ix,
                               uint8_t* storage) {
  uint8_t huffman_tree[704];
  uint8_t huffman_tree_extra_bits[704];
  size_t huffman_tree_size = 0;
  uint8_t code_length_bitdepth[18] = {0};
  uint16_t code_length_bitdepth_symbols[18] = {0};
  uint32_t huffman_tree_histogram[18] = {0};
  int num_codes = 0;
  size_t code = 0;
  write_huffman_tree(depths, num, &huffman_tree_size, huffman_tree, huffman_tree_extra_bits);
  for (size_t i = 0; i < huffman_tree_size; ++i) huffman_tree_histogram[huffman_tree[i]]++;
  for (size_t i = 0; i < 18; ++i) {
    if (huffman_tree_histogram[i] != 0) {
      if (num_codes == 0) {
        code = i;
        num_codes = 1;
      } else if (num_codes == 1) {
        num_codes = 2;
        break;
      }
    }
  }
  create_huffman_tree(huffman_tree_histogram, 18, 5, tree, code_length_bitdepth);
  convert_bit_depths_to_symbols(code_length_bitdepth, 18, code_length_bitdepth_symbols);
  store_huffman_tree_of_huffman_tree_to_bit_mask(num_codes, code_length_bitdepth, storage_ix, storage);
  if (num_codes == 1) code_length_bitdepth[code] = 0;
  for (size_t i = 0; i < huffman_tree_size; ++i) {
    size_t ix = huffman_tree[i];
    orc_write_bits(code_length_bitdepth[ix], code_length_bitdepth_symbols[ix], storage_ix, storage);
    if (ix == 16) {
      orc_write_bits(2, huffman_tree_extra_bits[i], storage_ix, storage);
    } else if (ix == 17) {
      orc_write_bits(3, huffman_tree_extra_bits[i], storage_ix, storage);
    }
  }
}

/* brotli_bit_stream.rs:1401-1443 */
static void store_simple_huffman_tree(const uint8_t* depths, size_t* symbols, size_t num_symbols, size_t max_bits,
                                      size_t* storage_ix, uint8_t* storage) {
  orc_write_bits(2, 1, storage_ix, storage);
  orc_write_bits(2, num_symbols - 1, storage_ix, storage);
  for (size_t i = 0; i < num_symbols; ++i) {
    for (size_t j = i + 1; j < num_symbols; ++j) {
      if (depths[symbols[j]] < depths[symbols[i]]) {
        size_t t = symbols[j];
        symbols[j] = symbols[i];
        symbols[i] = t;
      }
    }
  }
  for (size_t i = 0; i < num_symbols && i < 4; ++i) orc_write_bits((unsigned)max_bits, symbols[i], storage_ix, storage);
  if (num_symbols == 4) orc_write_bits(1, depths[symbols[0]] == 1 ? 1 : 0, storage_ix, storage);
}

/* brotli_bit_stream.rs:1445-1498 */
static void build_and_store_huffman_tree(const uint32_t* histogram, size_t histogram_length, size_t alphabet_size,
                                         HuffmanTree* tree, uint8_t* depth, uint16_t* bits, size_t* storage_ix,
                                         uint8_t* storage) {
  size_t count = 0;
  size_t s4[4] = {0, 0, 0, 0};
  size_t max_bits = 0;
  for (size_t i = 0; i < histogram_length; ++i) {
    if (histogram[i] != 0) {
      if (count < 4) {
        s4[count] = i;
      } else if (count > 4) {
        break;
      }
      count++;
    }
  }
  for (size_t c = alphabet_size - 1; c != 0; c >>= 1) max_bits++;
  if (count <= 1) {
    orc_write_bits(4, 1, storage_ix, storage);
    orc_write_bits((unsigned)max_bits, s4[0], storage_ix, storage);
    depth[s4[0]] = 0;
    bits[s4[0]] = 0;
    return;
  }
  memset(depth, 0, histogram_length);
  create_huffman_tree(histogram, histogram_length, 15, tree, depth);
  convert_bit_depths_to_symbols(depth, histogram_length, bits);
  if (count <= 4) {
    store_simple_huffman_tree(depth, s4, count, max_bits, storage_ix, storage);
  } else {
    store_huffman_tree(depth, histogram_length, tree, storage_ix, storage);
  }
}

/* brotli_bit_stream.rs:635-740 */
static const struct { uint32_t offset, nbits; } kBlockLengthPrefixCode[26] = {
    {1, 2},    {5, 2},    {9, 2},    {13, 2},    {17, 3},    {25, 3},    {33, 3},    {41, 3},   {49, 4},
    {65, 4},   {81, 4},   {97, 4},   {113, 5},   {145, 5},   {177, 5},   {209, 5},   {241, 6},  {305, 6},
    {369, 7},  {497, 8},  {753, 9},  {1265, 10}, {2289, 11}, {4337, 12}, {8433, 13}, {16625, 24}};

/* brotli_bit_stream.rs:1372-1388 */
static uint32_t block_length_prefix_code(uint32_t len) {
  uint32_t code = len >= 177 ? (len >= 753 ? 20 : 14) : (len >= 41 ? 7 : 0);
  while (code < 25 && len >= kBlockLengthPrefixCode[code + 1].offset) code++;
  return code;
}

typedef struct {
  size_t last_type, second_last_type;
} BlockTypeCodeCalculator;

/* brotli_bit_stream.rs:1357-1370 */
static size_t next_block_type_code(BlockTypeCodeCalculator* c, uint8_t type) {
  size_t type_code = (type == c->last_type + 1) ? 1 : (type == c->second_last_type ? 0 : (size_t)type + 2);
  c->second_last_type = c->last_type;
  c->last_type = type;
  return type_code;
}

/* brotli_bit_stream.rs:1390-1399 */
static void store_var_len_uint8(uint64_t n, size_t* storage_ix, uint8_t* storage) {
  if (n == 0) {
    orc_write_bits(1, 0, storage_ix, storage);
  } else {
    unsigned nbits = orc_log2_floor_nonzero(n);
    orc_write_bits(1, 1, storage_ix, storage);
    orc_write_bits(3, nbits, storage_ix, storage);
    orc_write_bits(nbits, n - (1ull << nbits), storage_ix, storage);
  }
}

typedef struct {
  BlockTypeCodeCalculator type_code_calculator;
  uint8_t type_depths[258];
  uint16_t type_bits[258];
  uint8_t length_depths[26];
  uint16_t length_bits[26];
} BlockSplitCode;

/* brotli_bit_stream.rs:1506-1534 */
static void store_block_switch(BlockSplitCode* code, uint32_t block_len, uint8_t block_type, int is_first_block,
                               size_t* storage_ix, uint8_t* storage) {
  size_t typecode = next_block_type_code(&code->type_code_calculator, block_type);
  if (!is_first_block) orc_write_bits(code->type_depths[typecode], code->type_bits[typecode], storage_ix, storage);
  uint32_t lencode = block_length_prefix_code(block_len);
  orc_write_bits(code->length_depths[lencode], code->length_bits[lencode], storage_ix, storage);
  orc_write_bits(kBlockLengthPrefixCode[lencode].nbits, block_len - kBlockLengthPrefixCode[lencode].offset,
                 storage_ix, storage);
}

/* brotli_bit_stream.rs:1536-1591 */
static void build_and_store_block_split_code(const uint8_t* types, const uint32_t* lengths, size_t num_blocks,
                                             size_t num_types, HuffmanTree* tree, BlockSplitCode* code,
                                             size_t* storage_ix, uint8_t* storage) {
  uint32_t type_histo[258] = {0};
  uint32_t length_histo[26] = {0};
  BlockTypeCodeCalculator calc = {1, 0};
  for (size_t i = 0; i < num_blocks; ++i) {
    size_t type_code = next_block_type_code(&calc, types[i]);
    if (i != 0) type_histo[type_code]++;
    length_histo[block_length_prefix_code(lengths[i])]++;
  }
  store_var_len_uint8(num_types - 1, storage_ix, storage);
  if (num_types > 1) {
    build_and_store_huffman_tree(type_histo, num_types + 2, num_types + 2, tree, code->type_depths, code->type_bits,
                                 storage_ix, storage);
    build_and_store_huffman_tree(length_histo, 26, 26, tree, code->length_depths, code->length_bits, storage_ix,
                                 storage);
    store_block_switch(code, lengths[0], types[0], 1, storage_ix, storage);
  }
}

/* brotli_bit_stream.rs:1613-1662 */
static void store_trivial_context_map(size_t num_types, size_t context_bits, HuffmanTree* tree, size_t* storage_ix,
                                      uint8_t* storage) {
  store_var_len_uint8(num_types - 1, storage_ix, storage);
  if (num_types > 1) {
    size_t repeat_code = context_bits - 1;
    size_t repeat_bits = (1u << repeat_code) - 1;
    size_t alphabet_size = num_types + repeat_code;
    uint32_t histogram[272] = {0};
    uint8_t depths[272] = {0};
    uint16_t bits[272] = {0};
    orc_write_bits(1, 1, storage_ix, storage);
    orc_write_bits(4, repeat_code - 1, storage_ix, storage);
    histogram[repeat_code] = (uint32_t)num_types;
    histogram[0] = 1;
    for (size_t i = context_bits; i < alphabet_size; ++i) histogram[i] = 1;
    build_and_store_huffman_tree(histogram, alphabet_size, alphabet_size, tree, depths, bits, storage_ix, storage);
    for (size_t i = 0; i < num_types; ++i) {
      size_t code = i == 0 ? 0 : i + context_bits - 1;
      orc_write_bits(depths[code], bits[code], storage_ix, storage);
      orc_write_bits(depths[repeat_code], bits[repeat_code], storage_ix, storage);
      orc_write_bits((unsigned)repeat_code, repeat_bits, storage_ix, storage);
    }
    orc_write_bits(1, 1, storage_ix, storage);
  }
}

/* brotli_bit_stream.rs:1690-1713 */
static void move_to_front_transform(const uint32_t* v_in, size_t v_size, uint32_t* v_out) {
  uint8_t mtf[256];
  if (v_size == 0) return;
  uint32_t max_value = v_in[0];
  for (size_t i = 1; i < v_size; ++i)
    if (v_in[i] > max_value) max_value = v_in[i];
  for (size_t i = 0; i <= max_value; ++i) mtf[i] = (uint8_t)i;
  size_t mtf_size = max_value + 1;
  for (size_t i = 0; i < v_size; ++i) {
    size_t index = 0;
    while (index < mtf_size && mtf[index] != (uint8_t)v_in[i]) index++;
    v_out[i] = (uint32_t)index;
    uint8_t value = mtf[index];
    for (size_t k = index; k != 0; --k) mtf[k] = mtf[k - 1];
    mtf[0] = value;
  }
}

/* brotli_bit_stream.rs:1715-1781 */
static void run_length_code_zeros(size_t in_size, uint32_t* v, size_t* out_size, uint32_t* max_run_length_prefix) {
  uint32_t max_reps = 0;
  for (size_t i = 0; i < in_size;) {
    uint32_t reps = 0;
    for (; i < in_size && v[i] != 0; ++i) {
    }
    for (; i < in_size && v[i] == 0; ++i) reps++;
    max_reps = ORC_MAX(reps, max_reps);
  }
  uint32_t max_prefix = max_reps > 0 ? orc_log2_floor_nonzero(max_reps) : 0;
  max_prefix = ORC_MIN(max_prefix, *max_run_length_prefix);
  *max_run_length_prefix = max_prefix;
  *out_size = 0;
  for (size_t i = 0; i < in_size;) {
    if (v[i] != 0) {
      v[*out_size] = v[i] + *max_run_length_prefix;
      i++;
      (*out_size)++;
    } else {
      uint32_t reps = 1;
      for (size_t k = i + 1; k < in_size && v[k] == 0; ++k) reps++;
      i += reps;
      while (reps != 0) {
        if (reps < (2u << max_prefix)) {
          uint32_t run_length_prefix = orc_log2_floor_nonzero(reps);
          uint32_t extra_bits = reps - (1u << run_length_prefix);
          v[*out_size] = run_length_prefix + (extra_bits << 9);
          (*out_size)++;
          break;
        } else {
          uint32_t extra_bits = (1u << max_prefix) - 1;
          v[*out_size] = max_prefix + (extra_bits << 9);
          reps -= (2u << max_prefix) - 1;
          (*out_size)++;
        }
      }
    }
  }
}

/* brotli_bit_stream.rs:1783-1858 */
static void encode_context_map(const uint32_t* context_map, size_t context_map_size, size_t num_clusters,
                               HuffmanTree* tree, size_t* storage_ix, uint8_t* storage) {
  uint32_t max_run_length_prefix = 6;
  size_t num_rle_symbols = 0;
  const uint32_t kSymbolMask = (1u << 9) - 1;
  uint8_t depths[272] = {0};
  uint16_t bits[272] = {0};
  uint32_t histogram[272] = {0};
  store_var_len_uint8(num_clusters - 1, storage_ix, storage);
  if (num_clusters == 1) return;
  uint32_t* rle_symbols = (uint32_t*)calloc(context_map_size ? context_map_size : 1, 4);
  move_to_front_transform(context_map, context_map_size, rle_symbols);
  run_length_code_zeros(context_map_size, rle_symbols, &num_rle_symbols, &max_run_length_prefix);
  for (size_t i = 0; i < num_rle_symbols; ++i) histogram[rle_symbols[i] & kSymbolMask]++;
  {
    int use_rle = max_run_length_prefix > 0;
    orc_write_bits(1, (uint64_t)use_rle, storage_ix, storage);
    if (use_rle) orc_write_bits(4, max_run_length_prefix - 1, storage_ix, storage);
  }
  build_and_store_huffman_tree(histogram, num_clusters + max_run_length_prefix, num_clusters + max_run_length_prefix,
                               tree, depths, bits, storage_ix, storage);
  for (size_t i = 0; i < num_rle_symbols; ++i) {
    uint32_t rle_symbol = rle_symbols[i] & kSymbolMask;
    uint32_t extra_bits_val = rle_symbols[i] >> 9;
    orc_write_bits(depths[rle_symbol], bits[rle_symbol], storage_ix, storage);
    if (rle_symbol > 0 && rle_symbol <= max_run_length_prefix) orc_write_bits(rle_symbol, extra_bits_val, storage_ix, storage);
  }
  orc_write_bits(1, 1, storage_ix, storage);
  free(rle_symbols);
}

typedef struct {
  size_t histogram_length_;
  size_t num_block_types_;
  const uint8_t* block_types_;
  const uint32_t* block_lengths_;
  size_t num_blocks_;
  BlockSplitCode block_split_code_;
  size_t block_ix_;
  size_t block_len_;
  size_t entropy_ix_;
  uint8_t* depths_;
  uint16_t* bits_;
} BlockEncoder;

/* brotli_bit_stream.rs:1319-1355 */
static void block_encoder_init(BlockEncoder* e, size_t histogram_length, const BlockSplit* split) {
  memset(e, 0, sizeof(*e));
  e->histogram_length_ = histogram_length;
  e->num_block_types_ = split->num_types;
  e->block_types_ = split->types;
  e->block_lengths_ = split->lengths;
  e->num_blocks_ = split->num_blocks;
  e->block_split_code_.type_code_calculator.last_type = 1;
  e->block_split_code_.type_code_calculator.second_last_type = 0;
  e->block_len_ = split->num_blocks != 0 ? split->lengths[0] : 0;
}

/* brotli_bit_stream.rs:1860-1889 */
static void block_encoder_build_and_store_entropy_codes(BlockEncoder* e, const uint32_t* histograms,
                                                        size_t histo_stride, size_t histograms_size,
                                                        size_t alphabet_size, HuffmanTree* tree, size_t* storage_ix,
                                                        uint8_t* storage) {
  size_t table_size = histograms_size * e->histogram_length_;
  e->depths_ = (uint8_t*)calloc(table_size ? table_size : 1, 1);
  e->bits_ = (uint16_t*)calloc(table_size ? table_size : 1, 2);
  for (size_t i = 0; i < histograms_size; ++i) {
    size_t ix = i * e->histogram_length_;
    build_and_store_huffman_tree(histograms + i * histo_stride, e->histogram_length_, alphabet_size, tree,
                                 e->depths_ + ix, e->bits_ + ix, storage_ix, storage);
  }
}

/* brotli_bit_stream.rs:1891-1920 */
static inline void block_encoder_store_symbol(BlockEncoder* e, size_t symbol, size_t* storage_ix, uint8_t* storage) {
  if (e->block_len_ == 0) {
    size_t block_ix = ++e->block_ix_;
    uint32_t block_len = e->block_lengths_[block_ix];
    uint8_t block_type = e->block_types_[block_ix];
    e->block_len_ = block_len;
    e->entropy_ix_ = (size_t)block_type * e->histogram_length_;
    store_block_switch(&e->block_split_code_, block_len, block_type, 0, storage_ix, storage);
  }
  e->block_len_--;
  size_t ix = e->entropy_ix_ + symbol;
  orc_write_bits(e->depths_[ix], e->bits_[ix], storage_ix, storage);
}

/* brotli_bit_stream.rs:1980-2020 */
static inline void block_encoder_store_symbol_with_context(BlockEncoder* e, size_t symbol, size_t context,
                                                           const uint32_t* context_map, size_t* storage_ix,
                                                           uint8_t* storage, size_t context_bits) {
  if (e->block_len_ == 0) {
    size_t block_ix = ++e->block_ix_;
    uint32_t block_len = e->block_lengths_[block_ix];
    uint8_t block_type = e->block_types_[block_ix];
    e->block_len_ = block_len;
    e->entropy_ix_ = (size_t)block_type << context_bits;
    store_block_switch(&e->block_split_code_, block_len, block_type, 0, storage_ix, storage);
  }
  e->block_len_--;
  size_t histo_ix = context_map[e->entropy_ix_ + context];
  size_t ix = histo_ix * e->histogram_length_ + symbol;
  orc_write_bits(e->depths_[ix], e->bits_[ix], storage_ix, storage);
}

/* brotli_bit_stream.rs:1272-1290 */
static void encode_mlen(uint32_t length, uint64_t* bits, uint32_t* numbits, uint32_t* nibblesbits) {
  uint32_t lg = length == 1 ? 1 : orc_log2_floor_nonzero((uint64_t)(length - 1)) + 1;
  uint32_t mnibbles = (lg < 16 ? 16 : lg + 3) / 4;
  *nibblesbits = mnibbles - 4;
  *numbits = mnibbles * 4;
  *bits = length - 1;
}

/* brotli_bit_stream.rs:1292-1311 */
static void store_compressed_meta_block_header(int is_final_block, size_t length, size_t* storage_ix,
                                               uint8_t* storage) {
  uint64_t lenbits;
  uint32_t nlenbits, nibblesbits;
  orc_write_bits(1, (uint64_t)is_final_block, storage_ix, storage);
  if (is_final_block) orc_write_bits(1, 0, storage_ix, storage);
  encode_mlen((uint32_t)length, &lenbits, &nlenbits, &nibblesbits);
  orc_write_bits(2, nibblesbits, storage_ix, storage);
  orc_write_bits(nlenbits, lenbits, storage_ix, storage);
  if (!is_final_block) orc_write_bits(1, 0, storage_ix, storage);
}

/* brotli_bit_stream.rs:1947-1961 */
static void store_command_extra(const Command* cmd, size_t* storage_ix, uint8_t* storage) {
  uint32_t copylen_code = orc_command_copy_len_code(cmd);
  uint16_t inscode = orc_get_insert_length_code(cmd->insert_len_);
  uint16_t copycode = orc_get_copy_length_code(copylen_code);
  uint32_t insnumextra = orc_ins_extra()[inscode];
  uint64_t insextraval = (uint64_t)(cmd->insert_len_ - orc_ins_base()[inscode]);
  uint64_t copyextraval = (uint64_t)(copylen_code - orc_copy_base()[copycode]);
  uint64_t bits = (copyextraval << insnumextra) | insextraval;
  orc_write_bits(insnumextra + orc_copy_extra()[copycode], bits, storage_ix, storage);
}

/* command.rs:28-36 */
uint32_t orc_command_distance_context(const Command* c) {
  uint32_t r = (uint32_t)(c->cmd_prefix_ >> 6);
  uint32_t cc = (uint32_t)(c->cmd_prefix_ & 7);
  if ((r == 0 || r == 2 || r == 4 || r == 7) && cc <= 2) return cc;
  return 3;
}

/* brotli_bit_stream.rs:2035-2261 */
void orc_store_meta_block(const uint8_t* input, size_t start_pos, size_t length, size_t mask, uint8_t prev_byte,
                          uint8_t prev_byte2, int is_last, const EncoderParams* params, int literal_context_mode,
                          const Command* commands, size_t n_commands, MetaBlockSplit* mb, size_t* storage_ix,
                          uint8_t* storage) {
  size_t pos = start_pos;
  size_t num_distance_symbols = params->dist.alphabet_size;
  size_t num_effective_distance_symbols = num_distance_symbols;
  BlockEncoder literal_enc, command_enc, distance_enc;
  const DistanceParams* dist = &params->dist;
  if (params->large_window && num_effective_distance_symbols > ORC_NUM_DISTANCE_HISTO_SYMBOLS)
    num_effective_distance_symbols = ORC_NUM_DISTANCE_HISTO_SYMBOLS;
  store_compressed_meta_block_header(is_last, length, storage_ix, storage);
  HuffmanTree* tree = (HuffmanTree*)malloc((2 * 704 + 1) * sizeof(HuffmanTree));
  block_encoder_init(&literal_enc, 256, &mb->literal_split);
  block_encoder_init(&command_enc, 704, &mb->command_split);
  block_encoder_init(&distance_enc, num_effective_distance_symbols, &mb->distance_split);
  build_and_store_block_split_code(literal_enc.block_types_, literal_enc.block_lengths_, literal_enc.num_blocks_,
                                   literal_enc.num_block_types_, tree, &literal_enc.block_split_code_, storage_ix,
                                   storage);
  build_and_store_block_split_code(command_enc.block_types_, command_enc.block_lengths_, command_enc.num_blocks_,
                                   command_enc.num_block_types_, tree, &command_enc.block_split_code_, storage_ix,
                                   storage);
  build_and_store_block_split_code(distance_enc.block_types_, distance_enc.block_lengths_, distance_enc.num_blocks_,
                                   distance_enc.num_block_types_, tree, &distance_enc.block_split_code_, storage_ix,
                                   storage);
  orc_write_bits(2, dist->distance_postfix_bits, storage_ix, storage);
  orc_write_bits(4, dist->num_direct_distance_codes >> dist->distance_postfix_bits, storage_ix, storage);
  for (size_t i = 0; i < mb->literal_split.num_types; ++i) orc_write_bits(2, (uint64_t)literal_context_mode, storage_ix, storage);
  if (mb->literal_context_map_size == 0) {
    store_trivial_context_map(mb->literal_histograms_size, 6, tree, storage_ix, storage);
  } else {
    encode_context_map(mb->literal_context_map, mb->literal_context_map_size, mb->literal_histograms_size, tree,
                       storage_ix, storage);
  }
  if (mb->distance_context_map_size == 0) {
    store_trivial_context_map(mb->distance_histograms_size, 2, tree, storage_ix, storage);
  } else {
    encode_context_map(mb->distance_context_map, mb->distance_context_map_size, mb->distance_histograms_size, tree,
                       storage_ix, storage);
  }
  block_encoder_build_and_store_entropy_codes(&literal_enc, mb->literal_histograms, 256, mb->literal_histograms_size,
                                              256, tree, storage_ix, storage);
  block_encoder_build_and_store_entropy_codes(&command_enc, mb->command_histograms, 704, mb->command_histograms_size,
                                              704, tree, storage_ix, storage);
  block_encoder_build_and_store_entropy_codes(&distance_enc, mb->distance_histograms, ORC_NUM_DISTANCE_HISTO_SYMBOLS,
                                              mb->distance_histograms_size, num_distance_symbols, tree, storage_ix,
                                              storage);
  free(tree);
  for (size_t i = 0; i < n_commands; ++i) {
    const Command cmd = commands[i];
    block_encoder_store_symbol(&command_enc, cmd.cmd_prefix_, storage_ix, storage);
    store_command_extra(&cmd, storage_ix, storage);
    if (mb->literal_context_map_size == 0) {
      for (size_t j = cmd.insert_len_; j != 0; --j) {
        block_encoder_store_symbol(&literal_enc, input[pos & mask], storage_ix, storage);
        pos++;
      }
    } else {
      for (size_t j = cmd.insert_len_; j != 0; --j) {
        size_t context = orc_context(prev_byte, prev_byte2, literal_context_mode);
        uint8_t literal = input[pos & mask];
        block_encoder_store_symbol_with_context(&literal_enc, literal, context, mb->literal_context_map, storage_ix,
                                                storage, 6);
        prev_byte2 = prev_byte;
        prev_byte = literal;
        pos++;
      }
    }
    pos += orc_command_copy_len(&cmd);
    if (orc_command_copy_len(&cmd) != 0) {
      prev_byte2 = input[(pos - 2) & mask];
      prev_byte = input[(pos - 1) & mask];
      if (cmd.cmd_prefix_ >= 128) {
        size_t dist_code = cmd.dist_prefix_ & 0x3ff;
        uint32_t distnumextra = (uint32_t)cmd.dist_prefix_ >> 10;
        uint64_t distextra = cmd.dist_extra_;
        if (mb->distance_context_map_size == 0) {
          block_encoder_store_symbol(&distance_enc, dist_code, storage_ix, storage);
        } else {
          block_encoder_store_symbol_with_context(&distance_enc, dist_code, orc_command_distance_context(&cmd),
                                                  mb->distance_context_map, storage_ix, storage, 2);
        }
        orc_write_bits(distnumextra, distextra, storage_ix, storage);
      }
    }
  }
  free(distance_enc.depths_);
  free(distance_enc.bits_);
  free(command_enc.depths_);
  free(command_enc.bits_);
  free(literal_enc.depths_);
  free(literal_enc.bits_);
  if (is_last) jump_to_byte_boundary(storage_ix, storage);
}

/* ------------------------------------------------------------------ quality 2 / 3 writers */
#include "../tables/brotli_fast_tables.h"

/* brotli_bit_stream.rs:925-1121 */
static void build_and_store_huffman_tree_fast(const uint32_t* histogram, size_t histogram_total, size_t max_bits,
                                              uint8_t* depth, uint16_t* bits, size_t* storage_ix, uint8_t* storage) {
  uint64_t count = 0;
  uint64_t symbols[4] = {0, 0, 0, 0};
  uint64_t length = 0;
  size_t total = histogram_total;
  while (total != 0) {
    if (histogram[length] != 0) {
      if (count < 4) symbols[count] = length;
      ++count;
      total -= histogram[length];
    }
    ++length;
  }
  if (count <= 1) {
    orc_write_bits(4, 1, storage_ix, storage);
    orc_write_bits((unsigned)max_bits, symbols[0], storage_ix, storage);
    depth[symbols[0]] = 0;
    bits[symbols[0]] = 0;
    return;
  }
  memset(depth, 0, length);
  {
    HuffmanTree* tree = (HuffmanTree*)malloc((2 * length + 1) * sizeof(HuffmanTree));
    const HuffmanTree sentinel = {0xffffffffu, -1, -1};
    for (uint32_t count_limit = 1;; count_limit *= 2) {
      uint32_t node_index = 0;
      for (uint64_t l = length; l != 0;) {
        --l;
        if (histogram[l] != 0) {
          tree[node_index].total_count_ = histogram[l] >= count_limit ? histogram[l] : count_limit;
          tree[node_index].index_left_ = -1;
          tree[node_index].index_right_or_value_ = (int16_t)l;
          ++node_index;
        }
      }
      {
        int n = (int)node_index;
        int i = 0, j = n + 1, k;
        sort_simple = 1;
        sort_huffman_tree_items(tree, (size_t)n);
        sort_simple = 0;
        tree[node_index + 1] = sentinel;
        tree[node_index] = sentinel;
        node_index += 2;
        for (k = n - 1; k > 0; --k) {
          int left, right;
          if (tree[i].total_count_ <= tree[j].total_count_) {
            left = i++;
          } else {
            left = j++;
          }
          if (tree[i].total_count_ <= tree[j].total_count_) {
            right = i++;
          } else {
            right = j++;
          }
          tree[node_index - 1].total_count_ = tree[left].total_count_ + tree[right].total_count_;
          tree[node_index - 1].index_left_ = (int16_t)left;
          tree[node_index - 1].index_right_or_value_ = (int16_t)right;
          tree[node_index] = sentinel;
          ++node_index;
        }
        if (set_depth(2 * n - 1, tree, depth, 14)) break;
      }
    }
    free(tree);
  }
  convert_bit_depths_to_symbols(depth, (size_t)length, bits);
  if (count <= 4) {
    orc_write_bits(2, 1, storage_ix, storage);
    orc_write_bits(2, count - 1, storage_ix, storage);
    for (size_t i = 0; i < count; ++i)
      for (size_t j = i + 1; j < count; ++j)
        if (depth[symbols[j]] < depth[symbols[i]]) {
          uint64_t t = symbols[j];
          symbols[j] = symbols[i];
          symbols[i] = t;
        }
    for (size_t i = 0; i < count; ++i) orc_write_bits((unsigned)max_bits, symbols[i], storage_ix, storage);
    if (count == 4) orc_write_bits(1, depth[symbols[0]] == 1 ? 1 : 0, storage_ix, storage);
  } else {
    uint8_t previous_value = 8;
    orc_write_bits(40, 0xff55555554ull, storage_ix, storage); /* StoreStaticCodeLengthCode :913-915 */
    for (uint64_t i = 0; i < length;) {
      const uint8_t value = depth[i];
      uint64_t reps = 1;
      for (uint64_t k = i + 1; k < length && depth[k] == value; ++k) ++reps;
      i += reps;
      if (value == 0) {
        orc_write_bits(kZeroRepsDepth[reps], kZeroRepsBits[reps], storage_ix, storage);
      } else {
        if (previous_value != value) {
          orc_write_bits(kCodeLengthDepth[value], kCodeLengthBits[value], storage_ix, storage);
          --reps;
        }
        if (reps < 3) {
          while (reps != 0) {
            --reps;
            orc_write_bits(kCodeLengthDepth[value], kCodeLengthBits[value], storage_ix, storage);
          }
        } else {
          reps -= 3;
          orc_write_bits(kNonZeroRepsDepth[reps], kNonZeroRepsBits[reps], storage_ix, storage);
        }
        previous_value = value;
      }
    }
  }
}

/* brotli_bit_stream.rs:2263-2290 */
static void build_histograms(const uint8_t* input, size_t start_pos, size_t mask, const Command* commands,
                             size_t n_commands, uint32_t* lit_histo, size_t* lit_total, uint32_t* cmd_histo,
                             size_t* cmd_total, uint32_t* dist_histo, size_t* dist_total) {
  size_t pos = start_pos;
  for (size_t i = 0; i < n_commands; ++i) {
    const Command cmd = commands[i];
    cmd_histo[cmd.cmd_prefix_]++;
    ++*cmd_total;
    for (size_t j = cmd.insert_len_; j != 0; --j) {
      lit_histo[input[pos & mask]]++;
      ++*lit_total;
      ++pos;
    }
    pos += orc_command_copy_len(&cmd);
    if (orc_command_copy_len(&cmd) != 0 && cmd.cmd_prefix_ >= 128) {
      dist_histo[cmd.dist_prefix_ & 0x03ff]++;
      ++*dist_total;
    }
  }
}

/* brotli_bit_stream.rs:2292-2343 */
static void store_data_with_huffman_codes(const uint8_t* input, size_t start_pos, size_t mask, const Command* commands,
                                          size_t n_commands, const uint8_t* lit_depth, const uint16_t* lit_bits,
                                          const uint8_t* cmd_depth, const uint16_t* cmd_bits, const uint8_t* dist_depth,
                                          const uint16_t* dist_bits, size_t* storage_ix, uint8_t* storage) {
  size_t pos = start_pos;
  for (size_t i = 0; i < n_commands; ++i) {
    const Command cmd = commands[i];
    const size_t cmd_code = cmd.cmd_prefix_;
    orc_write_bits(cmd_depth[cmd_code], cmd_bits[cmd_code], storage_ix, storage);
    store_command_extra(&cmd, storage_ix, storage);
    for (size_t j = cmd.insert_len_; j != 0; --j) {
      const uint8_t literal = input[pos & mask];
      orc_write_bits(lit_depth[literal], lit_bits[literal], storage_ix, storage);
      ++pos;
    }
    pos += orc_command_copy_len(&cmd);
    if (orc_command_copy_len(&cmd) != 0 && cmd.cmd_prefix_ >= 128) {
      const size_t dist_code = cmd.dist_prefix_ & 0x03ff;
      const uint32_t distnumextra = (uint32_t)cmd.dist_prefix_ >> 10;
      orc_write_bits(dist_depth[dist_code], dist_bits[dist_code], storage_ix, storage);
      orc_write_bits(distnumextra, cmd.dist_extra_, storage_ix, storage);
    }
  }
}

#define ORC_MAX_SIMPLE_DISTANCE_ALPHABET_SIZE 140

/* brotli_bit_stream.rs:2345-2465 (quality 3) */
void orc_store_meta_block_trivial(const uint8_t* input, size_t start_pos, size_t length, size_t mask, int is_last,
                                  const EncoderParams* params, const Command* commands, size_t n_commands,
                                  size_t* storage_ix, uint8_t* storage) {
  uint32_t lit_histo[256] = {0}, cmd_histo[704] = {0}, dist_histo[ORC_NUM_DISTANCE_HISTO_SYMBOLS] = {0};
  size_t lit_total = 0, cmd_total = 0, dist_total = 0;
  uint8_t lit_depth[256] = {0}, cmd_depth[704] = {0}, dist_depth[ORC_MAX_SIMPLE_DISTANCE_ALPHABET_SIZE] = {0};
  uint16_t lit_bits[256] = {0}, cmd_bits[704] = {0}, dist_bits[ORC_MAX_SIMPLE_DISTANCE_ALPHABET_SIZE] = {0};
  HuffmanTree* tree = (HuffmanTree*)malloc((2 * 704 + 1) * sizeof(HuffmanTree));
  size_t num_distance_symbols = params->dist.alphabet_size;
  store_compressed_meta_block_header(is_last, length, storage_ix, storage);
  build_histograms(input, start_pos, mask, commands, n_commands, lit_histo, &lit_total, cmd_histo, &cmd_total, dist_histo,
                   &dist_total);
  orc_write_bits(13, 0, storage_ix, storage);
  build_and_store_huffman_tree(lit_histo, 256, 256, tree, lit_depth, lit_bits, storage_ix, storage);
  build_and_store_huffman_tree(cmd_histo, 704, 704, tree, cmd_depth, cmd_bits, storage_ix, storage);
  build_and_store_huffman_tree(dist_histo, ORC_MAX_SIMPLE_DISTANCE_ALPHABET_SIZE, num_distance_symbols, tree, dist_depth,
                               dist_bits, storage_ix, storage);
  free(tree);
  store_data_with_huffman_codes(input, start_pos, mask, commands, n_commands, lit_depth, lit_bits, cmd_depth, cmd_bits,
                                dist_depth, dist_bits, storage_ix, storage);
  if (is_last) jump_to_byte_boundary(storage_ix, storage);
}

/* brotli_bit_stream.rs:2578-2742 (quality 2) */
void orc_store_meta_block_fast(const uint8_t* input, size_t start_pos, size_t length, size_t mask, int is_last,
                               const EncoderParams* params, const Command* commands, size_t n_commands,
                               size_t* storage_ix, uint8_t* storage) {
  uint32_t num_distance_symbols = params->dist.alphabet_size;
  uint32_t distance_alphabet_bits = orc_log2_floor_nonzero((uint64_t)num_distance_symbols - 1) + 1;
  store_compressed_meta_block_header(is_last, length, storage_ix, storage);
  orc_write_bits(13, 0, storage_ix, storage);
  if (n_commands <= 128) {
    uint32_t histogram[256] = {0};
    size_t pos = start_pos, num_literals = 0;
    uint8_t lit_depth[256] = {0};
    uint16_t lit_bits[256] = {0};
    for (size_t i = 0; i < n_commands; ++i) {
      const Command cmd = commands[i];
      for (size_t j = cmd.insert_len_; j != 0; --j) {
        histogram[input[pos & mask]]++;
        ++pos;
      }
      num_literals += cmd.insert_len_;
      pos += orc_command_copy_len(&cmd);
    }
    build_and_store_huffman_tree_fast(histogram, num_literals, 8, lit_depth, lit_bits, storage_ix, storage);
    orc_write_bits(56, 0x0092624416307003ull, storage_ix, storage); /* StoreStaticCommandHuffmanTree :2467-2470 */
    orc_write_bits(3, 0, storage_ix, storage);
    orc_write_bits(28, 0x0369dc03, storage_ix, storage); /* StoreStaticDistanceHuffmanTree :2472-2474 */
    store_data_with_huffman_codes(input, start_pos, mask, commands, n_commands, lit_depth, lit_bits,
                                  kStaticCommandCodeDepth, kStaticCommandCodeBits, kStaticDistanceCodeDepth,
                                  kStaticDistanceCodeBits, storage_ix, storage);
  } else {
    uint32_t lit_histo[256] = {0}, cmd_histo[704] = {0}, dist_histo[ORC_NUM_DISTANCE_HISTO_SYMBOLS] = {0};
    size_t lit_total = 0, cmd_total = 0, dist_total = 0;
    uint8_t lit_depth[256] = {0}, cmd_depth[704] = {0}, dist_depth[ORC_NUM_DISTANCE_HISTO_SYMBOLS] = {0};
    uint16_t lit_bits[256] = {0}, cmd_bits[704] = {0}, dist_bits[ORC_NUM_DISTANCE_HISTO_SYMBOLS] = {0};
    build_histograms(input, start_pos, mask, commands, n_commands, lit_histo, &lit_total, cmd_histo, &cmd_total,
                     dist_histo, &dist_total);
    build_and_store_huffman_tree_fast(lit_histo, lit_total, 8, lit_depth, lit_bits, storage_ix, storage);
    build_and_store_huffman_tree_fast(cmd_histo, cmd_total, 10, cmd_depth, cmd_bits, storage_ix, storage);
    build_and_store_huffman_tree_fast(dist_histo, dist_total, distance_alphabet_bits, dist_depth, dist_bits, storage_ix,
                                      storage);
    store_data_with_huffman_codes(input, start_pos, mask, commands, n_commands, lit_depth, lit_bits, cmd_depth, cmd_bits,
                                  dist_depth, dist_bits, storage_ix, storage);
  }
  if (is_last) jump_to_byte_boundary(storage_ix, storage);
}

/* brotli_bit_stream.rs:2743-2756, 2775-2833 */
void orc_store_uncompressed_meta_block(int is_final_block, const uint8_t* input, size_t position, size_t mask,
                                       size_t len, size_t* storage_ix, uint8_t* storage) {
  size_t masked_pos = position & mask;
  uint64_t lenbits;
  uint32_t nlenbits, nibblesbits;
  orc_write_bits(1, 0, storage_ix, storage);
  encode_mlen((uint32_t)len, &lenbits, &nlenbits, &nibblesbits);
  orc_write_bits(2, nibblesbits, storage_ix, storage);
  orc_write_bits(nlenbits, lenbits, storage_ix, storage);
  orc_write_bits(1, 1, storage_ix, storage);
  jump_to_byte_boundary(storage_ix, storage);
  if (masked_pos + len > mask + 1) {
    size_t len1 = mask + 1 - masked_pos;
    memcpy(&storage[*storage_ix >> 3], &input[masked_pos], len1);
    *storage_ix += len1 << 3;
    len -= len1;
    masked_pos = 0;
  }
  memcpy(&storage[*storage_ix >> 3], &input[masked_pos], len);
  *storage_ix += len << 3;
  storage[*storage_ix >> 3] = 0;
  if (is_final_block) {
    orc_write_bits(1, 1, storage_ix, storage);
    orc_write_bits(1, 1, storage_ix, storage);
    jump_to_byte_boundary(storage_ix, storage);
  }
}

/* brotli_bit_stream.rs:2840-2851 */
void orc_write_padding_meta_block(size_t* storage_ix, uint8_t* storage) {
  if (*storage_ix & 7) {
    orc_write_bits(6, 6, storage_ix, storage);
    jump_to_byte_boundary(storage_ix, storage);
  }
}
void orc_write_empty_last_meta_block(size_t* storage_ix, uint8_t* storage) {
  orc_write_bits(1, 1, storage_ix, storage);
  orc_write_bits(1, 1, storage_ix, storage);
  jump_to_byte_boundary(storage_ix, storage);
}

/* brotli_bit_stream.rs:2853-2896 */
void orc_write_metadata_meta_block(const EncoderParams* params, size_t* storage_ix, uint8_t* storage) {
  uint8_t b128[10];
  size_t count = 0;
  uint64_t value = params->size_hint;
  for (size_t index = 0; index < 10; ++index) {
    b128[index] = (uint8_t)(value & 0x7f);
    value >>= 7;
    count = index + 1;
    if (value != 0) {
      b128[index] |= 0x80;
    } else {
      break;
    }
  }
  orc_write_bits(1, 0, storage_ix, storage);
  orc_write_bits(2, 3, storage_ix, storage);
  orc_write_bits(1, 0, storage_ix, storage);
  orc_write_bits(2, 1, storage_ix, storage);
  orc_write_bits(8, 3 + count, storage_ix, storage);
  jump_to_byte_boundary(storage_ix, storage);
  uint8_t magic[3] = {0xe1, 0x97, 0x80};
  if (params->catable && !params->use_dictionary) {
    magic[2] = 0x81;
  } else if (params->appendable) {
    magic[2] = 0x82;
  }
  for (int i = 0; i < 3; ++i) orc_write_bits(8, magic[i], storage_ix, storage);
  orc_write_bits(8, 1 /* VERSION, src/lib.rs:67 */, storage_ix, storage);
  for (size_t i = 0; i < count; ++i) orc_write_bits(8, b128[i], storage_ix, storage);
}

/* ------------------------------------------------------------------ exports for orc_fragment.c (qualities 0 / 1) */
void orc_create_huffman_tree(const uint32_t* data, size_t length, int tree_limit, uint8_t* depth) {
  HuffmanTree* tree = (HuffmanTree*)malloc((2 * length + 1) * sizeof(HuffmanTree));
  create_huffman_tree(data, length, tree_limit, tree, depth);
  free(tree);
}
void orc_store_huffman_tree(const uint8_t* depths, size_t num, size_t* storage_ix, uint8_t* storage) {
  HuffmanTree* tree = (HuffmanTree*)malloc((2 * 704 + 1) * sizeof(HuffmanTree));
  store_huffman_tree(depths, num, tree, storage_ix, storage);
  free(tree);
}
void orc_convert_bit_depths_to_symbols(const uint8_t* depth, size_t len, uint16_t* bits) {
  convert_bit_depths_to_symbols(depth, len, bits);
}
void orc_build_and_store_huffman_tree_fast(const uint32_t* histogram, size_t histogram_total, size_t max_bits,
                                           uint8_t* depth, uint16_t* bits, size_t* storage_ix, uint8_t* storage) {
  build_and_store_huffman_tree_fast(histogram, histogram_total, max_bits, depth, bits, storage_ix, storage);
}
