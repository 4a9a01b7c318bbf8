/* oracle/orc_zopfli.c -- CPU restatement of rust-brotli's quality 10 / 11 LZ77 stage: the H10 binary-tree hasher and the
 * Zopfli-style shortest-path parse.  TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows:
 *   src/enc/backward_references/hash_to_binary_tree.rs:149-190 (InitializeH10), :262-330 (AnyHasher for H10),
 *                                                     :437-530 (StoreAndFindMatchesH10)
 *   src/enc/backward_references/hq.rs:57-148   ZopfliNode accessors, BrotliZopfliCreateCommands
 *                                     :150-252  MaxZopfliLen, ZopfliCostModel (init, set_from_literal_costs)
 *                                     :254-300  StitchToPreviousBlockH10
 *                                     :301-412  FindAllMatchesH10
 *                                     :414-855  queue, EvaluateNode, UpdateNodes
 *                                     :857-1041 ComputeShortestPathFromNodes, BrotliZopfliComputeShortestPath,
 *                                               BrotliCreateZopfliBackwardReferences
 *                                     :1043-1160 SetCost, set_from_commands
 *                                     :1162-1448 ZopfliIterate, BrotliCreateHqZopfliBackwardReferences
 *   src/enc/literal_cost.rs:8-239              BrotliEstimateBitCostsForLiterals
 * floatX = f32.  The node's `u` is a Rust enum (cost / next / shortcut); reading it as another variant yields 0, which the
 * tag below reproduces.
 *
 * Pins (tests/test_oracle.py): alice29 at quality 10 / 11, lgwin 22 -> exactly 47 488 / 46 493 bytes
 * (src/bin/integration_tests.rs:401-449).
 */
#include <math.h>
#include <stdio.h>

#include "orc_internal.h"

#define BUCKET_BITS_H10 17
#define MAX_TREE_COMP_LENGTH 128
#define MAX_TREE_SEARCH_DEPTH 64
#define STORE_LOOKAHEAD_H10 128
#define MAX_NUM_MATCHES_H10 128
#define BROTLI_WINDOW_GAP 16
#define kInvalidMatch 0x0fffffffu
static const float kInfinity = 1.7e38f;

static inline uint32_t load32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static size_t find_match_length_with_limit(const uint8_t* s1, const uint8_t* s2, size_t limit) {
  size_t i = 0;
  while (i < limit && s1[i] == s2[i]) ++i;
  return i;
}
/* mod.rs:42-54 */
static size_t fix_unbroken_len(size_t unbroken_len, size_t prev_ix, size_t ring_buffer_break) {
  if (ring_buffer_break != 0) {
    if (prev_ix < ring_buffer_break && prev_ix + unbroken_len > ring_buffer_break) return ring_buffer_break - prev_ix;
  }
  return unbroken_len;
}

/* ------------------------------------------------------------------ H10 */
/* hash_to_binary_tree.rs:149-190 (BrotliMakeHasher passes one_shot = false: the forest always has 2 << lgwin slots) */
int orc_h10_init(Hasher* h, const EncoderParams* params, size_t ringbuffer_break) {
  size_t num_nodes = (size_t)1 << params->lgwin;
  h->kind = 10;
  h->window_mask_ = ((size_t)1 << params->lgwin) - 1;
  h->invalid_pos_ = 0u - (uint32_t)h->window_mask_;
  h->ringbuffer_break = ringbuffer_break;
  h->bucket_count = (size_t)1 << BUCKET_BITS_H10;
  h->buckets = (uint32_t*)malloc(h->bucket_count * sizeof(uint32_t));
  h->forest = (uint32_t*)calloc(num_nodes * 2, sizeof(uint32_t));
  if (!h->buckets || !h->forest) return 0;
  for (size_t i = 0; i < h->bucket_count; ++i) h->buckets[i] = h->invalid_pos_;
  return 1;
}

/* hash_to_binary_tree.rs:324-335.  Returns 1 if newly prepared. */
int orc_h10_prepare(Hasher* h) {
  if (h->is_prepared_ != 0) return 0;
  for (size_t i = 0; i < h->bucket_count; ++i) h->buckets[i] = h->invalid_pos_;
  h->is_prepared_ = 1;
  return 1;
}

static inline size_t h10_hash(const uint8_t* data) { /* :279-282 */
  return (size_t)((load32(data) * 0x1e35a7bdu) >> (32 - BUCKET_BITS_H10));
}
#define LEFT_CHILD(h, pos) (2 * ((pos) & (h)->window_mask_))
#define RIGHT_CHILD(h, pos) (2 * ((pos) & (h)->window_mask_) + 1)

/* hash_to_binary_tree.rs:437-530.  matches == NULL means "no room for matches" (the empty slice of Store). */
static size_t h10_store_and_find_matches(Hasher* h, const uint8_t* data, size_t cur_ix, size_t ring_buffer_mask,
                                         size_t ringbuffer_break, size_t max_length, size_t max_backward,
                                         size_t* best_len, uint64_t* matches, size_t matches_cap) {
  size_t matches_offset = 0;
  size_t cur_ix_masked = cur_ix & ring_buffer_mask;
  size_t max_comp_len = ORC_MIN(max_length, (size_t)MAX_TREE_COMP_LENGTH);
  int should_reroot_tree = max_length >= MAX_TREE_COMP_LENGTH;
  size_t key = h10_hash(&data[cur_ix_masked]);
  uint32_t* forest = h->forest;
  size_t prev_ix = h->buckets[key];
  size_t node_left = LEFT_CHILD(h, cur_ix);
  size_t node_right = RIGHT_CHILD(h, cur_ix);
  size_t best_len_left = 0, best_len_right = 0;
  size_t depth_remaining = MAX_TREE_SEARCH_DEPTH;
  if (should_reroot_tree) h->buckets[key] = (uint32_t)cur_ix;
  for (;;) {
    size_t backward = cur_ix - prev_ix;
    size_t prev_ix_masked = prev_ix & ring_buffer_mask;
    if (backward == 0 || backward > max_backward || depth_remaining == 0) {
      if (should_reroot_tree) {
        forest[node_left] = h->invalid_pos_;
        forest[node_right] = h->invalid_pos_;
      }
      break;
    }
    size_t cur_len = ORC_MIN(best_len_left, best_len_right);
    size_t len = fix_unbroken_len(cur_len + find_match_length_with_limit(&data[cur_ix_masked + cur_len],
                                                                         &data[prev_ix_masked + cur_len],
                                                                         max_length - cur_len),
                                  prev_ix_masked, ringbuffer_break);
    if (matches_offset != matches_cap && len > *best_len) {
      *best_len = len;
      matches[matches_offset++] = (uint64_t)(uint32_t)backward | ((uint64_t)(uint32_t)(len << 5) << 32);
    }
    if (len >= max_comp_len) {
      if (should_reroot_tree) {
        forest[node_left] = forest[LEFT_CHILD(h, prev_ix)];
        forest[node_right] = forest[RIGHT_CHILD(h, prev_ix)];
      }
      break;
    }
    if (data[cur_ix_masked + len] > data[prev_ix_masked + len]) {
      best_len_left = len;
      if (should_reroot_tree) forest[node_left] = (uint32_t)prev_ix;
      node_left = RIGHT_CHILD(h, prev_ix);
      prev_ix = forest[node_left];
    } else {
      best_len_right = len;
      if (should_reroot_tree) forest[node_right] = (uint32_t)prev_ix;
      node_right = LEFT_CHILD(h, prev_ix);
      prev_ix = forest[node_right];
    }
    --depth_remaining;
  }
  return matches_offset;
}

/* hash_to_binary_tree.rs:283-296 */
void orc_h10_store(Hasher* h, const uint8_t* data, size_t mask, size_t ix) {
  size_t max_backward = h->window_mask_ - 16 + 1;
  size_t best_len = 0;
  h10_store_and_find_matches(h, data, ix, mask, h->ringbuffer_break, MAX_TREE_COMP_LENGTH, max_backward, &best_len, NULL,
                             0);
}

/* hash_to_binary_tree.rs:297-318 */
static void h10_store_range(Hasher* h, const uint8_t* data, size_t mask, size_t ix_start, size_t ix_end) {
  size_t i = ix_start, j = ix_start;
  if (ix_start + 63 <= ix_end) i = ix_end - 63;
  if (ix_start + 512 <= i) {
    for (; j < i; j += 8) orc_h10_store(h, data, mask, j);
  }
  for (; i < ix_end; ++i) orc_h10_store(h, data, mask, i);
}

/* hq.rs:254-300 */
void orc_h10_stitch(Hasher* h, size_t num_bytes, size_t position, const uint8_t* ringbuffer, size_t ringbuffer_mask) {
  if (num_bytes >= 4 - 1 && position >= MAX_TREE_COMP_LENGTH) {
    size_t i_start = position - MAX_TREE_COMP_LENGTH;
    size_t i_end = ORC_MIN(position, i_start + num_bytes);
    for (size_t i = i_start; i < i_end; ++i) {
      size_t max_backward = h->window_mask_ - ORC_MAX((size_t)(BROTLI_WINDOW_GAP - 1), position - i);
      size_t best_len = 0;
      h10_store_and_find_matches(h, ringbuffer, i, ringbuffer_mask, h->ringbuffer_break, MAX_TREE_COMP_LENGTH,
                                 max_backward, &best_len, NULL, 0);
    }
  }
}

/* hq.rs:301-412 */
static size_t find_all_matches_h10(Hasher* h, int use_dictionary, const uint8_t* data, size_t ring_buffer_mask,
                                   size_t ring_buffer_break, size_t cur_ix, size_t max_length, size_t max_backward,
                                   size_t gap, const EncoderParams* params, uint64_t* matches) {
  size_t matches_offset = 0;
  size_t cur_ix_masked = cur_ix & ring_buffer_mask;
  size_t best_len = 1;
  size_t short_match_max_backward = params->quality != 11 ? 16 : 64;
  size_t stop = cur_ix - short_match_max_backward;
  uint32_t dict_matches[38];
  if (cur_ix < short_match_max_backward) stop = 0;
  for (size_t i = cur_ix - 1; i > stop && best_len <= 2; --i) {
    size_t prev_ix = i;
    size_t backward = cur_ix - prev_ix;
    if (backward > max_backward) break;
    prev_ix &= ring_buffer_mask;
    if (data[cur_ix_masked] == data[prev_ix] && data[cur_ix_masked + 1] == data[prev_ix + 1]) {
      size_t len = find_match_length_with_limit(&data[prev_ix], &data[cur_ix_masked], max_length);
      if (len > best_len) {
        best_len = len;
        matches[matches_offset++] = (uint64_t)(uint32_t)backward | ((uint64_t)(uint32_t)(len << 5) << 32);
      }
    }
  }
  if (best_len < max_length) {
    matches_offset += h10_store_and_find_matches(h, data, cur_ix, ring_buffer_mask, ring_buffer_break, max_length,
                                                 max_backward, &best_len, matches + matches_offset,
                                                 MAX_NUM_MATCHES_H10 - matches_offset);
  }
  for (size_t i = 0; i <= 37; ++i) dict_matches[i] = kInvalidMatch;
  {
    size_t minlen = ORC_MAX((size_t)4, best_len + 1);
    if (use_dictionary &&
        orc_find_all_static_dictionary_matches(&data[cur_ix_masked], minlen, max_length, dict_matches) != 0) {
      size_t maxlen = ORC_MIN((size_t)37, max_length);
      for (size_t l = minlen; l <= maxlen; ++l) {
        uint32_t dict_id = dict_matches[l];
        if (dict_id < kInvalidMatch) {
          size_t distance = max_backward + gap + (size_t)(dict_id >> 5) + 1;
          if (distance <= params->dist.max_distance) {
            size_t len_code = dict_id & 31;
            matches[matches_offset++] = (uint64_t)(uint32_t)distance |
                                        ((uint64_t)(uint32_t)((l << 5) | (l == len_code ? 0 : len_code)) << 32);
          }
        }
      }
    }
  }
  return matches_offset;
}
static inline uint32_t match_distance(uint64_t m) { return (uint32_t)m; }
static inline size_t match_length(uint64_t m) { return (size_t)((uint32_t)(m >> 32) >> 5); }
static inline size_t match_length_code(uint64_t m) {
  size_t code = (uint32_t)(m >> 32) & 31u;
  return code != 0 ? code : match_length(m);
}

/* ------------------------------------------------------------------ literal_cost.rs */
static float fast_log2_f64arg(uint64_t v) { return orc_fast_log2(v); } /* FastLog2f64, util.rs:37-45: floatX = f32 */

static size_t utf8_position(size_t last, size_t c, size_t clamp) { /* :8-18 */
  if (c < 128) return 0;
  if (c >= 192) return ORC_MIN((size_t)1, clamp);
  if (last < 0xe0) return 0;
  return ORC_MIN((size_t)2, clamp);
}

static size_t decide_multi_byte_stats_level(size_t pos, size_t len, size_t mask, const uint8_t* data) { /* :20-46 */
  size_t counts[3] = {0, 0, 0};
  size_t max_utf8 = 1;
  size_t last_c = 0;
  for (size_t i = 0; i < len; ++i) {
    size_t c = data[(pos + i) & mask];
    counts[utf8_position(last_c, c, 2)]++;
    last_c = c;
  }
  if (counts[2] < 500) max_utf8 = 1;
  if (counts[1] + counts[2] < 25) max_utf8 = 0;
  return max_utf8;
}

static void estimate_bit_costs_for_literals_utf8(size_t pos, size_t len, size_t mask, const uint8_t* data,
                                                 float* cost) { /* :48-176 */
  size_t max_utf8 = decide_multi_byte_stats_level(pos, len, mask, data);
  static size_t histogram[3][256];
  size_t window_half = 495;
  size_t in_window = ORC_MIN(window_half, len);
  size_t in_window_utf8[3] = {0, 0, 0};
  memset(histogram, 0, sizeof(histogram));
  {
    size_t last_c = 0, utf8_pos = 0;
    for (size_t i = 0; i < in_window; ++i) {
      size_t c = data[(pos + i) & mask];
      histogram[utf8_pos][c]++;
      in_window_utf8[utf8_pos]++;
      utf8_pos = utf8_position(last_c, c, max_utf8);
      last_c = c;
    }
  }
  for (size_t i = 0; i < len; ++i) {
    if (i >= window_half) {
      size_t c = i < window_half + 1 ? 0 : data[(pos + i - window_half - 1) & mask];
      size_t last_c = i < window_half + 2 ? 0 : data[(pos + i - window_half - 2) & mask];
      size_t utf8_pos2 = utf8_position(last_c, c, max_utf8);
      histogram[utf8_pos2][data[(pos + i - window_half) & mask]]--;
      in_window_utf8[utf8_pos2]--;
    }
    if (i + window_half < len) {
      size_t c = data[(pos + i + window_half - 1) & mask];
      size_t last_c = data[(pos + i + window_half - 2) & mask];
      size_t utf8_pos2 = utf8_position(last_c, c, max_utf8);
      histogram[utf8_pos2][data[(pos + i + window_half) & mask]]++;
      in_window_utf8[utf8_pos2]++;
    }
    {
      size_t c = i < 1 ? 0 : data[(pos + i - 1) & mask];
      size_t last_c = i < 2 ? 0 : data[(pos + i - 2) & mask];
      size_t utf8_pos = utf8_position(last_c, c, max_utf8);
      size_t masked_pos = (pos + i) & mask;
      size_t histo = histogram[utf8_pos][data[masked_pos]];
      double lit_cost;
      if (histo == 0) histo = 1;
      lit_cost = (double)fast_log2_f64arg(in_window_utf8[utf8_pos]) - (double)fast_log2_f64arg(histo);
      lit_cost += 0.02905;
      if (lit_cost < 1.0) {
        lit_cost *= 0.5;
        lit_cost += 0.5;
      }
      if (i < 2000) lit_cost += (0.7 - (double)(2000 - i) / 2000.0 * 0.35);
      cost[i] = (float)lit_cost;
    }
  }
}

static void estimate_bit_costs_for_literals(size_t pos, size_t len, size_t mask, const uint8_t* data, float* cost) {
  /* :178-239 */
  if (orc_is_mostly_utf8(data, pos, mask, len, 0.75f)) {
    estimate_bit_costs_for_literals_utf8(pos, len, mask, data, cost);
  } else {
    size_t histogram[256];
    size_t window_half = 2000;
    size_t in_window = ORC_MIN(window_half, len);
    memset(histogram, 0, sizeof(histogram));
    for (size_t i = 0; i < in_window; ++i) histogram[data[(pos + i) & mask]]++;
    for (size_t i = 0; i < len; ++i) {
      size_t histo;
      if (i >= window_half) {
        histogram[data[(pos + i - window_half) & mask]]--;
        in_window--;
      }
      if (i + window_half < len) {
        histogram[data[(pos + i + window_half) & mask]]++;
        in_window++;
      }
      histo = histogram[data[(pos + i) & mask]];
      if (histo == 0) histo = 1;
      {
        double lit_cost = (double)fast_log2_f64arg(in_window) - (double)fast_log2_f64arg(histo);
        lit_cost += 0.029;
        if (lit_cost < 1.0) {
          lit_cost *= 0.5;
          lit_cost += 0.5;
        }
        cost[i] = (float)lit_cost;
      }
    }
  }
}

/* ------------------------------------------------------------------ Zopfli nodes */
enum { U_COST = 0, U_NEXT = 1, U_SHORTCUT = 2 };
typedef struct {
  uint32_t length;
  uint32_t distance;
  uint32_t dcode_insert_length;
  uint32_t tag;
  union {
    float cost;
    uint32_t next;
    uint32_t shortcut;
  } u;
} ZopfliNode;

static inline float node_cost(const ZopfliNode* n) { return n->tag == U_COST ? n->u.cost : 0.0f; }
static inline uint32_t node_next(const ZopfliNode* n) { return n->tag == U_NEXT ? n->u.next : 0; }
static inline uint32_t node_shortcut(const ZopfliNode* n) { return n->tag == U_SHORTCUT ? n->u.shortcut : 0; }
static inline void set_cost(ZopfliNode* n, float c) {
  n->tag = U_COST;
  n->u.cost = c;
}
static inline void set_next(ZopfliNode* n, uint32_t v) {
  n->tag = U_NEXT;
  n->u.next = v;
}
static inline void set_shortcut(ZopfliNode* n, uint32_t v) {
  n->tag = U_SHORTCUT;
  n->u.shortcut = v;
}

static void init_zopfli_nodes(ZopfliNode* array, size_t length) { /* hq.rs:57-66, ZopfliNode::default */
  for (size_t i = 0; i < length; ++i) {
    array[i].length = 1;
    array[i].distance = 0;
    array[i].dcode_insert_length = 0;
    set_cost(&array[i], kInfinity);
  }
}
static inline uint32_t node_copy_length(const ZopfliNode* n) { return n->length & 0x01ffffffu; }
static inline uint32_t node_copy_distance(const ZopfliNode* n) { return n->distance; }
static inline uint32_t node_length_code(const ZopfliNode* n) { return node_copy_length(n) + 9u - (n->length >> 25); }
static inline uint32_t node_distance_code(const ZopfliNode* n) {
  uint32_t short_code = n->dcode_insert_length >> 27;
  return short_code == 0 ? node_copy_distance(n) + 16u - 1u : short_code - 1u;
}
static inline uint32_t node_command_length(const ZopfliNode* n) {
  return node_copy_length(n) + (n->dcode_insert_length & 0x07ffffffu);
}

/* hq.rs:97-148 */
static void zopfli_create_commands(size_t num_bytes, size_t block_start, size_t max_backward_limit,
                                   const ZopfliNode* nodes, int32_t* dist_cache, size_t* last_insert_len,
                                   const EncoderParams* params, Command* commands, size_t* num_literals) {
  size_t pos = 0;
  uint32_t offset = node_next(&nodes[0]);
  const size_t gap = 0;
  for (size_t i = 0; offset != 0xffffffffu; ++i) {
    const ZopfliNode* next = &nodes[pos + offset];
    size_t copy_length = node_copy_length(next);
    size_t insert_length = next->dcode_insert_length & 0x07ffffffu;
    pos += insert_length;
    offset = node_next(next);
    if (i == 0) {
      insert_length += *last_insert_len;
      *last_insert_len = 0;
    }
    {
      size_t distance = node_copy_distance(next);
      size_t len_code = node_length_code(next);
      size_t max_distance = ORC_MIN(block_start + pos, max_backward_limit);
      int is_dictionary = distance > max_distance + gap;
      size_t dist_code = node_distance_code(next);
      orc_command_init(&commands[i], &params->dist, insert_length, copy_length, len_code, dist_code);
      if (!is_dictionary && dist_code > 0) {
        dist_cache[3] = dist_cache[2];
        dist_cache[2] = dist_cache[1];
        dist_cache[1] = dist_cache[0];
        dist_cache[0] = (int32_t)distance;
      }
    }
    *num_literals += insert_length;
    pos += copy_length;
  }
  *last_insert_len += num_bytes - pos;
}

static inline size_t max_zopfli_len(const EncoderParams* p) { return p->quality <= 10 ? 150 : 325; }     /* :150-157 */
static inline size_t max_zopfli_candidates(const EncoderParams* p) { return p->quality <= 10 ? 1 : 5; } /* :422-425 */

/* ------------------------------------------------------------------ cost model (hq.rs:159-252, 1043-1160) */
typedef struct {
  float cost_cmd_[ORC_NUM_COMMAND_SYMBOLS];
  float* cost_dist_;
  uint32_t distance_histogram_size;
  float* literal_costs_;
  float min_cost_cmd_;
  size_t num_bytes_;
} ZopfliCostModel;

static void cost_model_init(ZopfliCostModel* m, const DistanceParams* dist, size_t num_bytes) {
  memset(m, 0, sizeof(*m));
  m->num_bytes_ = num_bytes;
  m->literal_costs_ = (float*)calloc(num_bytes + 2, sizeof(float));
  m->cost_dist_ = (float*)calloc(num_bytes + dist->alphabet_size + 1, sizeof(float));
  m->distance_histogram_size = ORC_MIN(dist->alphabet_size, (uint32_t)544);
}
static void cost_model_cleanup(ZopfliCostModel* m) {
  free(m->literal_costs_);
  free(m->cost_dist_);
}
static void cost_model_set_from_literal_costs(ZopfliCostModel* m, size_t position, const uint8_t* ringbuffer,
                                              size_t ringbuffer_mask) {
  float* literal_costs = m->literal_costs_;
  float literal_carry = 0.0f;
  size_t num_bytes = m->num_bytes_;
  estimate_bit_costs_for_literals(position, num_bytes, ringbuffer_mask, ringbuffer, &literal_costs[1]);
  literal_costs[0] = 0.0f;
  for (size_t i = 0; i < num_bytes; ++i) {
    literal_carry = literal_carry + literal_costs[i + 1];
    literal_costs[i + 1] = literal_costs[i] + literal_carry;
    literal_carry -= literal_costs[i + 1] - literal_costs[i];
  }
  for (size_t i = 0; i < ORC_NUM_COMMAND_SYMBOLS; ++i) m->cost_cmd_[i] = orc_fast_log2(11 + (uint64_t)i);
  for (size_t i = 0; i < m->distance_histogram_size; ++i) m->cost_dist_[i] = orc_fast_log2(20 + (uint64_t)i);
  m->min_cost_cmd_ = orc_fast_log2(11);
}

/* hq.rs:1043-1071 */
static void set_cost_from_histogram(const uint32_t* histogram, size_t histogram_size, int literal_histogram,
                                    float* cost) {
  uint64_t sum = 0;
  for (size_t i = 0; i < histogram_size; ++i) sum += histogram[i];
  float log2sum = orc_fast_log2(sum);
  uint64_t missing_symbol_sum = sum;
  if (!literal_histogram) {
    for (size_t i = 0; i < histogram_size; ++i)
      if (histogram[i] == 0) missing_symbol_sum++;
  }
  float missing_symbol_cost = fast_log2_f64arg(missing_symbol_sum) + 2.0f;
  for (size_t i = 0; i < histogram_size; ++i) {
    if (histogram[i] == 0) {
      cost[i] = missing_symbol_cost;
    } else {
      cost[i] = log2sum - orc_fast_log2(histogram[i]);
      if (cost[i] < 1.0f) cost[i] = 1.0f;
    }
  }
}

/* hq.rs:1073-1160 */
static void cost_model_set_from_commands(ZopfliCostModel* m, size_t position, const uint8_t* ringbuffer,
                                         size_t ringbuffer_mask, const Command* commands, size_t num_commands,
                                         size_t last_insert_len) {
  uint32_t histogram_literal[256];
  uint32_t histogram_cmd[ORC_NUM_COMMAND_SYMBOLS];
  uint32_t histogram_dist[140]; /* BROTLI_SIMPLE_DISTANCE_ALPHABET_SIZE = 16 + 2 * 62 */
  float cost_literal[256];
  size_t pos = position - last_insert_len;
  float min_cost_cmd = kInfinity;
  memset(histogram_literal, 0, sizeof(histogram_literal));
  memset(histogram_cmd, 0, sizeof(histogram_cmd));
  memset(histogram_dist, 0, sizeof(histogram_dist));
  memset(cost_literal, 0, sizeof(cost_literal));
  for (size_t i = 0; i < num_commands; ++i) {
    size_t inslength = commands[i].insert_len_;
    size_t copylength = orc_command_copy_len(&commands[i]);
    size_t distcode = commands[i].dist_prefix_ & 0x03ff;
    size_t cmdcode = commands[i].cmd_prefix_;
    histogram_cmd[cmdcode]++;
    if (cmdcode >= 128) {
      if (distcode >= 140) {
        orc_reference_would_panic = 1; /* index out of bounds in the reference */
      } else {
        histogram_dist[distcode]++;
      }
    }
    for (size_t j = 0; j < inslength; ++j) histogram_literal[ringbuffer[(pos + j) & ringbuffer_mask]]++;
    pos += inslength + copylength;
  }
  set_cost_from_histogram(histogram_literal, 256, 1, cost_literal);
  set_cost_from_histogram(histogram_cmd, ORC_NUM_COMMAND_SYMBOLS, 0, m->cost_cmd_);
  set_cost_from_histogram(histogram_dist, ORC_MIN((size_t)m->distance_histogram_size, (size_t)140), 0, m->cost_dist_);
  for (size_t i = 0; i < 704; ++i) min_cost_cmd = fminf(min_cost_cmd, m->cost_cmd_[i]);
  m->min_cost_cmd_ = min_cost_cmd;
  {
    float* literal_costs = m->literal_costs_;
    float literal_carry = 0.0f;
    size_t num_bytes = m->num_bytes_;
    literal_costs[0] = 0.0f;
    for (size_t i = 0; i < num_bytes; ++i) {
      literal_carry += cost_literal[ringbuffer[(position + i) & ringbuffer_mask]];
      literal_costs[i + 1] = literal_costs[i] + literal_carry;
      literal_carry -= literal_costs[i + 1] - literal_costs[i];
    }
  }
}
static inline float get_literal_costs(const ZopfliCostModel* m, size_t from, size_t to) {
  return m->literal_costs_[to] - m->literal_costs_[from];
}

/* ------------------------------------------------------------------ queue + node evaluation (hq.rs:427-560) */
typedef struct {
  size_t pos;
  int32_t distance_cache[4];
  float costdiff;
  float cost;
} PosData;
typedef struct {
  PosData q_[8];
  size_t idx_;
} StartPosQueue;

static inline size_t queue_size(const StartPosQueue* q) { return ORC_MIN(q->idx_, (size_t)8); }
static void queue_push(StartPosQueue* q, const PosData* posdata) {
  size_t offset = ~q->idx_ & 7;
  q->idx_++;
  size_t len = queue_size(q);
  q->q_[offset] = *posdata;
  for (size_t i = 1; i < len; ++i) {
    if (q->q_[offset & 7].costdiff > q->q_[(offset + 1) & 7].costdiff) {
      PosData t = q->q_[offset & 7];
      q->q_[offset & 7] = q->q_[(offset + 1) & 7];
      q->q_[(offset + 1) & 7] = t;
    }
    ++offset;
  }
}
static inline const PosData* queue_at(const StartPosQueue* q, size_t k) { return &q->q_[(k - q->idx_) & 7]; }

/* hq.rs:427-452 */
static uint32_t compute_distance_shortcut(size_t block_start, size_t pos, size_t max_backward, size_t gap,
                                          const ZopfliNode* nodes) {
  size_t clen = node_copy_length(&nodes[pos]);
  size_t ilen = nodes[pos].dcode_insert_length & 0x07ffffffu;
  size_t dist = node_copy_distance(&nodes[pos]);
  if (pos == 0) return 0;
  if (dist + clen <= block_start + pos + gap && dist <= max_backward + gap && node_distance_code(&nodes[pos]) > 0)
    return (uint32_t)pos;
  return node_shortcut(&nodes[pos - clen - ilen]);
}

/* hq.rs:461-499 */
static void compute_distance_cache(size_t pos, const int32_t* starting_dist_cache, const ZopfliNode* nodes,
                                   int32_t* dist_cache) {
  int idx = 0;
  size_t p = node_shortcut(&nodes[pos]);
  while (idx < 4 && p > 0) {
    size_t ilen = nodes[p].dcode_insert_length & 0x07ffffffu;
    size_t clen = node_copy_length(&nodes[p]);
    size_t dist = node_copy_distance(&nodes[p]);
    dist_cache[idx++] = (int32_t)dist;
    p = node_shortcut(&nodes[p - clen - ilen]);
  }
  for (; idx < 4; ++idx) dist_cache[idx] = *starting_dist_cache++;
}

/* hq.rs:524-560 */
static void evaluate_node(size_t block_start, size_t pos, size_t max_backward_limit, size_t gap,
                          const int32_t* starting_dist_cache, const ZopfliCostModel* model, StartPosQueue* queue,
                          ZopfliNode* nodes) {
  float cost = node_cost(&nodes[pos]);
  set_shortcut(&nodes[pos], compute_distance_shortcut(block_start, pos, max_backward_limit, gap, nodes));
  if (cost <= get_literal_costs(model, 0, pos)) {
    PosData posdata;
    posdata.pos = pos;
    posdata.cost = cost;
    posdata.costdiff = cost - get_literal_costs(model, 0, pos);
    memset(posdata.distance_cache, 0, sizeof(posdata.distance_cache));
    compute_distance_cache(pos, starting_dist_cache, nodes, posdata.distance_cache);
    queue_push(queue, &posdata);
  }
}

/* hq.rs:577-602 */
static size_t compute_minimum_copy_length(float start_cost, const ZopfliNode* nodes, size_t num_bytes, size_t pos) {
  float min_cost = start_cost;
  size_t len = 2;
  size_t next_len_bucket = 4;
  size_t next_len_offset = 10;
  while (pos + len <= num_bytes && node_cost(&nodes[pos + len]) <= min_cost) {
    ++len;
    if (len == next_len_offset) {
      min_cost += 1.0f;
      next_len_offset += next_len_bucket;
      next_len_bucket *= 2;
    }
  }
  return len;
}

/* hq.rs:626-642 */
static inline void update_zopfli_node(ZopfliNode* nodes, size_t pos, size_t start_pos, size_t len, size_t len_code,
                                      size_t dist, size_t short_code, float cost) {
  ZopfliNode* next = &nodes[pos + len];
  next->length = (uint32_t)(len | ((len + 9u - len_code) << 25));
  next->distance = (uint32_t)dist;
  next->dcode_insert_length = (uint32_t)(pos - start_pos) | (uint32_t)(short_code << 27);
  set_cost(next, cost);
}

/* hq.rs:644-829 */
static size_t update_nodes(size_t num_bytes, size_t block_start, size_t pos, const uint8_t* ringbuffer,
                           size_t ringbuffer_mask, size_t ringbuffer_break, const EncoderParams* params,
                           size_t max_backward_limit, const int32_t* starting_dist_cache, size_t num_matches,
                           const uint64_t* matches, const ZopfliCostModel* model, StartPosQueue* queue,
                           ZopfliNode* nodes) {
  static const uint8_t kDistanceCacheIndex[16] = {0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
  static const int8_t kDistanceCacheOffset[16] = {0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3};
  size_t cur_ix = block_start + pos;
  size_t cur_ix_masked = cur_ix & ringbuffer_mask;
  size_t max_distance = ORC_MIN(cur_ix, max_backward_limit);
  size_t max_len = num_bytes - pos;
  size_t max_zlen = max_zopfli_len(params);
  size_t min_len;
  size_t result = 0;
  const size_t gap = 0;
  evaluate_node(block_start, pos, max_backward_limit, gap, starting_dist_cache, model, queue, nodes);
  {
    const PosData* posdata = queue_at(queue, 0);
    float min_cost = posdata->cost + model->min_cost_cmd_ + get_literal_costs(model, posdata->pos, pos);
    min_len = compute_minimum_copy_length(min_cost, nodes, num_bytes, pos);
  }
  size_t kmax = ORC_MIN(max_zopfli_candidates(params), queue_size(queue));
  for (size_t k = 0; k < kmax; ++k) {
    const PosData* posdata = queue_at(queue, k);
    size_t start = posdata->pos;
    uint16_t inscode = orc_get_insert_length_code(pos - start);
    float start_costdiff = posdata->costdiff;
    float base_cost = start_costdiff + (float)orc_ins_extra()[inscode] + get_literal_costs(model, 0, pos);
    size_t best_len = min_len - 1;
    for (size_t j = 0; j < 16; ++j) {
      if (best_len >= max_len) break;
      size_t idx = kDistanceCacheIndex[j];
      size_t backward = (size_t)(int64_t)(posdata->distance_cache[idx & 3] + (int32_t)kDistanceCacheOffset[j]);
      size_t prev_ix = cur_ix - backward;
      size_t len;
      uint8_t continuation = ringbuffer[cur_ix_masked + best_len];
      if (cur_ix_masked + best_len > ringbuffer_mask) break;
      if (backward > max_distance + gap) continue;
      if (backward > max_distance) continue;
      if (prev_ix >= cur_ix) continue;
      prev_ix &= ringbuffer_mask;
      if (prev_ix + best_len > ringbuffer_mask || continuation != ringbuffer[prev_ix + best_len]) continue;
      len = fix_unbroken_len(find_match_length_with_limit(&ringbuffer[prev_ix], &ringbuffer[cur_ix_masked], max_len),
                             prev_ix, ringbuffer_break);
      float dist_cost = base_cost + model->cost_dist_[j];
      for (size_t l = best_len + 1; l <= len; ++l) {
        uint16_t copycode = orc_get_copy_length_code(l);
        uint16_t cmdcode = orc_combine_length_codes(inscode, copycode, j == 0);
        float cost = (cmdcode < 128 ? base_cost : dist_cost) + (float)orc_copy_extra()[copycode] + model->cost_cmd_[cmdcode];
        if (cost < node_cost(&nodes[pos + l])) {
          update_zopfli_node(nodes, pos, start, l, l, backward, j + 1, cost);
          result = ORC_MAX(result, l);
        }
        best_len = l;
      }
    }
    if (k >= 2) continue;
    size_t len = min_len;
    for (size_t j = 0; j < num_matches; ++j) {
      uint64_t match = matches[j];
      size_t dist = match_distance(match);
      int is_dictionary_match = dist > max_distance + gap;
      size_t dist_code = dist + 16 - 1;
      uint16_t dist_symbol = 0;
      uint32_t distextra = 0;
      orc_prefix_encode_copy_distance(dist_code, params->dist.num_direct_distance_codes,
                                      params->dist.distance_postfix_bits, &dist_symbol, &distextra);
      uint32_t distnumextra = (uint32_t)dist_symbol >> 10;
      float dist_cost = base_cost + (float)distnumextra + model->cost_dist_[dist_symbol & 0x03ff];
      size_t max_match_len = match_length(match);
      if (len < max_match_len && (is_dictionary_match || max_match_len > max_zlen)) len = max_match_len;
      for (; len <= max_match_len; ++len) {
        size_t len_code = is_dictionary_match ? match_length_code(match) : len;
        uint16_t copycode = orc_get_copy_length_code(len_code);
        uint16_t cmdcode = orc_combine_length_codes(inscode, copycode, 0);
        float cost = dist_cost + (float)orc_copy_extra()[copycode] + model->cost_cmd_[cmdcode];
        if (nodes[pos + len].tag == U_COST && cost < nodes[pos + len].u.cost) {
          update_zopfli_node(nodes, pos, start, len, len_code, dist, 0, cost);
          result = ORC_MAX(result, len);
        }
      }
    }
  }
  return result;
}

/* hq.rs:857-871 */
static size_t compute_shortest_path_from_nodes(size_t num_bytes, ZopfliNode* nodes) {
  size_t index = num_bytes;
  size_t num_commands = 0;
  while ((nodes[index].dcode_insert_length & 0x07ffffffu) == 0 && nodes[index].length == 1) --index;
  set_next(&nodes[index], 0xffffffffu);
  while (index != 0) {
    size_t len = node_command_length(&nodes[index]);
    index -= len;
    set_next(&nodes[index], (uint32_t)len);
    ++num_commands;
  }
  return num_commands;
}

/* hq.rs:873-988 */
static size_t zopfli_compute_shortest_path(int use_dictionary, size_t num_bytes, size_t position,
                                           const uint8_t* ringbuffer, size_t ringbuffer_mask, size_t ringbuffer_break,
                                           const EncoderParams* params, size_t max_backward_limit,
                                           const int32_t* dist_cache, Hasher* handle, ZopfliNode* nodes) {
  size_t max_zlen = max_zopfli_len(params);
  ZopfliCostModel model;
  StartPosQueue queue;
  uint64_t matches[MAX_NUM_MATCHES_H10];
  size_t store_end = num_bytes >= STORE_LOOKAHEAD_H10 ? position + num_bytes - STORE_LOOKAHEAD_H10 + 1 : position;
  const size_t gap = 0;
  memset(matches, 0, sizeof(matches));
  nodes[0].length = 0;
  set_cost(&nodes[0], 0.0f);
  cost_model_init(&model, &params->dist, num_bytes);
  cost_model_set_from_literal_costs(&model, position, ringbuffer, ringbuffer_mask);
  memset(&queue, 0, sizeof(queue));
  for (size_t i = 0; i + 4 - 1 < num_bytes; ++i) {
    size_t pos = position + i;
    size_t max_distance = ORC_MIN(pos, max_backward_limit);
    size_t skip;
    size_t num_matches = find_all_matches_h10(handle, use_dictionary, ringbuffer, ringbuffer_mask, ringbuffer_break, pos,
                                              num_bytes - i, max_distance, gap, params, matches);
    if (num_matches > 0 && match_length(matches[num_matches - 1]) > max_zlen) {
      matches[0] = matches[num_matches - 1];
      num_matches = 1;
    }
    skip = update_nodes(num_bytes, position, i, ringbuffer, ringbuffer_mask, ringbuffer_break, params,
                        max_backward_limit, dist_cache, num_matches, matches, &model, &queue, nodes);
    if (skip < 16384) skip = 0;
    if (num_matches == 1 && match_length(matches[0]) > max_zlen) skip = ORC_MAX(match_length(matches[0]), skip);
    if (skip > 1) {
      h10_store_range(handle, ringbuffer, ringbuffer_mask, pos + 1, ORC_MIN(pos + skip, store_end));
      --skip;
      while (skip != 0) {
        ++i;
        if (i + 4 - 1 >= num_bytes) break;
        evaluate_node(position, i, max_backward_limit, gap, dist_cache, &model, &queue, nodes);
        --skip;
      }
    }
  }
  cost_model_cleanup(&model);
  return compute_shortest_path_from_nodes(num_bytes, nodes);
}

/* hq.rs:990-1041 */
void orc_create_zopfli_backward_references(size_t num_bytes, size_t position, const uint8_t* ringbuffer,
                                           size_t ringbuffer_mask, size_t ringbuffer_break, const EncoderParams* params,
                                           Hasher* hasher, int32_t* dist_cache, size_t* last_insert_len,
                                           Command* commands, size_t* num_commands, size_t* num_literals) {
  size_t max_backward_limit = ((size_t)1 << params->lgwin) - 16;
  ZopfliNode* nodes = (ZopfliNode*)malloc((num_bytes + 1) * sizeof(ZopfliNode));
  init_zopfli_nodes(nodes, num_bytes + 1);
  *num_commands += zopfli_compute_shortest_path(params->use_dictionary, num_bytes, position, ringbuffer, ringbuffer_mask,
                                                ringbuffer_break, params, max_backward_limit, dist_cache, hasher, nodes);
  zopfli_create_commands(num_bytes, position, max_backward_limit, nodes, dist_cache, last_insert_len, params, commands,
                         num_literals);
  free(nodes);
}

/* hq.rs:1162-1244 */
static size_t zopfli_iterate(size_t num_bytes, size_t position, const uint8_t* ringbuffer, size_t ringbuffer_mask,
                             size_t ringbuffer_break, const EncoderParams* params, size_t max_backward_limit, size_t gap,
                             const int32_t* dist_cache, const ZopfliCostModel* model, const uint32_t* num_matches,
                             const uint64_t* matches, ZopfliNode* nodes) {
  size_t max_zlen = max_zopfli_len(params);
  StartPosQueue queue;
  size_t cur_match_pos = 0;
  nodes[0].length = 0;
  set_cost(&nodes[0], 0.0f);
  memset(&queue, 0, sizeof(queue));
  for (size_t i = 0; i + 3 < num_bytes; ++i) {
    size_t skip = update_nodes(num_bytes, position, i, ringbuffer, ringbuffer_mask, ringbuffer_break, params,
                               max_backward_limit, dist_cache, num_matches[i], &matches[cur_match_pos], model, &queue,
                               nodes);
    if (skip < 16384) skip = 0;
    cur_match_pos += num_matches[i];
    if (num_matches[i] == 1 && match_length(matches[cur_match_pos - 1]) > max_zlen)
      skip = ORC_MAX(match_length(matches[cur_match_pos - 1]), skip);
    if (skip > 1) {
      --skip;
      while (skip != 0) {
        ++i;
        if (i + 3 >= num_bytes) break;
        evaluate_node(position, i, max_backward_limit, gap, dist_cache, model, &queue, nodes);
        cur_match_pos += num_matches[i];
        --skip;
      }
    }
  }
  return compute_shortest_path_from_nodes(num_bytes, nodes);
}

/* hq.rs:1246-1448 */
void orc_create_hq_zopfli_backward_references(size_t num_bytes, size_t position, const uint8_t* ringbuffer,
                                              size_t ringbuffer_mask, size_t ringbuffer_break,
                                              const EncoderParams* params, Hasher* hasher, int32_t* dist_cache,
                                              size_t* last_insert_len, Command* commands, size_t* num_commands,
                                              size_t* num_literals) {
  size_t max_backward_limit = ((size_t)1 << params->lgwin) - 16;
  uint32_t* num_matches = (uint32_t*)calloc(num_bytes ? num_bytes : 1, sizeof(uint32_t));
  size_t matches_size = 4 * num_bytes;
  size_t store_end = num_bytes >= STORE_LOOKAHEAD_H10 ? position + num_bytes - STORE_LOOKAHEAD_H10 + 1 : position;
  size_t cur_match_pos = 0;
  int32_t orig_dist_cache[4];
  ZopfliCostModel model;
  uint64_t* matches = (uint64_t*)calloc(matches_size ? matches_size : 1, sizeof(uint64_t));
  const size_t gap = 0;
  for (size_t i = 0; i + 4 - 1 < num_bytes; ++i) {
    size_t pos = position + i;
    size_t max_distance = ORC_MIN(pos, max_backward_limit);
    size_t max_length = num_bytes - i;
    if (matches_size < cur_match_pos + 128) {
      size_t new_size = matches_size == 0 ? cur_match_pos + 128 : matches_size;
      while (new_size < cur_match_pos + 128) new_size *= 2;
      matches = (uint64_t*)realloc(matches, new_size * sizeof(uint64_t));
      memset(matches + matches_size, 0, (new_size - matches_size) * sizeof(uint64_t));
      matches_size = new_size;
    }
    size_t num_found_matches =
        find_all_matches_h10(hasher, params->use_dictionary, ringbuffer, ringbuffer_mask, ringbuffer_break, pos,
                             max_length, max_distance, gap, params, &matches[cur_match_pos]);
    size_t cur_match_end = cur_match_pos + num_found_matches;
    num_matches[i] = (uint32_t)num_found_matches;
    if (num_found_matches > 0) {
      size_t mlen = match_length(matches[cur_match_end - 1]);
      if (mlen > 325) {
        size_t skip = mlen - 1;
        matches[cur_match_pos++] = matches[cur_match_end - 1];
        num_matches[i] = 1;
        h10_store_range(hasher, ringbuffer, ringbuffer_mask, pos + 1, ORC_MIN(pos + mlen, store_end));
        /* the reference zeroes num_matches[i + 1 .. i + 1 + skip) and panics if that runs past num_bytes */
        if (i + 1 + skip > num_bytes) {
          orc_reference_would_panic = 1;
          skip = num_bytes - i - 1;
        }
        memset(&num_matches[i + 1], 0, skip * sizeof(uint32_t));
        i += skip;
      } else {
        cur_match_pos = cur_match_end;
      }
    }
  }
  size_t orig_num_literals = *num_literals;
  size_t orig_last_insert_len = *last_insert_len;
  memcpy(orig_dist_cache, dist_cache, sizeof(orig_dist_cache));
  size_t orig_num_commands = *num_commands;
  ZopfliNode* nodes = (ZopfliNode*)malloc((num_bytes + 1) * sizeof(ZopfliNode));
  cost_model_init(&model, &params->dist, num_bytes);
  for (size_t i = 0; i < 2; ++i) {
    init_zopfli_nodes(nodes, num_bytes + 1);
    if (i == 0) {
      cost_model_set_from_literal_costs(&model, position, ringbuffer, ringbuffer_mask);
    } else {
      cost_model_set_from_commands(&model, position, ringbuffer, ringbuffer_mask, commands,
                                   *num_commands - orig_num_commands, orig_last_insert_len);
    }
    *num_commands = orig_num_commands;
    *num_literals = orig_num_literals;
    *last_insert_len = orig_last_insert_len;
    memcpy(dist_cache, orig_dist_cache, sizeof(orig_dist_cache));
    *num_commands += zopfli_iterate(num_bytes, position, ringbuffer, ringbuffer_mask, ringbuffer_break, params,
                                    max_backward_limit, gap, dist_cache, &model, num_matches, matches, nodes);
    zopfli_create_commands(num_bytes, position, max_backward_limit, nodes, dist_cache, last_insert_len, params, commands,
                           num_literals);
  }
  cost_model_cleanup(&model);
  free(nodes);
  free(matches);
  free(num_matches);
}
