// zopfli_device.h -- qualities 10 and 11 of the backward-reference stage: the H10 binary-tree hasher and the
// Zopfli-style shortest-path parse, as device code.  Shared by the gfx950 kernel (lz77_kernels.hip, k_zopfli_block) and the
// host emulation of the device seam (tests/emu, test infrastructure).
//
// What it replaces, per input block (encode.rs:2417-2453):
//   StitchToPreviousBlockH10                       hq.rs:254-300
//   extend_last_command                            encode.rs:360-400
//   BrotliCreateZopfliBackwardReferences (q10)     hq.rs:990-1041  -> BrotliZopfliComputeShortestPath :873-988
//   BrotliCreateHqZopfliBackwardReferences (q11)   hq.rs:1246-1448 -> ZopfliIterate :1162-1244, twice
// built from
//   StoreAndFindMatchesH10, Store, StoreRange      hash_to_binary_tree.rs:437-530, 283-318
//   FindAllMatchesH10                              hq.rs:301-412
//   BrotliFindAllStaticDictionaryMatches           static_dict.rs:309-1300 (here: a table of the affixes, not a cascade)
//   BrotliEstimateBitCostsForLiterals              literal_cost.rs:8-239
//   ZopfliCostModel, SetCost                       hq.rs:159-252, 1043-1160
//   StartPosQueue, EvaluateNode, UpdateNodes       hq.rs:414-855
//   ComputeShortestPathFromNodes, BrotliZopfliCreateCommands   hq.rs:857-871, 97-148
// All f32 arithmetic keeps the reference's operation order (floatX = f32; tolerance zero: byte identity of the stream is the
// test, pinned by the reference's own sizes 47 488 / 46 493 for alice29, src/bin/integration_tests.rs:401-449).
//
// How it runs: the matches of a block side by side, one lane per group of hash keys (the trees of
// different keys do not touch); the dynamic programme on one wavefront per stream with a position's candidates and copy lengths
// spread over the lanes; the f32 running sums and the sliding-window histograms in the reference's order on the way.  The text is flat (prefix + input); ring-buffer
// indices appear only where the reference's behaviour hangs on them (the custom-dictionary end, mod.rs:42-54).
#ifndef BROTLI_MI355X_ZOPFLI_DEVICE_H_
#define BROTLI_MI355X_ZOPFLI_DEVICE_H_

#include "lz77_chain.h"

namespace brotli_mi355x {

#if defined(BROTLI_HOST_EMU)
#define ZTICK() 0ull
#define ZDEV inline
#define ZCONST static const
#else
#define ZTICK() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define ZDEV __device__
#define ZCONST __device__ const
#endif

static constexpr uint32_t kZBucketBits = 17;          // hash_to_binary_tree.rs:106-112 (H10DefaultParams)
static constexpr uint32_t kZMaxTreeCompLength = 128;
static constexpr uint32_t kZMaxTreeSearchDepth = 64;
static constexpr uint32_t kZMaxMatches = 128;         // MAX_NUM_MATCHES_H10
static constexpr uint32_t kZInvalidMatch = 0x0fffffffu;
static constexpr uint32_t kZLongCopyQuickStep = 16384;
static constexpr float kZInfinity = 1.7e38f;

// read-only tables (device memory; the host arrays in the emulation)
struct ZopfliTables {
  const uint16_t* lut_buckets;    // kStaticDictionaryBuckets [32768]
  const uint32_t* lut_words;      // kStaticDictionaryWords   [31705]: len | transform << 8 | idx << 16, bit 7 of len = last of bucket
  const uint8_t* dict_data;
  const uint32_t* dict_offsets_by_length;
  const uint8_t* dict_size_bits_by_length;
  EntropyTables logs;
};

// ZopfliNode, hq.rs:27-66.  `u` is an enum in the reference (cost / next / shortcut); read as another variant it yields 0.
struct ZNode {
  uint32_t length;               // copy length | (copy length + 9 - length code) << 25
  uint32_t distance;
  uint32_t dcode_insert_length;  // insert length | short distance code << 27
  uint32_t u;                    // f32 bits / next / shortcut
  uint32_t tag;                  // 0 cost, 1 next, 2 shortcut
};

struct ZopfliParams {
  uint32_t quality;              // 10 or 11
  uint32_t lgwin;
  uint32_t max_backward_limit;   // (1 << lgwin) - 16
  uint32_t ring_mask;            // of the reference's ring buffer (custom-dictionary end rule only)
  uint32_t dict_break;           // ring_buffer_break, 0 = none
  uint32_t use_dictionary;
  uint32_t dist_max_distance;    // params.dist.max_distance
  uint32_t dist_alphabet_size;
  uint32_t ndirect, npostfix;
};

// per stream, device memory
struct ZopfliBuffers {
  uint32_t* buckets;       // [1 << 17]
  uint32_t* forest;        // [2 << lgwin]
  ZNode* nodes;            // [block_bytes + 1]
  float* literal_costs;    // [block_bytes + 2]
  float* cost_dist;        // [alphabet size]
  float* cost_cmd;         // [704]
  unsigned long long* matches;  // quality 10: [128]; quality 11: [128 * block_bytes] (the reference grows the array, this is its bound)
  uint32_t* num_matches;   // quality 11: [block_bytes]
  Command* tmp_cmds;       // quality 11: the commands of the first pass [block_bytes / 2 + 8]
  uint32_t* histo;         // [3 * 256 + 704 + 256] scratch
};

ZDEV float z_from_bits(uint32_t b) {
  float f;
  __builtin_memcpy(&f, &b, 4);
  return f;
}
ZDEV uint32_t z_to_bits(float f) {
  uint32_t b;
  __builtin_memcpy(&b, &f, 4);
  return b;
}
ZDEV float z_node_cost(const ZNode& n) { return n.tag == 0 ? z_from_bits(n.u) : 0.0f; }
ZDEV uint32_t z_node_next(const ZNode& n) { return n.tag == 1 ? n.u : 0u; }
ZDEV uint32_t z_node_shortcut(const ZNode& n) { return n.tag == 2 ? n.u : 0u; }
ZDEV void z_set_cost(ZNode& n, float c) {
  n.tag = 0;
  n.u = z_to_bits(c);
}
ZDEV uint32_t z_copy_length(const ZNode& n) { return n.length & 0x01ffffffu; }
ZDEV uint32_t z_insert_length(const ZNode& n) { return n.dcode_insert_length & 0x07ffffffu; }
ZDEV uint32_t z_length_code(const ZNode& n) { return z_copy_length(n) + 9u - (n.length >> 25); }
ZDEV uint32_t z_distance_code(const ZNode& n) {
  const uint32_t short_code = n.dcode_insert_length >> 27;
  return short_code == 0 ? n.distance + 16u - 1u : short_code - 1u;
}

// The nodes as the dynamic programme sees them: the array in device memory, and on the device a window of it in workgroup
// memory (the programme reads a few dozen node fields per position, one dependent round trip to memory each otherwise).
// Writes go to both, reads inside the window come from it; the window follows the position (keep()).
static constexpr uint32_t kZWin = 4096, kZWinBack = 1024, kZWinMinAhead = 1024;
struct ZNodeView {
  ZNode* g;
  ZNode* w;     // [kZWin], nullptr: no window (host emulation)
  uint32_t lo;  // the window holds the nodes [lo, lo + kZWin)
  const float* lc_g;  // the prefix sums of the literal costs (ZCostModel::literal_costs) ride along: read at the position and at
  float* lc_w;        // the start positions in the queue, a little way back
  ZDEV float lc(uint32_t i) const {
    const uint32_t o = i - lo;
    return (lc_w != nullptr && o < kZWin) ? lc_w[o] : lc_g[i];
  }
  ZDEV ZNode get(uint32_t i) const {
    const uint32_t o = i - lo;
    return (w != nullptr && o < kZWin) ? w[o] : g[i];
  }
  ZDEV void put(uint32_t i, const ZNode& n) {
    g[i] = n;
    const uint32_t o = i - lo;
    if (w != nullptr && o < kZWin) w[o] = n;
  }
  // called with the position the programme is at (uniform): moves the window when less than kZWinMinAhead nodes of it lie ahead
  ZDEV void keep(uint32_t pos, uint32_t count) {
    if (w == nullptr) return;
    if (pos >= lo && lo + kZWin - pos >= kZWinMinAhead) return;
    BR_SYNC();  // (everything written so far has reached the array)
    lo = pos > kZWinBack ? pos - kZWinBack : 0u;
    for (uint32_t o = (uint32_t)BR_LANE; o < kZWin; o += BR_NLANES)
      if (lo + o < count) {
        w[o] = g[lo + o];
        if (lc_w) lc_w[o] = lc_g[lo + o];
      }
    BR_SYNC();
  }
};

// FindMatchLengthWithLimit, static_dict.rs:125-132
ZDEV uint32_t z_match_len(const uint8_t* a, const uint8_t* b, uint32_t limit) {
  uint32_t i = 0;
  while (i + 8 <= limit) {
    const uint64_t x = br_load64(a + i) ^ br_load64(b + i);
    if (x != 0) return i + ((uint32_t)__builtin_ctzll(x) >> 3);
    i += 8;
  }
  while (i < limit && a[i] == b[i]) ++i;
  return i;
}
// fix_unbroken_len, mod.rs:42-54 (prev_ix is a ring-buffer index)
ZDEV uint32_t z_fix_unbroken(uint32_t len, uint32_t prev_ix_masked, uint32_t brk) {
  if (brk != 0 && prev_ix_masked < brk && prev_ix_masked + len > brk) return brk - prev_ix_masked;
  return len;
}
// FastLog2 / FastLog2f64 (util.rs:17-45): table below 256, else log2f of the value as f32
ZDEV float z_fast_log2(const EntropyTables& t, uint64_t v) { return v < 256 ? t.logs_8[v] : br_log2f((float)v); }

// ---- H10 ------------------------------------------------------------------------------------------------------------
struct ZH10 {
  uint32_t* buckets;
  uint32_t* forest;
  uint32_t window_mask;
  uint32_t invalid_pos;  // 0 - window_mask
  // Where the trees of different hash keys grow side by side (br_zopfli_matches_of_group) the nodes of the positions from
  // new_from on live in an array of their own: node storage is indexed by position mod window size, so the node of an in-block
  // position p shares its slot with the node of p - window, which a search at a position in front of p may still visit --
  // sequentially p comes later, side by side it may come first.  forest_new == forest, new_from = 0: one array (sequential use).
  uint32_t* forest_new;
  uint32_t new_from;
};
ZDEV uint32_t* z_h10_children(const ZH10& h, uint32_t pos) { return (pos >= h.new_from ? h.forest_new : h.forest) + 2 * (size_t)(pos & h.window_mask); }
ZDEV unsigned long long z_match(uint32_t distance, uint32_t length_and_code) { return (unsigned long long)distance | ((unsigned long long)length_and_code << 32); }
ZDEV uint32_t z_match_distance(unsigned long long m) { return (uint32_t)m; }
ZDEV uint32_t z_match_length(unsigned long long m) { return (uint32_t)(m >> 32) >> 5; }
ZDEV uint32_t z_match_length_code(unsigned long long m) {
  const uint32_t code = (uint32_t)(m >> 32) & 31u;
  return code != 0 ? code : z_match_length(m);
}

// StoreAndFindMatchesH10, hash_to_binary_tree.rs:437-530: walks down the tree of the position's hash key, re-roots it at
// cur_ix (if the position still has a full comparison length in front of it), and collects the matches that beat *best_len.
ZDEV uint32_t z_h10_store_and_find(const ZH10& h, const ZopfliParams& P, const uint8_t* data, uint32_t cur_ix, uint32_t max_length,
                                   uint64_t max_backward, uint32_t* best_len, unsigned long long* matches, uint32_t cap) {
  uint32_t found = 0;
  const uint32_t max_comp_len = max_length < kZMaxTreeCompLength ? max_length : kZMaxTreeCompLength;
  const bool reroot = max_length >= kZMaxTreeCompLength;
  const uint32_t key = (br_load32(data + cur_ix) * 0x1e35a7bdu) >> (32 - kZBucketBits);
  uint64_t prev_ix = h.buckets[key];
  uint32_t* node_left = z_h10_children(h, cur_ix);
  uint32_t* node_right = node_left + 1;
  uint32_t best_left = 0, best_right = 0;
  if (reroot) h.buckets[key] = cur_ix;
  for (uint32_t depth = kZMaxTreeSearchDepth;; --depth) {
    const uint64_t backward = (uint64_t)cur_ix - prev_ix;  // (usize arithmetic of the reference: wraps for the invalid position)
    if (backward == 0 || backward > max_backward || depth == 0) {
      if (reroot) {
        *node_left = h.invalid_pos;
        *node_right = h.invalid_pos;
      }
      break;
    }
    const uint32_t prev = (uint32_t)prev_ix;
    const uint32_t cur_len = best_left < best_right ? best_left : best_right;
    const uint32_t len = z_fix_unbroken(cur_len + z_match_len(data + cur_ix + cur_len, data + prev + cur_len, max_length - cur_len),
                                        prev & P.ring_mask, P.dict_break);
    if (found != cap && len > *best_len) {
      *best_len = len;
      matches[found++] = z_match((uint32_t)backward, len << 5);
    }
    uint32_t* prev_children = z_h10_children(h, prev);
    if (len >= max_comp_len) {
      if (reroot) {
        *node_left = prev_children[0];
        *node_right = prev_children[1];
      }
      break;
    }
    if (data[cur_ix + len] > data[prev + len]) {
      best_left = len;
      if (reroot) *node_left = prev;
      node_left = prev_children + 1;
      prev_ix = *node_left;
    } else {
      best_right = len;
      if (reroot) *node_right = prev;
      node_right = prev_children;
      prev_ix = *node_right;
    }
  }
  return found;
}
// Store, hash_to_binary_tree.rs:283-296
ZDEV void z_h10_store(const ZH10& h, const ZopfliParams& P, const uint8_t* data, uint32_t ix) {
  uint32_t best_len = 0;
  z_h10_store_and_find(h, P, data, ix, kZMaxTreeCompLength, (uint64_t)h.window_mask - 16 + 1, &best_len, nullptr, 0);
}
// StoreRange, hash_to_binary_tree.rs:297-318: every position of the last 63, every eighth in front of them if the range is long
ZDEV void z_h10_store_range(const ZH10& h, const ZopfliParams& P, const uint8_t* data, uint32_t ix_start, uint32_t ix_end) {
  uint32_t i = ix_start, j = ix_start;
  if (ix_start + 63 <= ix_end) i = ix_end - 63;
  if (ix_start + 512 <= i)
    for (; j < i; j += 8) z_h10_store(h, P, data, j);
  for (; i < ix_end; ++i) z_h10_store(h, P, data, i);
}
// StitchToPreviousBlockH10, hq.rs:254-300: the last 128 positions of the block in front are stored again, now that the
// bytes behind them are there
ZDEV void z_h10_stitch(const ZH10& h, const ZopfliParams& P, const uint8_t* data, uint32_t num_bytes, uint32_t position) {
  if (num_bytes >= 3 && position >= kZMaxTreeCompLength) {
    const uint32_t i_start = position - kZMaxTreeCompLength;
    const uint32_t i_end = position < i_start + num_bytes ? position : i_start + num_bytes;
    for (uint32_t i = i_start; i < i_end; ++i) {
      const uint32_t gap = position - i > 15 ? position - i : 15;
      uint32_t best_len = 0;
      z_h10_store_and_find(h, P, data, i, kZMaxTreeCompLength, (uint64_t)h.window_mask - gap, &best_len, nullptr, 0);
    }
  }
}

// ---- BrotliFindAllStaticDictionaryMatches ---------------------------------------------------------------------------
// matches[len] = min over the words that match `len` bytes of (distance-within-dictionary << 5 | length code); a word of
// length l with index id under transform t sits at id + t * (1 << size_bits[l]).  The reference spells the transforms out as
// a cascade of character tests; here they are rows of a table (affix bytes, transform id) -- the result per length is a
// minimum, so the order of the tests does not matter.
struct ZAffix {
  uint8_t n;       // bytes behind the word that must match
  uint8_t t;       // transform id
  char s[8];
};
ZDEV void z_add_match(uint32_t distance, uint32_t len, uint32_t len_code, uint32_t* matches) {
  const uint32_t m = (distance << 5) + len_code;
  if (m < matches[len]) matches[len] = m;
}
ZDEV bool z_affix_is(const uint8_t* s, const ZAffix& a) {
  for (uint32_t i = 0; i < a.n; ++i)
    if (s[i] != (uint8_t)a.s[i]) return false;
  return true;
}
// IsMatch, static_dict.rs:251-288: the word as it is (t 0), with its first letter in upper case (t 10), or all upper case
ZDEV bool z_word_matches(const ZopfliTables& T, uint32_t l, uint32_t t, uint32_t idx, const uint8_t* data, uint32_t max_length) {
  if (l > max_length) return false;
  const uint8_t* w = T.dict_data + T.dict_offsets_by_length[l] + l * idx;
  if (t == 0) return z_match_len(w, data, l) == l;
  if (t == 10) return w[0] >= 'a' && w[0] <= 'z' && (uint8_t)(w[0] ^ 32) == data[0] && z_match_len(w + 1, data + 1, l - 1) == l - 1;
  for (uint32_t i = 0; i < l; ++i) {
    const uint8_t c = (w[i] >= 'a' && w[i] <= 'z') ? (uint8_t)(w[i] ^ 32) : w[i];
    if (c != data[i]) return false;
  }
  return true;
}
// transforms "word + suffix" of the identity form (static_dict.rs:395-760)
ZCONST ZAffix kZPlainSuffix[] = {
    {1, 1, " "},      {3, 28, " a "},    {4, 46, " as "},   {4, 60, " at "},   {5, 10, " and "},  {4, 38, " by "},  {4, 16, " in "},
    {4, 47, " is "},  {5, 25, " for "},  {6, 37, " from "}, {4, 8, " of "},    {4, 45, " on "},   {5, 80, " not "}, {5, 5, " the "},
    {6, 29, " that "}, {4, 17, " to "},  {6, 35, " with "}, {1, 19, "\""},     {2, 21, "\">"},    {1, 20, "."},     {2, 31, ". "},
    {6, 43, ". The "}, {7, 75, ". This "}, {1, 76, ","},    {2, 14, ", "},     {1, 22, "\n"},     {2, 50, "\n\t"},  {1, 24, "]"},
    {1, 36, "'"},     {1, 51, ":"},      {1, 57, "("},      {2, 70, "=\""},    {2, 86, "='"},     {3, 84, "al "},   {3, 53, "ed "},
    {3, 82, "er "},   {4, 95, "est "},   {4, 90, "ful "},   {4, 92, "ive "},   {4, 100, "ize "},  {5, 93, "less "}, {3, 61, "ly "},
    {4, 106, "ous "}};
// suffixes of the upper-case forms: {first letter upper case, all upper case} (static_dict.rs:761-900)
ZCONST ZAffix kZUpperSuffix[][2] = {
    {{1, 4, " "}, {1, 68, " "}},       {{1, 66, "\""}, {1, 87, "\""}},    {{2, 69, "\">"}, {2, 97, "\">"}},  {{1, 79, "."}, {1, 101, "."}},
    {{2, 88, ". "}, {2, 114, ". "}},   {{1, 99, ","}, {1, 112, ","}},     {{2, 58, ", "}, {2, 107, ", "}},   {{1, 74, "'"}, {1, 94, "'"}},
    {{1, 78, "("}, {1, 113, "("}},     {{2, 104, "=\""}, {2, 105, "=\""}}, {{2, 108, "='"}, {2, 116, "='"}}};
ZCONST uint8_t kZOmitLastN[10] = {0, 12, 27, 23, 42, 63, 56, 48, 59, 64};
// " word" / ".word" followed by ... (static_dict.rs:901-1140)
ZCONST ZAffix kZSpacePlain[] = {{1, 2, " "}, {1, 89, "("}, {1, 103, ","}, {2, 33, ", "}, {1, 71, "."}, {2, 52, ". "}, {2, 81, "=\""}, {2, 98, "='"}};
ZCONST ZAffix kZDotPlain[] = {{1, 77, " "}, {1, 67, "("}};
ZCONST ZAffix kZSpaceUpper[][2] = {{{1, 15, " "}, {1, 83, " "}},   {{1, 109, ","}, {0, 0, ""}},        {{2, 65, ", "}, {2, 111, ", "}}, {{1, 96, "."}, {1, 115, "."}},
                                   {{2, 91, ". "}, {2, 117, ". "}}, {{2, 118, "=\""}, {2, 110, "=\""}}, {{2, 120, "='"}, {2, 119, "='"}}};
ZDEV bool z_find_all_dictionary_matches(const ZopfliTables& T, const uint8_t* data, uint32_t min_length, uint32_t max_length, uint32_t* matches) {
  bool any = false;
  // one bucket walk of the lookup table: calls `visit(l, t, idx, n)` for every word of the bucket of the four bytes at `at`
#define Z_FOR_EACH_WORD(at)                                                                                       \
  for (uint32_t off_ = T.lut_buckets[(br_load32(at) * 0x1e35a7bdu) >> (32 - 15)], end_ = off_ == 0; !end_;)    \
    if (const uint32_t packed_ = T.lut_words[off_++]; true)                                                       \
      if (const uint32_t l = packed_ & 0x1fu, t = (packed_ >> 8) & 0xffu, id = packed_ >> 16,                    \
          n = 1u << T.dict_size_bits_by_length[packed_ & 0x1fu];                                                  \
          (end_ = (packed_ & 0x80u) != 0), true)
  // ---- the word starts at data[0] (static_dict.rs:321-900)
  Z_FOR_EACH_WORD(data) {
    if (t == 0) {
      const uint8_t* w = T.dict_data + T.dict_offsets_by_length[l] + l * id;
      const uint32_t matchlen = z_match_len(w, data, l < max_length ? l : max_length);
      if (matchlen == l) {
        z_add_match(id, l, l, matches);
        any = true;
      }
      if (matchlen + 1 >= l) {  // all but the last byte: "omit last 1", and that followed by "ing "
        z_add_match(id + 12 * n, l - 1, l, matches);
        if (l + 2 < max_length && data[l - 1] == 'i' && data[l] == 'n' && data[l + 1] == 'g' && data[l + 2] == ' ') z_add_match(id + 49 * n, l + 3, l, matches);
        any = true;
      }
      // "omit last 2 .. 9"
      uint32_t minlen = min_length;
      if (l > 9 && l - 9 > minlen) minlen = l - 9;
      const uint32_t maxlen = matchlen < l - 2 ? matchlen : l - 2;
      for (uint32_t len = minlen; len <= maxlen; ++len) {
        z_add_match(id + (uint32_t)kZOmitLastN[l - len] * n, len, l, matches);
        any = true;
      }
      if (matchlen < l || l + 6 >= max_length) continue;
      for (const ZAffix& a : kZPlainSuffix)
        if (z_affix_is(data + l, a)) z_add_match(id + (uint32_t)a.t * n, l + a.n, l, matches);
    } else {
      const uint32_t caps = t != 10 ? 1u : 0u;
      if (!z_word_matches(T, l, t, id, data, max_length)) continue;
      z_add_match(id + (caps ? 44u : 9u) * n, l, l, matches);
      any = true;
      if (l + 1 >= max_length) continue;
      for (const auto& a : kZUpperSuffix)
        if (z_affix_is(data + l, a[caps])) z_add_match(id + (uint32_t)a[caps].t * n, l + a[caps].n, l, matches);
    }
  }
  // ---- " word" and ".word" (static_dict.rs:901-1140)
  if (max_length >= 5 && (data[0] == ' ' || data[0] == '.')) {
    const bool space = data[0] == ' ';
    Z_FOR_EACH_WORD(data + 1) {
      if (t == 0) {
        if (!z_word_matches(T, l, 0, id, data + 1, max_length - 1)) continue;
        z_add_match(id + (space ? 6u : 32u) * n, l + 1, l, matches);
        any = true;
        if (l + 2 >= max_length) continue;
        if (space) {
          for (const ZAffix& a : kZSpacePlain)
            if (z_affix_is(data + l + 1, a)) z_add_match(id + (uint32_t)a.t * n, l + 1 + a.n, l, matches);
        } else {
          for (const ZAffix& a : kZDotPlain)
            if (z_affix_is(data + l + 1, a)) z_add_match(id + (uint32_t)a.t * n, l + 1 + a.n, l, matches);
        }
      } else if (space) {
        const uint32_t caps = t != 10 ? 1u : 0u;
        if (!z_word_matches(T, l, t, id, data + 1, max_length - 1)) continue;
        z_add_match(id + (caps ? 85u : 30u) * n, l + 1, l, matches);
        any = true;
        if (l + 2 >= max_length) continue;
        for (const auto& a : kZSpaceUpper)
          if (a[caps].n != 0 && z_affix_is(data + l + 1, a[caps])) z_add_match(id + (uint32_t)a[caps].t * n, l + 1 + a[caps].n, l, matches);
      }
    }
  }
  // ---- "e word ", "s word ", ", word " and U+00A0 + word (static_dict.rs:1141-1230)
  if (max_length >= 6 && ((data[1] == ' ' && (data[0] == 'e' || data[0] == 's' || data[0] == ',')) || (data[0] == 0xc2 && data[1] == 0xa0))) {
    Z_FOR_EACH_WORD(data + 2) {
      if (t != 0 || !z_word_matches(T, l, 0, id, data + 2, max_length - 2)) continue;
      if (data[0] == 0xc2) {
        z_add_match(id + 102 * n, l + 2, l, matches);
        any = true;
      } else if (l + 2 < max_length && data[l + 2] == ' ') {
        z_add_match(id + (data[0] == 'e' ? 18u : (data[0] == 's' ? 7u : 13u)) * n, l + 3, l, matches);
        any = true;
      }
    }
  }
  // ---- " the word", ".com/word", " the word of ", " the word of the " (static_dict.rs:1231-1300)
  if (max_length >= 9 && ((data[0] == ' ' && data[1] == 't' && data[2] == 'h' && data[3] == 'e' && data[4] == ' ') ||
                          (data[0] == '.' && data[1] == 'c' && data[2] == 'o' && data[3] == 'm' && data[4] == '/'))) {
    Z_FOR_EACH_WORD(data + 5) {
      if (t != 0 || !z_word_matches(T, l, 0, id, data + 5, max_length - 5)) continue;
      z_add_match(id + (data[0] == ' ' ? 41u : 72u) * n, l + 5, l, matches);
      any = true;
      if (l + 5 < max_length) {
        const uint8_t* s = data + l + 5;
        if (data[0] == ' ' && l + 8 < max_length && s[0] == ' ' && s[1] == 'o' && s[2] == 'f' && s[3] == ' ') {
          z_add_match(id + 62 * n, l + 9, l, matches);
          if (l + 12 < max_length && s[4] == 't' && s[5] == 'h' && s[6] == 'e' && s[7] == ' ') z_add_match(id + 73 * n, l + 13, l, matches);
        }
      }
    }
  }
#undef Z_FOR_EACH_WORD
  return any;
}

// FindAllMatchesH10, hq.rs:301-412: short distances by direct comparison, then the tree, then the static dictionary
ZDEV uint32_t z_find_all_matches(const ZH10& h, const ZopfliParams& P, const ZopfliTables& T, const uint8_t* data, uint32_t cur_ix,
                                 uint32_t max_length, uint32_t max_backward, unsigned long long* matches, bool* went_into_tree = nullptr) {
  uint32_t found = 0;
  uint32_t best_len = 1;
  const uint32_t short_reach = P.quality != 11 ? 16u : 64u;
  const uint32_t stop = cur_ix < short_reach ? 0u : cur_ix - short_reach;
  for (uint32_t i = cur_ix - 1; i > stop && best_len <= 2 && cur_ix != 0; --i) {
    const uint32_t backward = cur_ix - i;
    if (backward > max_backward) break;
    if (data[cur_ix] == data[i] && data[cur_ix + 1] == data[i + 1]) {
      const uint32_t len = z_match_len(data + i, data + cur_ix, max_length);
      if (len > best_len) {
        best_len = len;
        matches[found++] = z_match(backward, len << 5);
      }
    }
  }
  if (went_into_tree) *went_into_tree = best_len < max_length;
  if (best_len < max_length) found += z_h10_store_and_find(h, P, data, cur_ix, max_length, max_backward, &best_len, matches + found, kZMaxMatches - found);
  if (P.use_dictionary) {
    uint32_t dict_matches[38];
    for (uint32_t i = 0; i <= 37; ++i) dict_matches[i] = kZInvalidMatch;
    const uint32_t minlen = best_len + 1 > 4 ? best_len + 1 : 4;
    if (z_find_all_dictionary_matches(T, data + cur_ix, minlen, max_length, dict_matches)) {
      const uint32_t maxlen = max_length < 37 ? max_length : 37;
      for (uint32_t l = minlen; l <= maxlen; ++l) {
        const uint32_t dict_id = dict_matches[l];
        if (dict_id < kZInvalidMatch) {
          const uint64_t distance = (uint64_t)max_backward + (dict_id >> 5) + 1;  // (gap = 0)
          if (distance <= P.dist_max_distance) {
            const uint32_t len_code = dict_id & 31u;
            matches[found++] = z_match((uint32_t)distance, (l << 5) | (l == len_code ? 0u : len_code));
          }
        }
      }
    }
  }
  return found;
}

// ---- BrotliEstimateBitCostsForLiterals, literal_cost.rs ---------------------------------------------------------------------
ZDEV uint32_t z_utf8_position(uint32_t last, uint32_t c, uint32_t clamp) {  // :8-18
  if (c < 128) return 0;
  if (c >= 192) return clamp < 1 ? clamp : 1;
  if (last < 0xe0) return 0;
  return clamp < 2 ? clamp : 2;
}
// BrotliIsMostlyUTF8, utf8_util.rs:3-62
ZDEV bool z_is_mostly_utf8(const uint8_t* data, uint32_t length, float min_fraction) {
  uint32_t size_utf8 = 0;
  for (uint32_t i = 0; i < length;) {
    const uint32_t size = length - i;
    const uint32_t b0 = data[i], b1 = size > 1 ? data[i + 1] : 0u, b2 = size > 2 ? data[i + 2] : 0u, b3 = size > 3 ? data[i + 3] : 0u;
    uint32_t read = 0;
    bool valid = true;
    if ((b0 & 0x80) == 0 && b0 > 0) {
      read = 1;
    } else if (size > 1 && (b0 & 0xe0) == 0xc0 && (b1 & 0xc0) == 0x80 && (((b0 & 0x1f) << 6) | (b1 & 0x3f)) > 0x7f) {
      read = 2;
    } else if (size > 2 && (b0 & 0xf0) == 0xe0 && (b1 & 0xc0) == 0x80 && (b2 & 0xc0) == 0x80 && (((b0 & 0x0f) << 12) | ((b1 & 0x3f) << 6) | (b2 & 0x3f)) > 0x7ff) {
      read = 3;
    } else {
      const uint32_t sym4 = ((b0 & 0x07) << 18) | ((b1 & 0x3f) << 12) | ((b2 & 0x3f) << 6) | (b3 & 0x3f);
      if (size > 3 && (b0 & 0xf8) == 0xf0 && (b1 & 0xc0) == 0x80 && (b2 & 0xc0) == 0x80 && (b3 & 0xc0) == 0x80 && sym4 > 0xffff && sym4 <= 0x10ffff) {
        read = 4;
      } else {
        read = 1;
        valid = false;  // (0x110000 | first byte)
      }
    }
    i += read;
    if (valid) size_utf8 += read;
  }
  return (float)size_utf8 > min_fraction * (float)length;
}
// cost[i] for the `len` bytes at data (literal_cost.rs:48-239); histo: 3 * 256 words of scratch
ZDEV void z_literal_costs(const ZopfliTables& T, const uint8_t* data, uint32_t len, uint32_t* histo, float* cost) {
  if (z_is_mostly_utf8(data, len, 0.75f)) {
    // DecideMultiByteStatsLevel, :20-46
    uint32_t counts[3] = {0, 0, 0};
    {
      uint32_t last_c = 0;
      for (uint32_t i = 0; i < len; ++i) {
        const uint32_t c = data[i];
        counts[z_utf8_position(last_c, c, 2)]++;
        last_c = c;
      }
    }
    uint32_t max_utf8 = 1;
    if (counts[2] < 500) max_utf8 = 1;
    if (counts[1] + counts[2] < 25) max_utf8 = 0;
    const uint32_t window_half = 495;
    const uint32_t in_window = window_half < len ? window_half : len;
    uint32_t in_window_utf8[3] = {0, 0, 0};
    for (uint32_t i = 0; i < 3 * 256; ++i) histo[i] = 0;
    {
      uint32_t last_c = 0, utf8_pos = 0;
      for (uint32_t i = 0; i < in_window; ++i) {
        const uint32_t c = data[i];
        histo[utf8_pos * 256 + c]++;
        in_window_utf8[utf8_pos]++;
        utf8_pos = z_utf8_position(last_c, c, max_utf8);
        last_c = c;
      }
    }
    for (uint32_t i = 0; i < len; ++i) {
      if (i >= window_half) {
        const uint32_t c = i < window_half + 1 ? 0u : data[i - window_half - 1];
        const uint32_t last_c = i < window_half + 2 ? 0u : data[i - window_half - 2];
        const uint32_t u = z_utf8_position(last_c, c, max_utf8);
        histo[u * 256 + data[i - window_half]]--;
        in_window_utf8[u]--;
      }
      if (i + window_half < len) {
        const uint32_t c = data[i + window_half - 1];
        const uint32_t last_c = data[i + window_half - 2];
        const uint32_t u = z_utf8_position(last_c, c, max_utf8);
        histo[u * 256 + data[i + window_half]]++;
        in_window_utf8[u]++;
      }
      const uint32_t c = i < 1 ? 0u : data[i - 1];
      const uint32_t last_c = i < 2 ? 0u : data[i - 2];
      const uint32_t u = z_utf8_position(last_c, c, max_utf8);
      uint32_t hv = histo[u * 256 + data[i]];
      if (hv == 0) hv = 1;
      double lit_cost = (double)z_fast_log2(T.logs, in_window_utf8[u]) - (double)z_fast_log2(T.logs, hv);
      lit_cost += 0.02905;
      if (lit_cost < 1.0) {
        lit_cost *= 0.5;
        lit_cost += 0.5;
      }
      if (i < 2000) lit_cost += (0.7 - (double)(2000 - i) / 2000.0 * 0.35);
      cost[i] = (float)lit_cost;
    }
  } else {
    const uint32_t window_half = 2000;
    uint32_t in_window = window_half < len ? window_half : len;
    for (uint32_t i = 0; i < 256; ++i) histo[i] = 0;
    for (uint32_t i = 0; i < in_window; ++i) histo[data[i]]++;
    for (uint32_t i = 0; i < len; ++i) {
      if (i >= window_half) {
        histo[data[i - window_half]]--;
        in_window--;
      }
      if (i + window_half < len) {
        histo[data[i + window_half]]++;
        in_window++;
      }
      uint32_t hv = histo[data[i]];
      if (hv == 0) hv = 1;
      double lit_cost = (double)z_fast_log2(T.logs, in_window) - (double)z_fast_log2(T.logs, hv);
      lit_cost += 0.029;
      if (lit_cost < 1.0) {
        lit_cost *= 0.5;
        lit_cost += 0.5;
      }
      cost[i] = (float)lit_cost;
    }
  }
}

// ---- ZopfliCostModel, hq.rs:159-252, 1043-1160 --------------------------------------------------------------------------------
struct ZCostModel {
  float* cost_cmd;        // [704]
  float* cost_dist;
  float* literal_costs;   // prefix sums, [num_bytes + 2]
  uint32_t distance_histogram_size;
  float min_cost_cmd;
  uint32_t num_bytes;
};
ZDEV float z_literal_cost_between(const ZCostModel& m, uint32_t from, uint32_t to) { return m.literal_costs[to] - m.literal_costs[from]; }
struct ZNodeView;
ZDEV float z_literal_cost_between(const ZNodeView& v, uint32_t from, uint32_t to);
// set_from_literal_costs, hq.rs:199-240 (the running sum carries its rounding error along, Kahan style, in this order)
ZDEV void z_model_from_literal_costs(ZCostModel& m, const ZopfliTables& T, const uint8_t* data, uint32_t* histo) {
  float* lc = m.literal_costs;
  z_literal_costs(T, data, m.num_bytes, histo, lc + 1);
  lc[0] = 0.0f;
  float carry = 0.0f;
  for (uint32_t i = 0; i < m.num_bytes; ++i) {
    carry = carry + lc[i + 1];
    lc[i + 1] = lc[i] + carry;
    carry -= lc[i + 1] - lc[i];
  }
  for (uint32_t i = 0; i < 704; ++i) m.cost_cmd[i] = z_fast_log2(T.logs, 11 + (uint64_t)i);
  for (uint32_t i = 0; i < m.distance_histogram_size; ++i) m.cost_dist[i] = z_fast_log2(T.logs, 20 + (uint64_t)i);
  m.min_cost_cmd = z_fast_log2(T.logs, 11);
}
// SetCost, hq.rs:1043-1071
ZDEV void z_set_cost_from_histogram(const ZopfliTables& T, const uint32_t* histogram, uint32_t size, bool literal, float* cost) {
  uint64_t sum = 0;
  for (uint32_t i = 0; i < size; ++i) sum += histogram[i];
  const float log2sum = z_fast_log2(T.logs, sum);
  uint64_t missing = sum;
  if (!literal)
    for (uint32_t i = 0; i < size; ++i)
      if (histogram[i] == 0) missing++;
  const float missing_cost = z_fast_log2(T.logs, missing) + 2.0f;
  for (uint32_t i = 0; i < size; ++i) {
    if (histogram[i] == 0) {
      cost[i] = missing_cost;
    } else {
      float c = log2sum - z_fast_log2(T.logs, histogram[i]);
      if (c < 1.0f) c = 1.0f;
      cost[i] = c;
    }
  }
}
// set_from_commands, hq.rs:1073-1160.  histo: 256 + 704 + 140 words + 256 floats of scratch.  Returns false where the reference
// indexes its 140-entry distance histogram out of bounds (it panics there).
ZDEV bool z_model_from_commands(ZCostModel& m, const ZopfliTables& T, const uint8_t* text, uint32_t position, const Command* cmds,
                                uint32_t num_commands, uint32_t last_insert_len, uint32_t* histo) {
  uint32_t* h_lit = histo;
  uint32_t* h_cmd = histo + 256;
  uint32_t* h_dist = histo + 256 + 704;
  float* cost_literal = (float*)(histo + 256 + 704 + 140);
  for (uint32_t i = 0; i < 256 + 704 + 140; ++i) histo[i] = 0;
  for (uint32_t i = 0; i < 256; ++i) cost_literal[i] = 0.0f;
  bool ok = true;
  uint32_t pos = position - last_insert_len;
  for (uint32_t i = 0; i < num_commands; ++i) {
    const uint32_t inslength = cmds[i].insert_len_;
    const uint32_t copylength = cmds[i].copy_len_ & 0x01ffffffu;
    const uint32_t distcode = cmds[i].dist_prefix_ & 0x03ffu;
    const uint32_t cmdcode = cmds[i].cmd_prefix_;
    h_cmd[cmdcode]++;
    if (cmdcode >= 128) {
      if (distcode >= 140) ok = false;
      else h_dist[distcode]++;
    }
    for (uint32_t j = 0; j < inslength; ++j) h_lit[text[pos + j]]++;
    pos += inslength + copylength;
  }
  z_set_cost_from_histogram(T, h_lit, 256, true, cost_literal);
  z_set_cost_from_histogram(T, h_cmd, 704, false, m.cost_cmd);
  z_set_cost_from_histogram(T, h_dist, m.distance_histogram_size < 140 ? m.distance_histogram_size : 140, false, m.cost_dist);
  float min_cost_cmd = kZInfinity;
  for (uint32_t i = 0; i < 704; ++i) min_cost_cmd = m.cost_cmd[i] < min_cost_cmd ? m.cost_cmd[i] : min_cost_cmd;
  m.min_cost_cmd = min_cost_cmd;
  float* lc = m.literal_costs;
  float carry = 0.0f;
  lc[0] = 0.0f;
  for (uint32_t i = 0; i < m.num_bytes; ++i) {
    carry += cost_literal[text[position + i]];
    lc[i + 1] = lc[i] + carry;
    carry -= lc[i + 1] - lc[i];
  }
  return ok;
}

ZDEV float z_literal_cost_between(const ZNodeView& v, uint32_t from, uint32_t to) { return v.lc(to) - v.lc(from); }

// ---- StartPosQueue, EvaluateNode, UpdateNodes (hq.rs:414-855) ------------------------------------------------------------------
struct ZPosData {
  uint32_t pos;
  int32_t distance_cache[4];
  float costdiff;
  float cost;
};
struct ZQueue {
  ZPosData q[8];
  uint32_t idx;
};
ZDEV uint32_t z_queue_size(const ZQueue& q) { return q.idx < 8 ? q.idx : 8; }
ZDEV void z_queue_push(ZQueue& q, const ZPosData& d) {  // keeps the entries ordered by costdiff (one bubble pass)
  uint32_t offset = ~q.idx & 7u;
  q.idx++;
  const uint32_t len = z_queue_size(q);
  q.q[offset] = d;
  for (uint32_t i = 1; i < len; ++i) {
    if (q.q[offset & 7].costdiff > q.q[(offset + 1) & 7].costdiff) {
      const ZPosData t = q.q[offset & 7];
      q.q[offset & 7] = q.q[(offset + 1) & 7];
      q.q[(offset + 1) & 7] = t;
    }
    ++offset;
  }
}
ZDEV const ZPosData& z_queue_at(const ZQueue& q, uint32_t k) { return q.q[(k - q.idx) & 7u]; }

// ComputeDistanceShortcut, hq.rs:427-452
ZDEV uint32_t z_distance_shortcut(uint32_t block_start, uint32_t pos, uint32_t max_backward, const ZNodeView& nodes) {
  const ZNode n = nodes.get(pos);
  const uint32_t clen = z_copy_length(n);
  const uint32_t ilen = z_insert_length(n);
  const uint32_t dist = n.distance;
  if (pos == 0) return 0;
  if ((uint64_t)dist + clen <= (uint64_t)block_start + pos && dist <= max_backward && z_distance_code(n) > 0) return pos;
  return z_node_shortcut(nodes.get(pos - clen - ilen));
}
// ComputeDistanceCache, hq.rs:461-499
ZDEV void z_distance_cache_at(uint32_t pos, const int32_t* starting, const ZNodeView& nodes, int32_t* out) {
  int idx = 0;
  uint32_t p = z_node_shortcut(nodes.get(pos));
  while (idx < 4 && p > 0) {
    const ZNode n = nodes.get(p);
    const uint32_t ilen = z_insert_length(n);
    const uint32_t clen = z_copy_length(n);
    out[idx++] = (int32_t)n.distance;
    p = z_node_shortcut(nodes.get(p - clen - ilen));
  }
  for (; idx < 4; ++idx) out[idx] = *starting++;
}
// EvaluateNode, hq.rs:524-560
ZDEV void z_evaluate_node(uint32_t block_start, uint32_t pos, uint32_t max_backward_limit, const int32_t* starting_dist_cache,
                          const ZCostModel& model, ZQueue& queue, ZNodeView& nodes) {
  ZNode here = nodes.get(pos);
  const float cost = z_node_cost(here);
  const uint32_t shortcut = z_distance_shortcut(block_start, pos, max_backward_limit, nodes);
  here.tag = 2;
  here.u = shortcut;
  nodes.put(pos, here);
  if (cost <= z_literal_cost_between(nodes, 0, pos)) {
    ZPosData d;
    d.pos = pos;
    d.cost = cost;
    d.costdiff = cost - z_literal_cost_between(nodes, 0, pos);
    z_distance_cache_at(pos, starting_dist_cache, nodes, d.distance_cache);
    z_queue_push(queue, d);
  }
}
// ComputeMinimumCopyLength, hq.rs:577-602
ZDEV uint32_t z_minimum_copy_length(float start_cost, const ZNodeView& nodes, uint32_t num_bytes, uint32_t pos) {
  float min_cost = start_cost;
  uint32_t len = 2, next_len_bucket = 4, next_len_offset = 10;
  while (pos + len <= num_bytes && z_node_cost(nodes.get(pos + len)) <= min_cost) {
    ++len;
    if (len == next_len_offset) {
      min_cost += 1.0f;
      next_len_offset += next_len_bucket;
      next_len_bucket *= 2;
    }
  }
  return len;
}
ZDEV uint32_t z_ins_extra(uint32_t code) {  // kInsExtra
  return code < 6 ? 0u : (code < 8 ? 1u : (code < 10 ? 2u : (code < 12 ? 3u : (code < 14 ? 4u : (code < 16 ? 5u : (code == 16 ? 6u : (code == 17 ? 7u : (code == 18 ? 8u : (code == 19 ? 9u : (code == 20 ? 10u : (code == 21 ? 12u : (code == 22 ? 14u : 24u))))))))))));
}
ZDEV uint32_t z_copy_extra(uint32_t code) {  // kCopyExtra
  return code < 8 ? 0u : (code < 10 ? 1u : (code < 12 ? 2u : (code < 14 ? 3u : (code < 16 ? 4u : (code < 18 ? 5u : (code == 18 ? 6u : (code == 19 ? 7u : (code == 20 ? 8u : (code == 21 ? 9u : (code == 22 ? 10u : 24u))))))))));
}
// PrefixEncodeCopyDistance, command.rs:134-173: the symbol with its number of extra bits in the top six bits
ZDEV uint32_t z_distance_symbol(uint32_t distance_code, uint32_t ndirect, uint32_t npostfix) {
  if (distance_code < 16 + ndirect) return distance_code;
  const uint64_t dist = (1ull << (npostfix + 2)) + ((uint64_t)distance_code - 16 - ndirect);
  const uint32_t bucket = (63u ^ (uint32_t)__builtin_clzll(dist)) - 1;
  const uint64_t postfix = dist & ((1u << npostfix) - 1);
  const uint64_t prefix = (dist >> bucket) & 1;
  const uint64_t nbits = bucket - npostfix;
  return (uint32_t)((nbits << 10) | (16 + ndirect + ((2 * (nbits - 1) + prefix) << npostfix) + postfix));
}
ZDEV void z_update_node(ZNodeView& nodes, uint32_t pos, uint32_t start_pos, uint32_t len, uint32_t len_code, uint32_t dist, uint32_t short_code, float cost) {
  ZNode next;
  next.length = len | ((len + 9u - len_code) << 25);
  next.distance = dist;
  next.dcode_insert_length = (pos - start_pos) | (short_code << 27);
  z_set_cost(next, cost);
  nodes.put(pos + len, next);
}
// UpdateNodes, hq.rs:644-829: the paths that reach `pos` are extended by every copy that starts there -- the 16 distance-cache
// codes of up to five start positions, then the matches of the position for the two best of them
ZDEV uint32_t z_update_nodes(const ZopfliParams& P, const uint8_t* text, uint32_t num_bytes, uint32_t block_start, uint32_t pos,
                             const int32_t* starting_dist_cache, uint32_t num_matches, const unsigned long long* matches, const ZCostModel& model,
                             ZQueue& queue, ZNodeView& nodes) {
  const uint32_t cur_ix = block_start + pos;
  const uint32_t max_distance = cur_ix < P.max_backward_limit ? cur_ix : P.max_backward_limit;
  const uint32_t max_len = num_bytes - pos;
  const uint32_t max_zlen = P.quality <= 10 ? 150u : 325u;
  uint32_t result = 0;
  z_evaluate_node(block_start, pos, P.max_backward_limit, starting_dist_cache, model, queue, nodes);
  uint32_t min_len;
  {
    const ZPosData& d = z_queue_at(queue, 0);
    const float min_cost = d.cost + model.min_cost_cmd + z_literal_cost_between(nodes, d.pos, pos);
    min_len = z_minimum_copy_length(min_cost, nodes, num_bytes, pos);
  }
  const uint32_t max_candidates = P.quality <= 10 ? 1u : 5u;
  const uint32_t kmax = max_candidates < z_queue_size(queue) ? max_candidates : z_queue_size(queue);
#if !BR_SCALAR
  // The wave's turn (the whole wave runs this function with identical scalar state).  First the match lengths of ALL distance-
  // cache candidates of the position -- sixteen codes for each of the (up to five) start positions -- in two goes: lanes
  // 16 * k + j for the first four start positions, lanes j for the fifth.  One wait for the text instead of one per start position.
  uint32_t cand_len[2] = {0, 0}, cand_back[2] = {0, 0};
  {
    const uint32_t lane = (uint32_t)BR_LANE;
    for (uint32_t go = 0; go < 2; ++go) {
      const uint32_t k = go == 0 ? lane >> 4 : 4u;
      const uint32_t j = lane & 15u;
      if (k < kmax && (go == 0 || lane < 16)) {
        const ZPosData& dk = z_queue_at(queue, k);
        const uint32_t idx = j < 4 ? j : (j < 10 ? 0u : 1u);
        const int32_t off = j < 4 ? 0 : (int32_t)(((j - 4) % 6) / 2 + 1) * (((j - 4) & 1) ? 1 : -1);
        const uint64_t backward = (uint64_t)(int64_t)(dk.distance_cache[idx] + off);
        const uint64_t prev64 = (uint64_t)cur_ix - backward;
        if (backward <= max_distance && prev64 < cur_ix) {
          const uint32_t prev_ix = (uint32_t)prev64;
          cand_len[go] = z_fix_unbroken(z_match_len(text + prev_ix, text + cur_ix, max_len), prev_ix & P.ring_mask, P.dict_break);
          cand_back[go] = (uint32_t)backward;
        }
      }
      if (kmax <= 4) break;
    }
  }
  // ... and the matches of the position, one per lane
  unsigned long long my_match = 0;
  if ((uint32_t)BR_LANE < num_matches) my_match = matches[BR_LANE];
#endif
  for (uint32_t k = 0; k < kmax; ++k) {
    const ZPosData& d = z_queue_at(queue, k);
    const uint32_t start = d.pos;
    const uint32_t inscode = br_insert_length_code(pos - start);
    const float start_costdiff = d.costdiff;
    const float base_cost = start_costdiff + (float)z_ins_extra(inscode) + z_literal_cost_between(nodes, 0, pos);
    uint32_t best_len = min_len - 1;
#if BR_SCALAR
    for (uint32_t j = 0; j < 16; ++j) {
      if (best_len >= max_len) break;
      // kDistanceCacheIndex / kDistanceCacheOffset (mod.rs:653-655)
      const uint32_t idx = j < 4 ? j : (j < 10 ? 0u : 1u);
      const int32_t off = j < 4 ? 0 : (int32_t)(((j - 4) % 6) / 2 + 1) * (((j - 4) & 1) ? 1 : -1);
      const uint64_t backward = (uint64_t)(int64_t)(d.distance_cache[idx] + off);
      // (the reference's ring buffer: no candidate is looked at across its end, hq.rs:721-736)
      if ((cur_ix & P.ring_mask) + best_len > P.ring_mask) break;
      if (backward > max_distance) continue;
      const uint64_t prev64 = (uint64_t)cur_ix - backward;
      if (prev64 >= cur_ix) continue;
      const uint32_t prev_ix = (uint32_t)prev64;
      if ((prev_ix & P.ring_mask) + best_len > P.ring_mask) continue;
      if (text[cur_ix + best_len] != text[prev_ix + best_len]) continue;
      const uint32_t len = z_fix_unbroken(z_match_len(text + prev_ix, text + cur_ix, max_len), prev_ix & P.ring_mask, P.dict_break);
      const float dist_cost = base_cost + model.cost_dist[j];
      for (uint32_t l = best_len + 1; l <= len; ++l) {
        const uint32_t copycode = br_copy_length_code(l);
        const uint32_t cmdcode = br_combine_length_codes(inscode, copycode, j == 0);
        const float cost = (cmdcode < 128 ? base_cost : dist_cost) + (float)z_copy_extra(copycode) + model.cost_cmd[cmdcode];
        if (cost < z_node_cost(nodes.get(pos + l))) {
          z_update_node(nodes, pos, start, l, l, (uint32_t)backward, j + 1, cost);
          result = result > l ? result : l;
        }
        best_len = l;
      }
    }
#else
    {
      // candidate by candidate in the reference's order (only those that reach beyond min_len - 1 at all), the lengths it adds
      // beyond best_len on as many lanes -- different lengths are different nodes.  (The reference's look at the byte behind
      // best_len only saves it a comparison: a candidate that fails it has no length beyond best_len to offer.)
      const uint32_t lane = (uint32_t)BR_LANE;
      const uint32_t my_len = k < 4 ? cand_len[0] : cand_len[1], my_backward = k < 4 ? cand_back[0] : cand_back[1];
      const uint32_t group = k < 4 ? 16u * k : 0u;  // first lane of this start position's candidates
      unsigned long long todo = (__ballot(my_len > best_len) >> group) & 0xffffull;
      while (todo != 0) {
        if (best_len >= max_len) break;
        const uint32_t j = (uint32_t)__ffsll((long long)todo) - 1u;
        todo &= todo - 1ull;
        const uint32_t len = (uint32_t)__shfl((int)my_len, (int)(group + j), 64);
        const uint32_t backward = (uint32_t)__shfl((int)my_backward, (int)(group + j), 64);
        // (the reference's ring buffer: no candidate is looked at across its end, hq.rs:721-736; these two hang on best_len as
        // it stands when the candidate's turn comes)
        if ((cur_ix & P.ring_mask) + best_len > P.ring_mask) break;
        if (((cur_ix - backward) & P.ring_mask) + best_len > P.ring_mask) continue;
        if (len <= best_len) continue;
        const float dist_cost = base_cost + model.cost_dist[j];
        for (uint32_t first = best_len + 1; first <= len; first += 64) {
          const uint32_t l = first + lane;
          bool updated = false;
          if (l <= len) {
            const uint32_t copycode = br_copy_length_code(l);
            const uint32_t cmdcode = br_combine_length_codes(inscode, copycode, j == 0);
            const float cost = (cmdcode < 128 ? base_cost : dist_cost) + (float)z_copy_extra(copycode) + model.cost_cmd[cmdcode];
            if (cost < z_node_cost(nodes.get(pos + l))) {
              z_update_node(nodes, pos, start, l, l, backward, j + 1, cost);
              updated = true;
            }
          }
          const unsigned long long um = __ballot(updated);
          if (um != 0) {
            const uint32_t top = first + 63u - (uint32_t)__builtin_clzll(um);
            result = result > top ? result : top;
          }
        }
        best_len = len;
      }
      BR_SYNC();  // (the nodes written by single lanes are read by the others from here on)
    }
#endif
    if (k >= 2) continue;
    uint32_t len = min_len;
    for (uint32_t j = 0; j < num_matches; ++j) {
#if BR_SCALAR
      const unsigned long long match = matches[j];
#else
      const unsigned long long match = j < 64 ? (unsigned long long)__shfl((long long)my_match, (int)j, 64) : matches[j];
#endif
      const uint32_t dist = z_match_distance(match);
      const bool is_dictionary_match = dist > max_distance;
      const uint32_t dist_symbol = z_distance_symbol(dist + 16 - 1, P.ndirect, P.npostfix);
      const float dist_cost = base_cost + (float)(dist_symbol >> 10) + model.cost_dist[dist_symbol & 0x03ffu];
      const uint32_t max_match_len = z_match_length(match);
      if (len < max_match_len && (is_dictionary_match || max_match_len > max_zlen)) len = max_match_len;
#if BR_SCALAR
      for (; len <= max_match_len; ++len) {
        const uint32_t len_code = is_dictionary_match ? z_match_length_code(match) : len;
        const uint32_t copycode = br_copy_length_code(len_code);
        const uint32_t cmdcode = br_combine_length_codes(inscode, copycode, false);
        const float cost = dist_cost + (float)z_copy_extra(copycode) + model.cost_cmd[cmdcode];
        const ZNode there = nodes.get(pos + len);
        if (there.tag == 0 && cost < z_from_bits(there.u)) {
          z_update_node(nodes, pos, start, len, len_code, dist, 0, cost);
          result = result > len ? result : len;
        }
      }
#else
      // the lengths len .. max_match_len of this match, one per lane
      for (uint32_t first = len; first <= max_match_len; first += 64) {
        const uint32_t l = first + (uint32_t)BR_LANE;
        bool updated = false;
        if (l <= max_match_len) {
          const uint32_t len_code = is_dictionary_match ? z_match_length_code(match) : l;
          const uint32_t copycode = br_copy_length_code(len_code);
          const uint32_t cmdcode = br_combine_length_codes(inscode, copycode, false);
          const float cost = dist_cost + (float)z_copy_extra(copycode) + model.cost_cmd[cmdcode];
          const ZNode there = nodes.get(pos + l);
          if (there.tag == 0 && cost < z_from_bits(there.u)) {
            z_update_node(nodes, pos, start, l, len_code, dist, 0, cost);
            updated = true;
          }
        }
        const unsigned long long um = __ballot(updated);
        if (um != 0) {
          const uint32_t top = first + 63u - (uint32_t)__builtin_clzll(um);
          result = result > top ? result : top;
        }
      }
      if (len <= max_match_len) len = max_match_len + 1;
      BR_SYNC();
#endif
    }
  }
  return result;
}
// ComputeShortestPathFromNodes, hq.rs:857-871
ZDEV uint32_t z_shortest_path_from_nodes(uint32_t num_bytes, ZNode* nodes) {
  uint32_t index = num_bytes, num_commands = 0;
  while (z_insert_length(nodes[index]) == 0 && nodes[index].length == 1) --index;
  nodes[index].tag = 1;
  nodes[index].u = 0xffffffffu;
  while (index != 0) {
    const uint32_t len = z_copy_length(nodes[index]) + z_insert_length(nodes[index]);
    index -= len;
    nodes[index].tag = 1;
    nodes[index].u = len;
    ++num_commands;
  }
  return num_commands;
}
ZDEV void z_init_nodes(ZNode* nodes, uint32_t count) {  // (spread over the lanes of the wave)
  for (uint32_t i = (uint32_t)BR_LANE; i < count; i += BR_NLANES) {
    nodes[i].length = 1;
    nodes[i].distance = 0;
    nodes[i].dcode_insert_length = 0;
    z_set_cost(nodes[i], kZInfinity);
  }
  BR_SYNC();
}
// BrotliZopfliCreateCommands, hq.rs:97-148.  The commands come out complete (prefix codes included): the cost model of the
// second quality-11 pass reads them.  Returns the number of commands; *pending = bytes behind the last copy.
ZDEV uint32_t z_create_commands(const ZopfliParams& P, uint32_t num_bytes, uint32_t block_start, const ZNode* nodes, int32_t* dist_cache,
                                uint32_t last_insert_len, Command* commands, Command* raw, uint32_t* num_literals, uint32_t* pending,
                                uint32_t* last_dist_code, uint32_t* last_copy_len) {
  uint32_t pos = 0, count = 0;
  uint32_t offset = z_node_next(nodes[0]);
  for (uint32_t i = 0; offset != 0xffffffffu; ++i) {
    const ZNode& next = nodes[pos + offset];
    const uint32_t copy_length = z_copy_length(next);
    uint32_t insert_length = z_insert_length(next);
    pos += insert_length;
    offset = z_node_next(next);
    if (i == 0) insert_length += last_insert_len;
    const uint32_t distance = next.distance;
    const uint32_t len_code = z_length_code(next);
    const uint32_t max_distance = block_start + pos < P.max_backward_limit ? block_start + pos : P.max_backward_limit;
    const bool is_dictionary = distance > max_distance;
    const uint32_t dist_code = z_distance_code(next);
    // (raw: as the gather pass wants them -- without prefix codes, and the first one with its LOCAL literals only: the
    // resolver adds what was pending at the entry of the block)
    if (raw) raw[count] = br_raw_command(i == 0 ? insert_length - last_insert_len : insert_length, copy_length, len_code, dist_code);
    if (commands) commands[count] = br_make_command(P.ndirect, P.npostfix, insert_length, copy_length, len_code, dist_code);
    ++count;
    *last_dist_code = dist_code;
    *last_copy_len = copy_length;
    if (!is_dictionary && dist_code > 0) {
      dist_cache[3] = dist_cache[2];
      dist_cache[2] = dist_cache[1];
      dist_cache[1] = dist_cache[0];
      dist_cache[0] = (int32_t)distance;
    }
    *num_literals += insert_length;
    pos += copy_length;
  }
  *pending = num_bytes - pos;
  return count;
}

// workgroup memory of the parse kernel (all null in the host emulation)
struct ZFast {
  ZNode* window = nullptr;      // [kZWin]
  float* lc_window = nullptr;   // [kZWin]
  float* cost_cmd = nullptr;    // [704]
  float* cost_dist = nullptr;   // [>= distance alphabet]
  struct ZQueue* queue = nullptr;
};

// ---- one input block -------------------------------------------------------------------------------------------------------------
// BrotliZopfliComputeShortestPath, hq.rs:873-988 (quality 10): matches and node updates position by position
ZDEV void z_shortest_path_q10(const ZH10& h, const ZopfliParams& P, const ZopfliTables& T, const ZopfliBuffers& B, const uint8_t* text,
                              uint32_t num_bytes, uint32_t position, const int32_t* dist_cache, const ZFast& fast) {
  ZNodeView nodes;
  nodes.g = B.nodes;
  nodes.w = fast.window;
  nodes.lo = 0;
  nodes.lc_g = B.literal_costs;
  nodes.lc_w = fast.lc_window;
  const uint32_t store_end = num_bytes >= kZMaxTreeCompLength ? position + num_bytes - kZMaxTreeCompLength + 1 : position;
  {
    ZNode first = B.nodes[0];
    first.length = 0;
    z_set_cost(first, 0.0f);
    B.nodes[0] = first;
  }
  BR_SYNC();
  ZCostModel model;
  model.cost_cmd = B.cost_cmd;
  model.cost_dist = B.cost_dist;
  model.literal_costs = B.literal_costs;
  model.distance_histogram_size = P.dist_alphabet_size < 544 ? P.dist_alphabet_size : 544;
  model.num_bytes = num_bytes;
  z_model_from_literal_costs(model, T, text + position, B.histo);
  BR_SYNC();
  nodes.lo = 0xffffffffu;  // (nothing in the window yet)
  nodes.keep(0, num_bytes + 1);
  ZQueue local_queue;
  ZQueue& queue = fast.queue ? *fast.queue : local_queue;
  queue.idx = 0;
  for (uint32_t k = 0; k < 8; ++k) {
    queue.q[k].pos = 0;
    queue.q[k].costdiff = queue.q[k].cost = 0.0f;
    for (int c = 0; c < 4; ++c) queue.q[k].distance_cache[c] = 0;
  }
  unsigned long long* matches = B.matches;
  for (uint32_t i = 0; i + 3 < num_bytes; ++i) {
    nodes.keep(i, num_bytes + 1);
    const uint32_t pos = position + i;
    const uint32_t max_distance = pos < P.max_backward_limit ? pos : P.max_backward_limit;
    uint32_t num_matches = z_find_all_matches(h, P, T, text, pos, num_bytes - i, max_distance, matches);
    const uint32_t max_zlen = 150;
    if (num_matches > 0 && z_match_length(matches[num_matches - 1]) > max_zlen) {
      matches[0] = matches[num_matches - 1];
      num_matches = 1;
    }
    uint32_t skip = z_update_nodes(P, text, num_bytes, position, i, dist_cache, num_matches, matches, model, queue, nodes);
    if (skip < kZLongCopyQuickStep) skip = 0;
    if (num_matches == 1 && z_match_length(matches[0]) > max_zlen) skip = z_match_length(matches[0]) > skip ? z_match_length(matches[0]) : skip;
    if (skip > 1) {
      z_h10_store_range(h, P, text, pos + 1, pos + skip < store_end ? pos + skip : store_end);
      --skip;
      while (skip != 0) {
        ++i;
        if (i + 3 >= num_bytes) break;
        nodes.keep(i, num_bytes + 1);
        z_evaluate_node(position, i, P.max_backward_limit, dist_cache, model, queue, nodes);
        --skip;
      }
    }
  }
  BR_SYNC();
}
// ZopfliIterate, hq.rs:1162-1244: the node updates over matches collected beforehand.  Quality 11 as the reference runs it
// (matches packed one behind the other, stride 0); with stride = 128 the matches of position i sit at matches + 128 * i
// (br_zopfli_matches_of_group) -- then also quality 10, whose reference interleaves matching and node updates
// (BrotliZopfliComputeShortestPath, hq.rs:873-988): the same sequence of UpdateNodes calls as long as no position is skipped;
// returns false where one would be (a copy beyond the quick step: the reference stores the skipped positions differently).
ZDEV bool z_iterate(const ZopfliParams& P, const ZopfliBuffers& B, const uint8_t* text, uint32_t num_bytes, uint32_t position,
                    const int32_t* dist_cache, const ZCostModel& model, uint32_t stride, const ZFast& fast) {
  ZNodeView nodes;
  nodes.g = B.nodes;
  nodes.w = fast.window;
  nodes.lo = 0xffffffffu;  // (nothing in the window yet)
  nodes.lc_g = model.literal_costs;
  nodes.lc_w = fast.lc_window;
  const uint32_t max_zlen = P.quality <= 10 ? 150u : 325u;
  {
    ZNode first = B.nodes[0];
    first.length = 0;
    z_set_cost(first, 0.0f);
    B.nodes[0] = first;
  }
  BR_SYNC();
  nodes.keep(0, num_bytes + 1);
  ZQueue local_queue;
  ZQueue& queue = fast.queue ? *fast.queue : local_queue;
  queue.idx = 0;
  for (uint32_t k = 0; k < 8; ++k) {
    queue.q[k].pos = 0;
    queue.q[k].costdiff = queue.q[k].cost = 0.0f;
    for (int c = 0; c < 4; ++c) queue.q[k].distance_cache[c] = 0;
  }
  size_t cur_match_pos = 0;
  for (uint32_t i = 0; i + 3 < num_bytes; ++i) {
    nodes.keep(i, num_bytes + 1);
    const unsigned long long* m = stride ? B.matches + (size_t)stride * i : B.matches + cur_match_pos;
    const uint32_t nm = B.num_matches[i];
    uint32_t skip = z_update_nodes(P, text, num_bytes, position, i, dist_cache, nm, m, model, queue, nodes);
    if (skip < kZLongCopyQuickStep) skip = 0;
    cur_match_pos += nm;
    if (nm == 1 && z_match_length(m[0]) > max_zlen) skip = z_match_length(m[0]) > skip ? z_match_length(m[0]) : skip;
    if (skip > 1) {
      if (P.quality <= 10) {
        BR_SYNC();
        return false;
      }
      --skip;
      while (skip != 0) {
        ++i;
        if (i + 3 >= num_bytes) break;
        nodes.keep(i, num_bytes + 1);
        z_evaluate_node(position, i, P.max_backward_limit, dist_cache, model, queue, nodes);
        cur_match_pos += B.num_matches[i];
        --skip;
      }
    }
  }
  BR_SYNC();
  return true;
}

// ---- one input block ---------------------------------------------------------------------------------------------------------
static constexpr uint32_t kZopfliOk = 0, kZopfliReferencePanics = 1, kZopfliRedo = 2;
// what travels between the kernels of a block (device memory)
struct ZBlockCtl {
  uint32_t position, num_bytes;  // what is left of the block behind extend_last_command
  uint32_t ext_len;
  uint32_t event;                // some position found a match beyond the Zopfli length: the reference skips positions there
  uint32_t pad[4];
  unsigned long long ticks[4];   // cycles of the parse kernel by phase (BROTLI_MI355X_DEBUG): cost model, programme, commands, matches (sequential)
};
ZDEV ZH10 z_hasher_of(const ZopfliParams& P, const ZopfliBuffers& B) {
  ZH10 h;
  h.buckets = B.buckets;
  h.forest = B.forest;
  h.window_mask = (1u << P.lgwin) - 1u;
  h.invalid_pos = 0u - h.window_mask;
  h.forest_new = B.forest;
  h.new_from = 0;
  return h;
}
ZDEV ZCostModel z_model_of(const ZopfliParams& P, const ZopfliBuffers& B, uint32_t num_bytes) {
  ZCostModel model;
  model.cost_cmd = B.cost_cmd;
  model.cost_dist = B.cost_dist;
  model.literal_costs = B.literal_costs;
  model.distance_histogram_size = P.dist_alphabet_size < 544 ? P.dist_alphabet_size : 544;
  model.min_cost_cmd = 0.0f;
  model.num_bytes = num_bytes;
  return model;
}
// StitchToPreviousBlockH10 + extend_last_command (encode.rs:2417-2437): what comes in front of the parse of a block
ZDEV void br_zopfli_begin(const ZopfliParams& P, const ZopfliBuffers& B, const uint8_t* text, const Segment& seg, const SegEntry& entry, ZBlockCtl* ctl) {
  const ZH10 h = z_hasher_of(P, B);
  uint32_t position = seg.blk_start;
  uint32_t num_bytes = seg.blk_end - seg.blk_start;
  z_h10_stitch(h, P, text, num_bytes, position);
  uint32_t ext_len = 0;
  if (entry.ext_allowed) {  // (the resolver has checked everything but the bytes, encode.rs:360-400)
    const uint32_t d = (uint32_t)entry.cache[0];
    while (num_bytes != 0 && text[position] == text[position - d]) {
      ++ext_len;
      ++position;
      --num_bytes;
    }
  }
  ctl->position = position;
  ctl->num_bytes = num_bytes;
  ctl->ext_len = ext_len;
  ctl->event = 0;
}
// The matches of the positions of one group of hash keys -- the positions by_key[lo .. hi) of the block, in ascending order --
// as FindAllMatchesH10 finds them when every position of the block goes through it in turn: B.matches + 128 * i, B.num_matches[i],
// i = position in the block.  The trees of different keys do not touch (node storage apart, see ZH10), so the groups of a block
// run side by side; what hangs on the order ACROSS keys -- positions skipped behind a match beyond the Zopfli length, stored
// through StoreRange instead -- is reported (ctl->event) and the block is parsed again sequentially.
ZDEV void br_zopfli_matches_of_group(const ZopfliParams& P, const ZopfliTables& T, const ZopfliBuffers& B, uint32_t* forest_new, uint8_t* rerooted,
                                     const uint8_t* text, const uint32_t* by_key, uint32_t lo, uint32_t hi, ZBlockCtl* ctl) {
  ZH10 h = z_hasher_of(P, B);
  const uint32_t position = ctl->position, num_bytes = ctl->num_bytes;
  h.forest_new = forest_new;
  h.new_from = position;
  const uint32_t max_zlen = P.quality <= 10 ? 150u : 325u;
  // first slot of the group whose position is >= position
  uint32_t a = lo, b = hi;
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if (by_key[mid] < position) a = mid + 1; else b = mid;
  }
  for (uint32_t idx = a; idx < hi; ++idx) {
    const uint32_t pos = by_key[idx];
    const uint32_t i = pos - position;
    if (i + 3 >= num_bytes) break;
    const uint32_t max_distance = pos < P.max_backward_limit ? pos : P.max_backward_limit;
    const uint32_t max_length = num_bytes - i;
    unsigned long long* m = B.matches + (size_t)128 * i;
    bool tree = false;
    const uint32_t found = z_find_all_matches(h, P, T, text, pos, max_length, max_distance, m, &tree);
    B.num_matches[i] = found;
    rerooted[i] = (tree && max_length >= kZMaxTreeCompLength) ? 1 : 0;
    if (found > 0 && z_match_length(m[found - 1]) > max_zlen) ctl->event = 1;
  }
}
// the nodes that the positions of the block got in forest_new move to their slots of the one array (after all groups are through)
ZDEV void br_zopfli_merge_node(const ZopfliParams& P, const ZopfliBuffers& B, const uint32_t* forest_new, const uint8_t* rerooted, const ZBlockCtl* ctl, uint32_t i) {
  if (i >= ctl->num_bytes || !rerooted[i]) return;
  const size_t slot = 2 * (size_t)((ctl->position + i) & ((1u << P.lgwin) - 1u));
  B.forest[slot] = forest_new[slot];
  B.forest[slot + 1] = forest_new[slot + 1];
}

// The parse of the block behind br_zopfli_begin: commands into the block's slab (raw: the gather pass finishes them) and the exit
// record for the host resolver (Lz77Stage::Resolve).  precomputed: the matches are there (br_zopfli_matches_of_group); otherwise
// the block is walked the reference's way, matching and tree updates position by position.  Returns kZopfliRedo when the
// precomputed matches do not hold (nothing has been written then; the caller restores the trees and comes again).
ZDEV uint32_t br_zopfli_parse(const ZopfliParams& P, const ZopfliTables& T, const ZopfliBuffers& B, const uint8_t* text, const Segment& seg,
                              const SegEntry& entry, ZBlockCtl* ctl, bool precomputed, Command* slab, SegExit* exit_out, const ZFast& fast = ZFast()) {
  const ZH10 h = z_hasher_of(P, B);
  ZopfliBuffers Bf = B;  // (the cost tables of the model in workgroup memory where there is some)
  if (fast.cost_cmd) Bf.cost_cmd = fast.cost_cmd;
  if (fast.cost_dist) Bf.cost_dist = fast.cost_dist;
  unsigned long long t_model = 0, t_dp = 0, t_cmd = 0, t_match = 0, t0;
  uint32_t status = kZopfliOk;
  const uint32_t position = ctl->position, num_bytes = ctl->num_bytes;
  if (precomputed && ctl->event) return kZopfliRedo;
  int32_t dist_cache[4];
  for (int i = 0; i < 4; ++i) dist_cache[i] = entry.cache[i];
  uint32_t n_cmds = 0, n_lits = 0, pending = num_bytes, last_dist_code = 0xffffffffu, last_copy_len = 0;
  if (num_bytes != 0) {
    if (P.quality <= 10) {
      z_init_nodes(B.nodes, num_bytes + 1);
      if (precomputed) {
        ZCostModel model = z_model_of(P, Bf, num_bytes);
        t0 = ZTICK();
        z_model_from_literal_costs(model, T, text + position, B.histo);
        t_model += ZTICK() - t0;
        for (uint32_t i = num_bytes >= 3 ? num_bytes - 3 : 0; i < num_bytes; ++i) B.num_matches[i] = 0;
        t0 = ZTICK();
        if (!z_iterate(P, B, text, num_bytes, position, dist_cache, model, 128, fast)) return kZopfliRedo;
        t_dp += ZTICK() - t0;
      } else {
        t0 = ZTICK();
        z_shortest_path_q10(h, P, T, Bf, text, num_bytes, position, dist_cache, fast);
        t_dp += ZTICK() - t0;
      }
      t0 = ZTICK();
      z_shortest_path_from_nodes(num_bytes, B.nodes);
      n_cmds = z_create_commands(P, num_bytes, position, B.nodes, dist_cache, entry.insert_len, nullptr, slab, &n_lits, &pending, &last_dist_code, &last_copy_len);
      t_cmd += ZTICK() - t0;
    } else {
      // BrotliCreateHqZopfliBackwardReferences, hq.rs:1246-1448: all matches first ...
      if (!precomputed) {
        t0 = ZTICK();
        const uint32_t store_end = num_bytes >= kZMaxTreeCompLength ? position + num_bytes - kZMaxTreeCompLength + 1 : position;
        size_t cur_match_pos = 0;
        for (uint32_t i = 0; i < num_bytes; ++i) B.num_matches[i] = 0;
        for (uint32_t i = 0; i + 3 < num_bytes; ++i) {
          const uint32_t pos = position + i;
          const uint32_t max_distance = pos < P.max_backward_limit ? pos : P.max_backward_limit;
          const uint32_t found = z_find_all_matches(h, P, T, text, pos, num_bytes - i, max_distance, B.matches + cur_match_pos);
          const size_t cur_match_end = cur_match_pos + found;
          B.num_matches[i] = found;
          if (found > 0) {
            const uint32_t mlen = z_match_length(B.matches[cur_match_end - 1]);
            if (mlen > 325) {
              uint32_t skip = mlen - 1;
              B.matches[cur_match_pos++] = B.matches[cur_match_end - 1];
              B.num_matches[i] = 1;
              z_h10_store_range(h, P, text, pos + 1, pos + mlen < store_end ? pos + mlen : store_end);
              if ((uint64_t)i + 1 + skip > num_bytes) {  // the reference clears num_matches[i + 1 .. i + 1 + skip) and panics past the end
                status = kZopfliReferencePanics;
                skip = num_bytes - i - 1;
              }
              for (uint32_t k = 0; k < skip; ++k) B.num_matches[i + 1 + k] = 0;
              i += skip;
            } else {
              cur_match_pos = cur_match_end;
            }
          }
        }
        t_match += ZTICK() - t0;
      } else {
        for (uint32_t i = num_bytes >= 3 ? num_bytes - 3 : 0; i < num_bytes; ++i) B.num_matches[i] = 0;
      }
      // ... then two passes of the dynamic programme: literal-cost model, then the model of the first pass's commands
      ZCostModel model = z_model_of(P, Bf, num_bytes);
      int32_t orig_cache[4];
      for (int i = 0; i < 4; ++i) orig_cache[i] = dist_cache[i];
      for (uint32_t pass = 0; pass < 2; ++pass) {
        z_init_nodes(B.nodes, num_bytes + 1);
        t0 = ZTICK();
        if (pass == 0) {
          z_model_from_literal_costs(model, T, text + position, B.histo);
        } else {
          if (!z_model_from_commands(model, T, text, position, B.tmp_cmds, n_cmds, entry.insert_len, B.histo)) status = kZopfliReferencePanics;
        }
        t_model += ZTICK() - t0;
        for (int i = 0; i < 4; ++i) dist_cache[i] = orig_cache[i];
        n_lits = 0;
        t0 = ZTICK();
        z_iterate(P, B, text, num_bytes, position, dist_cache, model, precomputed ? 128u : 0u, fast);
        t_dp += ZTICK() - t0;
        t0 = ZTICK();
        z_shortest_path_from_nodes(num_bytes, B.nodes);
        n_cmds = z_create_commands(P, num_bytes, position, B.nodes, dist_cache, entry.insert_len, B.tmp_cmds, slab, &n_lits, &pending, &last_dist_code, &last_copy_len);
        t_cmd += ZTICK() - t0;
      }
    }
  }
  if (n_cmds != 0) n_lits -= entry.insert_len;
  SegExit x;
  x.pos = seg.blk_end;
  x.apply = 0;
  for (int i = 0; i < 4; ++i) x.cache[i] = dist_cache[i];
  x.insert_len = pending;
  x.n_cmds = n_cmds;
  x.n_lits = n_lits;
  x.ext_len = ctl->ext_len;
  x.dict_lookups = x.dict_matches = 0;
  x.last_dist_code = last_dist_code;
  x.bad_commands = status;  // (where the reference panics the product refuses, like for the copies it cannot encode)
  x.n_searches = seg.blk_end - seg.blk_start;
  x.last_copy_len = last_copy_len;
  x.dict_mode = 0;
  x.dict_maxdef = 0;
  x.n_pushes = 4;
  x.tail_kind = kHeadNone;
  x.tail_base = x.tail_p1 = 0;
  x.n_pushes_all = 4;
  x.dict_entry_lookups = x.dict_entry_matches = 0;
  *exit_out = x;
  ctl->ticks[0] = t_model;
  ctl->ticks[1] = t_dp;
  ctl->ticks[2] = t_cmd;
  ctl->ticks[3] = t_match;
  return status;
}

}  // namespace brotli_mi355x
#endif
