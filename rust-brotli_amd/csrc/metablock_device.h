// metablock_device.h -- device functions of the meta-block stage (shared by the HIP kernels and, compiled
// with BROTLI_HOST_EMU, by the CPU emulation used in tests).
//
// Restates for single device threads / workgroups:
//   src/enc/bit_cost.rs:13-42, util.rs:17-25           f32 entropy (sequential sums, exact order)
//   src/enc/entropy_encode.rs                          Huffman tree, RLE smoothing, canonical codes
//   src/enc/brotli_bit_stream.rs:742-911,1272-1858     bit writer, tree / block-switch / context-map storage
#ifndef BROTLI_MI355X_METABLOCK_DEVICE_H_
#define BROTLI_MI355X_METABLOCK_DEVICE_H_

#include "lz77_chain.h"
#include "metablock_types.h"

namespace brotli_mi355x {

// ---------------------------------------------------------------------------------------------- bit sink
// Sequential bit writer into zero-initialised 64-bit words (single writer).  Same LSB-first layout as
// BrotliWriteBits (brotli_bit_stream.rs:742-757).
struct BitSink {
  uint64_t* words;
  uint64_t pos;
  BR_DEV void put(uint32_t n_bits, uint64_t bits) {
    if (n_bits == 0) return;
    const uint32_t sh = (uint32_t)(pos & 63u);
    uint64_t* w = words + (pos >> 6);
    w[0] |= bits << sh;
    if (sh + n_bits > 64) w[1] |= bits >> (64 - sh);
    pos += n_bits;
  }
};

// BR_CODES_PROFILE (experiment builds only): cycles per phase of the Huffman jobs, summed in g_codes_prof[phase] and
// the slowest job's in g_codes_prof[16 + phase]; printed by mb_build_codes
#if defined(BR_CODES_PROFILE) && !defined(BROTLI_HOST_EMU)
extern __device__ unsigned long long g_codes_prof[32];
struct CodesPhaseClock {
  unsigned long long last;
  BR_DEV CodesPhaseClock() : last(__builtin_readcyclecounter()) {}
  BR_DEV void mark(int phase) {
    const unsigned long long now = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) {
      atomicAdd(&g_codes_prof[phase], now - last);
      atomicMax(&g_codes_prof[16 + phase], now - last);
    }
    last = __builtin_readcyclecounter();
  }
};
#define BR_PHASE_CLOCK() CodesPhaseClock br_phase_clock
#define BR_PHASE(k) br_phase_clock.mark(k)
#else
#define BR_PHASE_CLOCK()
#define BR_PHASE(k)
#endif

// ---------------------------------------------------------------------------------------------- Huffman
struct HuffmanTree {
  uint32_t total_count_;
  int16_t index_left_;
  int16_t index_right_or_value_;
};

// BrotliSetDepth, entropy_encode.rs:27-56
BR_DEV bool br_set_depth(int p0, HuffmanTree* pool, uint8_t* depth, int max_depth) {
  int stack[16];
  int level = 0;
  int p = p0;
  stack[0] = -1;
  for (;;) {
    if (pool[p].index_left_ >= 0) {
      level++;
      if (level > max_depth) return false;
      stack[level] = pool[p].index_right_or_value_;
      p = pool[p].index_left_;
      continue;
    } else {
      depth[pool[p].index_right_or_value_] = (uint8_t)level;
    }
    while (level >= 0 && stack[level] == -1) level--;
    if (level < 0) return true;
    p = stack[level];
    stack[level] = -1;
  }
}

BR_DEV bool br_sort_cmp(const HuffmanTree& v0, const HuffmanTree& v1) {  // entropy_encode.rs:61-69
  if (v0.total_count_ != v1.total_count_) return v0.total_count_ < v1.total_count_;
  return v0.index_right_or_value_ > v1.index_right_or_value_;
}

// SortHuffmanTreeItems, entropy_encode.rs:71-116 (insertion sort below 13 items, else shell sort)
// coop_tmp != nullptr (device only): all 64 lanes of the wavefront are executing this call with identical state; the order
// is a strict total order (count, then symbol), so any sorting method gives the reference's result -- here a rank sort:
// every lane counts, for its items, how many items precede them.
BR_DEV void br_sort_huffman_tree_items(HuffmanTree* items, uint32_t n, HuffmanTree* coop_tmp) {
#if !defined(BROTLI_HOST_EMU)
  if (coop_tmp != nullptr && n >= 13) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t e = lane; e < n; e += 64) coop_tmp[e] = items[e];
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e = lane; e < n; e += 64) {
      const HuffmanTree mine = coop_tmp[e];
      uint32_t rank = 0;
      for (uint32_t f = 0; f < n; ++f) rank += br_sort_cmp(coop_tmp[f], mine) ? 1u : 0u;
      items[rank] = mine;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    return;
  }
#endif
  (void)coop_tmp;
  const uint32_t gaps[6] = {132, 57, 23, 10, 4, 1};
  if (n < 13) {
    for (uint32_t i = 1; i < n; ++i) {
      HuffmanTree tmp = items[i];
      uint32_t k = i;
      uint32_t j = i - 1;
      while (br_sort_cmp(tmp, items[j])) {
        items[k] = items[j];
        k = j;
        if (j-- == 0) break;
      }
      items[k] = tmp;
    }
  } else {
    for (int g = n < 57 ? 2 : 0; g < 6; ++g) {
      const uint32_t gap = gaps[g];
      for (uint32_t i = gap; i < n; ++i) {
        uint32_t j = i;
        HuffmanTree tmp = items[i];
        for (; j >= gap && br_sort_cmp(tmp, items[j - gap]); j -= gap) items[j] = items[j - gap];
        items[j] = tmp;
      }
    }
  }
}

// BrotliCreateHuffmanTree, entropy_encode.rs:133-210.  tree must hold 2*length+1 nodes.
BR_DEV void br_create_huffman_tree(const uint32_t* data, uint32_t length, int tree_limit, HuffmanTree* tree, uint8_t* depth,
                                   HuffmanTree* coop_tmp = nullptr) {
  HuffmanTree sentinel;
  sentinel.total_count_ = 0xffffffffu;
  sentinel.index_left_ = -1;
  sentinel.index_right_or_value_ = -1;
#if !defined(BROTLI_HOST_EMU)
  // cooperative build: the sort's scratch is free again when the nodes are merged and holds the parent links then
  uint16_t* parent = (uint16_t*)coop_tmp;
  static_assert(sizeof(HuffmanTree) * 704 >= sizeof(uint16_t) * (2 * 704 + 2), "parent links fit the sort scratch");
#endif
  BR_PHASE_CLOCK();
  for (uint32_t count_limit = 1;; count_limit *= 2) {
    uint32_t n = 0;
    BR_PHASE(15);
#if !defined(BROTLI_HOST_EMU)
    if (coop_tmp != nullptr) {
      // the leaves in descending symbol order: lane l of a chunk takes the l-th symbol from its top, the leaf's slot is the
      // number of used symbols above it
      const uint32_t lane = threadIdx.x & 63u;
      for (uint32_t top = length; top != 0; top = top > 64 ? top - 64 : 0) {
        const bool valid = lane < top;
        const uint32_t i = valid ? top - 1 - lane : 0u;
        const uint32_t v = valid ? data[i] : 0u;
        const unsigned long long m = __ballot(v != 0);
        if (v != 0) {
          HuffmanTree leaf;
          leaf.total_count_ = v > count_limit ? v : count_limit;
          leaf.index_left_ = -1;
          leaf.index_right_or_value_ = (int16_t)i;
          tree[n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = leaf;
        }
        n += (uint32_t)__popcll(m);
      }
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    } else
#endif
    for (uint32_t i = length; i != 0;) {
      --i;
      if (data[i] != 0) {
        tree[n].total_count_ = data[i] > count_limit ? data[i] : count_limit;
        tree[n].index_left_ = -1;
        tree[n].index_right_or_value_ = (int16_t)i;
        n++;
      }
    }
    if (n == 1) {
      depth[tree[0].index_right_or_value_] = 1;
      break;
    }
    BR_PHASE(0);
    br_sort_huffman_tree_items(tree, n, coop_tmp);
    BR_PHASE(1);
    tree[n] = sentinel;
    tree[n + 1] = sentinel;
    uint32_t i = 0, j = n + 1;
    for (uint32_t k = n - 1; k != 0; --k) {
      uint32_t left, right;
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
      const uint32_t j_end = 2 * n - k;
      tree[j_end].total_count_ = tree[left].total_count_ + tree[right].total_count_;
      tree[j_end].index_left_ = (int16_t)left;
      tree[j_end].index_right_or_value_ = (int16_t)right;
      tree[j_end + 1] = sentinel;
#if !defined(BROTLI_HOST_EMU)
      if (parent != nullptr) parent[left] = parent[right] = (uint16_t)j_end;
#endif
    }
    BR_PHASE(2);
    bool fits;
#if !defined(BROTLI_HOST_EMU)
    if (parent != nullptr) {
      // BrotliSetDepth from the leaves: every lane climbs from its leaves to the root (node 2n-1) and gives up beyond
      // tree_limit levels -- the same depths as the traversal from the root, and the same verdict
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      const uint32_t lane = threadIdx.x & 63u, root = 2 * n - 1;
      bool too_deep = false;
      for (uint32_t e = lane; e < n; e += 64) {
        uint32_t p = e, d = 0;
        while (p != root && d <= (uint32_t)tree_limit) {
          p = parent[p];
          d++;
        }
        if (p != root || d > (uint32_t)tree_limit) {
          too_deep = true;
        } else {
          depth[tree[e].index_right_or_value_] = (uint8_t)d;
        }
      }
      fits = __ballot(too_deep) == 0ull;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    } else
#endif
    fits = br_set_depth((int)(2 * n - 1), tree, depth, tree_limit);
    BR_PHASE(3);
    if (fits) break;
  }
}

// BrotliOptimizeHuffmanCountsForRle, entropy_encode.rs:211-345.  good_for_rle: 704 bytes of scratch.
// (a / b for operands that mostly fit 32 bits: the 64-bit division is a long software routine on the device)
BR_DEV uint64_t br_div_u64(uint64_t a, uint64_t b) {
  if (((a | b) >> 32) == 0) return (uint64_t)((uint32_t)a / (uint32_t)b);
  return a / b;
}

#if !defined(BROTLI_HOST_EMU)
// The two passes of BrotliOptimizeHuffmanCountsForRle behind its early exits (entropy_encode.rs:268-345), by all 64 lanes of a
// wavefront in lock step (round 6: the 704-symbol command code spent 0.17 ms here, one dependent LDS round trip and a division per
// symbol).  (1) good_for_rle -- runs of >= 5 zeros / >= 7 equal non-zero counts -- from the start and the end of every symbol's run
// (two scans across the lanes instead of a walk).  (2) The smoothing walk itself is sequential (stride, sum and limit hang on
// everything in front), but what it READS lies at or behind its position and is never one of its own writes (those go in front of it):
// the counts and marks are held 64 to a register across the lanes and fetched with v_readlane at the (uniform) position, so a step
// is arithmetic only; the writes of a collapsed stride go to LDS, a lane per symbol.
BR_DEV void br_optimize_counts_tail_coop(uint32_t length, uint32_t* counts, uint8_t* good_for_rle) {
  const uint32_t lane = threadIdx.x & 63u;
  length = (uint32_t)__builtin_amdgcn_readfirstlane((int)length);
  const uint32_t chunks = (length + 63u) / 64u;  // <= 11
  // ---- (1) run start of every symbol (max-scan forward), run end (min-scan backward), the marks
  uint32_t rs[11];
  {
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t c = 0; c < 11; ++c) {
      rs[c] = 0;
      if (c >= chunks) continue;
      const uint32_t i = c * 64u + lane;
      const uint32_t v = i < length ? counts[i] : 0u;
      const uint32_t prev = (i != 0 && i < length) ? counts[i - 1] : ~v;
      uint32_t x = (i < length && (i == 0 || v != prev)) ? i : 0u;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, off, 64);
        if ((int)lane >= off) x = x > y ? x : y;
      }
      x = x > carry ? x : carry;
      rs[c] = x;
      carry = (uint32_t)__shfl((int)x, 63, 64);
    }
  }
  uint32_t gv[11];  // the marks, 64 to a register
  {
    uint32_t carry = 0xffffffffu;
#pragma unroll
    for (int c = 10; c >= 0; --c) {
      gv[c] = 0;
      if ((uint32_t)c >= chunks) continue;
      const uint32_t i = (uint32_t)c * 64u + lane;
      const uint32_t v = i < length ? counts[i] : 0u;
      const bool is_end = i < length && (i + 1 == length || counts[i + 1] != v);
      uint32_t x = is_end ? i : 0xffffffffu;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = (uint32_t)__shfl_down((int)x, off, 64);
        if ((int)lane + off < 64) x = x < y ? x : y;
      }
      x = x < carry ? x : carry;
      carry = (uint32_t)__shfl((int)x, 0, 64);
      const uint32_t len = i < length ? x - rs[c] + 1u : 0u;
      gv[c] = (i < length && ((v == 0 && len >= 5) || (v != 0 && len >= 7))) ? 1u : 0u;
    }
  }
  // ---- (2) the walk
  uint32_t cv[11];
#pragma unroll
  for (uint32_t c = 0; c < 11; ++c) {
    const uint32_t i = c * 64u + lane;
    cv[c] = (c < chunks && i < length) ? counts[i] : 0u;
  }
  const uint64_t streak_limit = 1240;
  uint64_t stride = 0;
  uint64_t limit = 0;
  uint64_t sum = 0;
  uint32_t good_prev = 0;
  // one chunk of 64 positions: its counts and marks in `cur` / `good`, the counts of the chunk behind it in `nxt` (the walk looks two
  // positions ahead); the position inside the chunk is uniform, so every fetch is one v_readlane with no branch around it
  auto walk_chunk = [&](uint32_t base, uint32_t cur, uint32_t nxt, uint32_t good) {
    if (base == 0) {
      limit = (uint64_t)((uint32_t)(256u * ((uint32_t)__builtin_amdgcn_readlane((int)cur, 0) + (uint32_t)__builtin_amdgcn_readlane((int)cur, 1) +
                                            (uint32_t)__builtin_amdgcn_readlane((int)cur, 2))) / 3u + 420u);
    }
    for (uint32_t l = 0; l < 64u && base + l <= length; ++l) {
      const uint32_t i = base + l;
      const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)l);    // (0 behind the last symbol)
      const uint32_t gi = (uint32_t)__builtin_amdgcn_readlane((int)good, (int)l);
      if (i == length || gi != 0 || (i != 0 && good_prev != 0) || (uint64_t)(uint32_t)(256u * ci) - limit + streak_limit >= 2 * streak_limit) {
        if (stride >= 4 || (stride >= 3 && sum == 0)) {
          uint64_t count = br_div_u64(sum + stride / 2, stride);
          if (count == 0) count = 1;
          if (sum == 0) count = 0;
          for (uint32_t k = lane; k < (uint32_t)stride; k += 64u) counts[i - k - 1] = (uint32_t)count;
        }
        stride = 0;
        sum = 0;
        if (i + 2 < length) {
          const uint32_t c1 = l + 1u < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)((l + 1u) & 63u)) : (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)((l + 1u) & 63u));
          const uint32_t c2 = l + 2u < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)((l + 2u) & 63u)) : (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)((l + 2u) & 63u));
          limit = (uint64_t)((uint32_t)(256u * (ci + c1 + c2)) / 3u + 420u);
        } else if (i < length) {
          limit = (uint64_t)(uint32_t)(256u * ci);
        } else {
          limit = 0;
        }
      }
      stride++;
      if (i != length) {
        sum += ci;
        if (stride >= 4) limit = br_div_u64(256 * sum + stride / 2, stride);
        if (stride == 4) limit += 120;
      }
      good_prev = gi;
    }
  };
#pragma unroll
  for (uint32_t c = 0; c < 11; ++c) {
    if (c * 64u > length) break;
    walk_chunk(c * 64u, cv[c], c + 1 < 11 ? cv[c + 1] : 0u, gv[c]);
  }
  if (length == 704u) {  // (the closing step i == length of a full alphabet lies behind the last chunk)
    walk_chunk(704u, 0u, 0u, 0u);
  }
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}
#endif

// coop (device): called by all 64 lanes of a wavefront in lock step on the same buffers; the counting passes are shared out
BR_DEV void br_optimize_huffman_counts_for_rle(uint32_t length, uint32_t* counts, uint8_t* good_for_rle, bool coop = false) {
  uint32_t nonzero_count = 0;
  const uint64_t streak_limit = 1240;
  uint32_t nonzeros = 0;
  uint32_t smallest_nonzero = 1u << 30;
#if !defined(BROTLI_HOST_EMU)
  if (coop) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t last = 0, smallest = 1u << 30;
    for (uint32_t base = 0; base < length; base += 64) {
      const uint32_t i = base + lane;
      const uint32_t v = i < length ? counts[i] : 0u;
      const unsigned long long m = __ballot(v != 0);
      nonzero_count += (uint32_t)__popcll(m);
      if (m != 0) last = base + 64u - (uint32_t)__builtin_clzll(m);  // one past the last used symbol
      if (v != 0 && v < smallest) smallest = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)smallest, off, 64);
      smallest = o < smallest ? o : smallest;
    }
    if (nonzero_count < 16) return;
    length = last;
    nonzeros = nonzero_count;  // (the symbols trimmed off the end were unused)
    smallest_nonzero = smallest;
  } else
#endif
  {
    for (uint32_t i = 0; i < length; ++i)
      if (counts[i] != 0) nonzero_count++;
    if (nonzero_count < 16) return;
    while (length != 0 && counts[length - 1] == 0) length--;
    if (length == 0) return;
    for (uint32_t i = 0; i < length; ++i) {
      if (counts[i] != 0) {
        nonzeros++;
        if (smallest_nonzero > counts[i]) smallest_nonzero = counts[i];
      }
    }
  }
  {
    if (nonzeros < 5) return;
    if (smallest_nonzero < 4) {
      const uint32_t zeros = length - nonzeros;
      if (zeros < 6) {
        for (uint32_t i = 1; i + 1 < length; ++i)
          if (counts[i - 1] != 0 && counts[i] == 0 && counts[i + 1] != 0) counts[i] = 1;
      }
    }
    if (nonzeros < 28) return;
  }
#if !defined(BROTLI_HOST_EMU) && !defined(BR_NO_COOP_RLE_TAIL)
  if (coop) {
    br_optimize_counts_tail_coop(length, counts, good_for_rle);
    return;
  }
#endif
  for (uint32_t i = 0; i < 704; ++i) good_for_rle[i] = 0;
  {
    uint32_t symbol = counts[0];
    uint32_t step = 0;
    for (uint32_t i = 0; i <= length; ++i) {
      if (i == length || counts[i] != symbol) {
        if ((symbol == 0 && step >= 5) || (symbol != 0 && step >= 7)) {
          for (uint32_t k = 0; k < step; ++k) good_for_rle[i - k - 1] = 1;
        }
        step = 1;
        if (i != length) symbol = counts[i];
      } else {
        step++;
      }
    }
  }
  uint64_t stride = 0;
  uint64_t limit = (uint64_t)((uint32_t)(256u * (counts[0] + counts[1] + counts[2])) / 3u + 420u);
  uint64_t sum = 0;
  for (uint32_t i = 0; i <= length; ++i) {
    if (i == length || good_for_rle[i] != 0 || (i != 0 && good_for_rle[i - 1] != 0) ||
        (uint64_t)(uint32_t)(256u * counts[i]) - limit + streak_limit >= 2 * streak_limit) {
      if (stride >= 4 || (stride >= 3 && sum == 0)) {
        uint64_t count = br_div_u64(sum + stride / 2, stride);
        if (count == 0) count = 1;
        if (sum == 0) count = 0;
        for (uint64_t k = 0; k < stride; ++k) counts[i - k - 1] = (uint32_t)count;
      }
      stride = 0;
      sum = 0;
      if (i + 2 < length) {
        limit = (uint64_t)((uint32_t)(256u * (counts[i] + counts[i + 1] + counts[i + 2])) / 3u + 420u);
      } else if (i < length) {
        limit = (uint64_t)(uint32_t)(256u * counts[i]);
      } else {
        limit = 0;
      }
    }
    stride++;
    if (i != length) {
      sum += counts[i];
      if (stride >= 4) limit = br_div_u64(256 * sum + stride / 2, stride);
      if (stride == 4) limit += 120;
    }
  }
}

BR_DEV void br_reverse_u8(uint8_t* v, uint32_t start, uint32_t end) {
  end--;
  while (start < end) {
    const uint8_t t = v[start];
    v[start] = v[end];
    v[end] = t;
    start++;
    end--;
  }
}

// BrotliWriteHuffmanTree + helpers, entropy_encode.rs:347-525.  tree / extra: 704 bytes each.
BR_DEV void br_write_huffman_tree(const uint8_t* depth, uint32_t length, uint32_t* tree_size, uint8_t* tree, uint8_t* extra) {
  uint8_t previous_value = 8;
  bool use_rle_for_non_zero = false, use_rle_for_zero = false;
  uint32_t new_length = length;
  for (uint32_t i = 0; i < length; ++i) {
    if (depth[length - i - 1] == 0) {
      new_length--;
    } else {
      break;
    }
  }
  if (length > 50) {  // decide_over_rle_use
    uint32_t total_reps_zero = 0, total_reps_non_zero = 0, count_reps_zero = 1, count_reps_non_zero = 1;
    for (uint32_t i = 0; i < new_length;) {
      const uint8_t value = depth[i];
      uint32_t reps = 1;
      for (uint32_t k = i + 1; k < new_length && depth[k] == value; ++k) reps++;
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
    use_rle_for_non_zero = total_reps_non_zero > count_reps_non_zero * 2;
    use_rle_for_zero = total_reps_zero > count_reps_zero * 2;
  }
  uint32_t n = *tree_size;
  for (uint32_t i = 0; i < new_length;) {
    const uint8_t value = depth[i];
    uint32_t reps = 1;
    if ((value != 0 && use_rle_for_non_zero) || (value == 0 && use_rle_for_zero)) {
      for (uint32_t k = i + 1; k < new_length && depth[k] == value; ++k) reps++;
    }
    uint32_t repetitions = reps;
    if (value == 0) {  // BrotliWriteHuffmanTreeRepetitionsZeros
      if (repetitions == 11) {
        tree[n] = 0;
        extra[n] = 0;
        n++;
        repetitions--;
      }
      if (repetitions < 3) {
        for (uint32_t r = 0; r < repetitions; ++r) {
          tree[n] = 0;
          extra[n] = 0;
          n++;
        }
      } else {
        const uint32_t start = n;
        repetitions -= 3;
        for (;;) {
          tree[n] = 17;
          extra[n] = (uint8_t)(repetitions & 0x7);
          n++;
          repetitions >>= 3;
          if (repetitions == 0) break;
          repetitions--;
        }
        br_reverse_u8(tree, start, n);
        br_reverse_u8(extra, start, n);
      }
    } else {  // BrotliWriteHuffmanTreeRepetitions
      if (previous_value != value) {
        tree[n] = value;
        extra[n] = 0;
        n++;
        repetitions--;
      }
      if (repetitions == 7) {
        tree[n] = value;
        extra[n] = 0;
        n++;
        repetitions--;
      }
      if (repetitions < 3) {
        for (uint32_t r = 0; r < repetitions; ++r) {
          tree[n] = value;
          extra[n] = 0;
          n++;
        }
      } else {
        const uint32_t start = n;
        repetitions -= 3;
        for (;;) {
          tree[n] = 16;
          extra[n] = (uint8_t)(repetitions & 0x3);
          n++;
          repetitions >>= 2;
          if (repetitions == 0) break;
          repetitions--;
        }
        br_reverse_u8(tree, start, n);
        br_reverse_u8(extra, start, n);
      }
      previous_value = value;
    }
    i += reps;
  }
  *tree_size = n;
}

// BrotliReverseBits / BrotliConvertBitDepthsToSymbols, entropy_encode.rs:527-575
BR_DEV uint16_t br_reverse_bits(uint32_t num_bits, uint16_t bits) {
#if !defined(BROTLI_HOST_EMU)
  if (num_bits != 0) return (uint16_t)(__builtin_bitreverse32((uint32_t)bits) >> (32u - num_bits));  // (1..15 bits: the same value)
#endif
  const uint32_t kLut[16] = {0x0, 0x8, 0x4, 0xc, 0x2, 0xa, 0x6, 0xe, 0x1, 0x9, 0x5, 0xd, 0x3, 0xb, 0x7, 0xf};
  uint32_t retval = kLut[bits & 0xf];
  for (uint32_t i = 4; i < num_bits; i += 4) {
    retval <<= 4;
    bits = (uint16_t)(bits >> 4);
    retval |= kLut[bits & 0xf];
  }
  retval >>= ((0u - num_bits) & 0x3u);
  return (uint16_t)retval;
}
BR_DEV void br_convert_bit_depths_to_symbols(const uint8_t* depth, uint32_t len, uint16_t* bits) {
  uint16_t bl_count[16];
  uint16_t next_code[16];
  for (int i = 0; i < 16; ++i) bl_count[i] = 0;
  for (uint32_t i = 0; i < len; ++i) bl_count[depth[i]]++;
  bl_count[0] = 0;
  next_code[0] = 0;
  int code = 0;
  for (int i = 1; i < 16; ++i) {
    code = (code + bl_count[i - 1]) << 1;
    next_code[i] = (uint16_t)code;
  }
  for (uint32_t i = 0; i < len; ++i)
    if (depth[i] != 0) bits[i] = br_reverse_bits(depth[i], next_code[depth[i]]++);
}
#if !defined(BROTLI_HOST_EMU)
// the same by all 64 lanes of a wavefront (depth / bits in LDS or global memory): lane k keeps the running next_code[k];
// a symbol's code is next_code[its depth] plus the number of symbols of that depth in front of it
BR_DEV void br_convert_bit_depths_to_symbols_coop(const uint8_t* depth, uint32_t len, uint16_t* bits) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t cnt = 0;
  for (uint32_t base = 0; base < len; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t d = i < len ? depth[i] : 0u;
#pragma unroll
    for (uint32_t k = 1; k < 16; ++k) {
      const unsigned long long m = __ballot(d == k);
      if (lane == k) cnt += (uint32_t)__popcll(m);
    }
  }
  uint32_t code = 0, running = 0;
#pragma unroll
  for (uint32_t k = 1; k < 16; ++k) {
    const uint32_t c_prev = k == 1 ? 0u : (uint32_t)__shfl((int)cnt, (int)(k - 1), 64);
    code = (code + c_prev) << 1;
    if (lane == k) running = code;
  }
  for (uint32_t base = 0; base < len; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t d = i < len ? depth[i] : 0u;
    const uint32_t next = (uint32_t)__shfl((int)running, (int)d, 64);
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t k = 1; k < 16; ++k) {
      const unsigned long long m = __ballot(d == k);
      if (d == k) rank = (uint32_t)__popcll(m & below);
      if (lane == k) running += (uint32_t)__popcll(m);
    }
    if (d != 0) bits[i] = br_reverse_bits(d, (uint16_t)(next + rank));
  }
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}
#endif

// Scratch for one Huffman build (per thread, in global memory)
struct HuffmanScratch {
  HuffmanTree tree[2 * 704 + 2];
  uint8_t rle_tree[704];
  uint8_t rle_extra[704];
  uint8_t good_for_rle[704];
  uint8_t pad[8];
  HuffmanTree sort_tmp[704];  // cooperative sort only
};

// BrotliStoreHuffmanTree (+ ...OfHuffmanTreeToBitMask, ...ToBitMask), brotli_bit_stream.rs:764-911
BR_DEV void br_store_huffman_tree(const uint8_t* depths, uint32_t num, HuffmanScratch* sc, BitSink& out) {
  const uint8_t kStorageOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  const uint8_t kSymbols[6] = {0, 7, 3, 2, 1, 15};
  const uint8_t kBitLengths[6] = {2, 4, 3, 2, 2, 4};
  uint32_t huffman_tree_size = 0;
  uint8_t code_length_bitdepth[18];
  uint16_t code_length_bitdepth_symbols[18];
  uint32_t huffman_tree_histogram[18];
  for (int i = 0; i < 18; ++i) {
    code_length_bitdepth[i] = 0;
    code_length_bitdepth_symbols[i] = 0;
    huffman_tree_histogram[i] = 0;
  }
  BR_PHASE_CLOCK();
  br_write_huffman_tree(depths, num, &huffman_tree_size, sc->rle_tree, sc->rle_extra);
  BR_PHASE(4);
  for (uint32_t i = 0; i < huffman_tree_size; ++i) huffman_tree_histogram[sc->rle_tree[i]]++;
  BR_PHASE(5);
  int num_codes = 0;
  uint32_t code = 0;
  for (uint32_t i = 0; i < 18; ++i) {
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
  br_create_huffman_tree(huffman_tree_histogram, 18, 5, sc->tree, code_length_bitdepth);
  br_convert_bit_depths_to_symbols(code_length_bitdepth, 18, code_length_bitdepth_symbols);
  {
    uint32_t skip_some = 0;
    uint32_t codes_to_store = 18;
    if (num_codes > 1) {
      for (; codes_to_store > 0; --codes_to_store)
        if (code_length_bitdepth[kStorageOrder[codes_to_store - 1]] != 0) break;
    }
    if (code_length_bitdepth[kStorageOrder[0]] == 0 && code_length_bitdepth[kStorageOrder[1]] == 0) {
      skip_some = 2;
      if (code_length_bitdepth[kStorageOrder[2]] == 0) skip_some = 3;
    }
    out.put(2, skip_some);
    for (uint32_t i = skip_some; i < codes_to_store; ++i) {
      const uint32_t l = code_length_bitdepth[kStorageOrder[i]];
      out.put(kBitLengths[l], kSymbols[l]);
    }
  }
  if (num_codes == 1) code_length_bitdepth[code] = 0;
  BR_PHASE(6);
  for (uint32_t i = 0; i < huffman_tree_size; ++i) {
    const uint32_t ix = sc->rle_tree[i];
    out.put(code_length_bitdepth[ix], code_length_bitdepth_symbols[ix]);
    if (ix == 16) {
      out.put(2, sc->rle_extra[i]);
    } else if (ix == 17) {
      out.put(3, sc->rle_extra[i]);
    }
  }
  BR_PHASE(7);
}

// BuildAndStoreHuffmanTree, brotli_bit_stream.rs:1445-1498 (+ StoreSimpleHuffmanTree :1401-1443)
BR_DEV void br_build_and_store_huffman_tree(const uint32_t* histogram, uint32_t histogram_length, uint32_t alphabet_size,
                                            HuffmanScratch* sc, uint8_t* depth, uint16_t* bits, BitSink& out, bool coop = false) {
  uint32_t count = 0;
  uint32_t s4[4] = {0, 0, 0, 0};
  uint32_t max_bits = 0;
  for (uint32_t i = 0; i < histogram_length; ++i) {
    if (histogram[i] != 0) {
      if (count < 4) {
        s4[count] = i;
      } else if (count > 4) {
        break;
      }
      count++;
    }
  }
  for (uint32_t c = alphabet_size - 1; c != 0; c >>= 1) max_bits++;
  if (count <= 1) {
    out.put(4, 1);
    out.put(max_bits, s4[0]);
    depth[s4[0]] = 0;
    bits[s4[0]] = 0;
    return;
  }
  for (uint32_t i = 0; i < histogram_length; ++i) depth[i] = 0;
  BR_PHASE_CLOCK();
  br_create_huffman_tree(histogram, histogram_length, 15, sc->tree, depth, coop ? sc->sort_tmp : nullptr);
  BR_PHASE(14);
#if !defined(BROTLI_HOST_EMU)
  if (coop) br_convert_bit_depths_to_symbols_coop(depth, histogram_length, bits); else
#endif
  br_convert_bit_depths_to_symbols(depth, histogram_length, bits);
  BR_PHASE(8);
  if (count <= 4) {
    out.put(2, 1);
    out.put(2, count - 1);
    for (uint32_t i = 0; i < count; ++i) {
      for (uint32_t j = i + 1; j < count; ++j) {
        if (depth[s4[j]] < depth[s4[i]]) {
          const uint32_t t = s4[j];
          s4[j] = s4[i];
          s4[i] = t;
        }
      }
    }
    for (uint32_t i = 0; i < count; ++i) out.put(max_bits, s4[i]);
    if (count == 4) out.put(1, depth[s4[0]] == 1 ? 1 : 0);
  } else {
    br_store_huffman_tree(depth, histogram_length, sc, out);
  }
}

// ---------------------------------------------------------------------------------------------- block split code
BR_DEV uint32_t br_block_length_offset(uint32_t code) {
  const uint32_t kOffset[26] = {1,   5,   9,   13,  17,  25,  33,  41,  49,   65,   81,   97,   113,
                                145, 177, 209, 241, 305, 369, 497, 753, 1265, 2289, 4337, 8433, 16625};
  return kOffset[code];
}
BR_DEV uint32_t br_block_length_nbits(uint32_t code) {
  const uint8_t kNbits[26] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24};
  return kNbits[code];
}
// BlockLengthPrefixCode, brotli_bit_stream.rs:1372-1388
BR_DEV uint32_t br_block_length_prefix_code(uint32_t len) {
  uint32_t code = len >= 177 ? (len >= 753 ? 20 : 14) : (len >= 41 ? 7 : 0);
  while (code < 25 && len >= br_block_length_offset(code + 1)) code++;
  return code;
}
struct BlockTypeCodeCalculator {
  uint32_t last_type, second_last_type;
};
// NextBlockTypeCode, brotli_bit_stream.rs:1357-1370
BR_DEV uint32_t br_next_block_type_code(BlockTypeCodeCalculator& c, uint8_t type) {
  const uint32_t type_code = (type == c.last_type + 1) ? 1u : (type == c.second_last_type ? 0u : (uint32_t)type + 2u);
  c.second_last_type = c.last_type;
  c.last_type = type;
  return type_code;
}
// StoreVarLenUint8, brotli_bit_stream.rs:1390-1399
BR_DEV void br_store_var_len_uint8(uint32_t n, BitSink& out) {
  if (n == 0) {
    out.put(1, 0);
  } else {
    const uint32_t nbits = br_log2_floor_nonzero(n);
    out.put(1, 1);
    out.put(3, nbits);
    out.put(nbits, n - (1u << nbits));
  }
}

struct BlockSplitCode {
  uint8_t type_depths[258];
  uint16_t type_bits[258];
  uint8_t length_depths[26];
  uint16_t length_bits[26];
};

// Block-switch command for one block (StoreBlockSwitch, brotli_bit_stream.rs:1506-1534) packed as
// (nbits, bits); at most 15 + 15 + 24 bits.
BR_DEV uint64_t br_block_switch_bits(const BlockSplitCode& code, uint32_t typecode, uint32_t block_len, bool is_first_block,
                                     uint32_t* nbits_out) {
  uint64_t bits = 0;
  uint32_t n = 0;
  if (!is_first_block) {
    bits = code.type_bits[typecode];
    n = code.type_depths[typecode];
  }
  const uint32_t lencode = br_block_length_prefix_code(block_len);
  bits |= (uint64_t)code.length_bits[lencode] << n;
  n += code.length_depths[lencode];
  bits |= (uint64_t)(block_len - br_block_length_offset(lencode)) << n;
  n += br_block_length_nbits(lencode);
  *nbits_out = n;
  return bits;
}

// BuildAndStoreBlockSplitCode, brotli_bit_stream.rs:1536-1591.  Also fills switch_bits/switch_nbits for
// every block after the first (the command emitted in front of the first symbol of block i).
// The type code of block i without the running calculator: the last type is that of block i - 1, the one before that of
// block i - 2 (the calculator starts at 1 / 0) -- every block on its own (k_write_headers).
BR_DEV uint32_t br_block_type_code_at(const uint8_t* types, uint32_t i) {
  const uint32_t type = types[i];
  const uint32_t last = i >= 1 ? types[i - 1] : 1u;
  const uint32_t second_last = i >= 2 ? types[i - 2] : (i == 1 ? 1u : 0u);
  return (type == last + 1) ? 1u : (type == second_last ? 0u : type + 2u);
}
// histograms_in (type histogram [258] followed by length histogram [26]): counted by the caller, who then also fills
// switch_bits / switch_nbits of the blocks behind the first one from *code (the wave of k_write_headers, a block per lane;
// the two walks over all blocks were most of what the one lane composing a header did)
BR_DEV void br_build_and_store_block_split_code(const uint8_t* types, const uint32_t* lengths, uint32_t num_blocks,
                                                uint32_t num_types, HuffmanScratch* sc, BlockSplitCode* code,
                                                uint64_t* switch_bits, uint8_t* switch_nbits, BitSink& out,
                                                const uint32_t* histograms_in = nullptr) {
  uint32_t type_histo[258];
  uint32_t length_histo[26];
  BlockTypeCodeCalculator calc;
  if (histograms_in == nullptr) {
    for (int i = 0; i < 258; ++i) type_histo[i] = 0;
    for (int i = 0; i < 26; ++i) length_histo[i] = 0;
    calc.last_type = 1;
    calc.second_last_type = 0;
    for (uint32_t i = 0; i < num_blocks; ++i) {
      const uint32_t type_code = br_next_block_type_code(calc, types[i]);
      if (i != 0) type_histo[type_code]++;
      length_histo[br_block_length_prefix_code(lengths[i])]++;
    }
  }
  br_store_var_len_uint8(num_types - 1, out);
  if (num_types > 1) {
    br_build_and_store_huffman_tree(histograms_in ? histograms_in : type_histo, num_types + 2, num_types + 2, sc, code->type_depths, code->type_bits, out);
    br_build_and_store_huffman_tree(histograms_in ? histograms_in + 258 : length_histo, 26, 26, sc, code->length_depths, code->length_bits, out);
    if (histograms_in != nullptr) {
      uint32_t nb;
      const uint64_t b = br_block_switch_bits(*code, 0, lengths[0], true, &nb);
      out.put(nb, b);
      return;
    }
    calc.last_type = 1;
    calc.second_last_type = 0;
    for (uint32_t i = 0; i < num_blocks; ++i) {
      const uint32_t typecode = br_next_block_type_code(calc, types[i]);
      uint32_t nb;
      const uint64_t b = br_block_switch_bits(*code, typecode, lengths[i], i == 0, &nb);
      if (i == 0) {
        out.put(nb, b);
      } else {
        switch_bits[i] = b;
        switch_nbits[i] = (uint8_t)nb;
      }
    }
  }
}

// StoreTrivialContextMap, brotli_bit_stream.rs:1613-1662
BR_DEV void br_store_trivial_context_map(uint32_t num_types, uint32_t context_bits, HuffmanScratch* sc, BitSink& out) {
  br_store_var_len_uint8(num_types - 1, out);
  if (num_types > 1) {
    const uint32_t repeat_code = context_bits - 1;
    const uint32_t repeat_bits = (1u << repeat_code) - 1;
    const uint32_t alphabet_size = num_types + repeat_code;
    uint32_t histogram[272];
    uint8_t depths[272];
    uint16_t bits[272];
    for (int i = 0; i < 272; ++i) {
      histogram[i] = 0;
      depths[i] = 0;
      bits[i] = 0;
    }
    out.put(1, 1);
    out.put(4, repeat_code - 1);
    histogram[repeat_code] = num_types;
    histogram[0] = 1;
    for (uint32_t i = context_bits; i < alphabet_size; ++i) histogram[i] = 1;
    br_build_and_store_huffman_tree(histogram, alphabet_size, alphabet_size, sc, depths, bits, out);
    for (uint32_t i = 0; i < num_types; ++i) {
      const uint32_t code = i == 0 ? 0 : i + context_bits - 1;
      out.put(depths[code], bits[code]);
      out.put(depths[repeat_code], bits[repeat_code]);
      out.put(repeat_code, repeat_bits);
    }
    out.put(1, 1);
  }
}

// EncodeContextMap (+ MoveToFrontTransform, RunLengthCodeZeros), brotli_bit_stream.rs:1690-1858.
// rle_symbols: scratch of context_map_size uint32.
BR_DEV void br_encode_context_map(const uint32_t* context_map, uint32_t context_map_size, uint32_t num_clusters,
                                  uint32_t* rle_symbols, HuffmanScratch* sc, BitSink& out) {
  br_store_var_len_uint8(num_clusters - 1, out);
  if (num_clusters == 1) return;
  // MoveToFrontTransform
  {
#if defined(BROTLI_HOST_EMU)
    uint8_t mtf[256];
#else
    __shared__ uint8_t mtf[256];  // (one lane works here; a private array would live in scratch memory)
#endif
    uint32_t max_value = context_map[0];
    for (uint32_t i = 1; i < context_map_size; ++i)
      if (context_map[i] > max_value) max_value = context_map[i];
    for (uint32_t i = 0; i <= max_value; ++i) mtf[i] = (uint8_t)i;
    const uint32_t mtf_size = max_value + 1;
    for (uint32_t i = 0; i < context_map_size; ++i) {
      uint32_t index = 0;
      while (index < mtf_size && mtf[index] != (uint8_t)context_map[i]) index++;
      rle_symbols[i] = index;
      const uint8_t value = mtf[index];
      for (uint32_t k = index; k != 0; --k) mtf[k] = mtf[k - 1];
      mtf[0] = value;
    }
  }
  // RunLengthCodeZeros
  uint32_t max_run_length_prefix = 6;
  uint32_t num_rle_symbols = 0;
  {
    uint32_t max_reps = 0;
    for (uint32_t i = 0; i < context_map_size;) {
      uint32_t reps = 0;
      for (; i < context_map_size && rle_symbols[i] != 0; ++i) {
      }
      for (; i < context_map_size && rle_symbols[i] == 0; ++i) reps++;
      if (reps > max_reps) max_reps = reps;
    }
    uint32_t max_prefix = max_reps > 0 ? br_log2_floor_nonzero(max_reps) : 0;
    if (max_prefix > max_run_length_prefix) max_prefix = max_run_length_prefix;
    max_run_length_prefix = max_prefix;
    uint32_t o = 0;
    for (uint32_t i = 0; i < context_map_size;) {
      if (rle_symbols[i] != 0) {
        rle_symbols[o++] = rle_symbols[i] + max_run_length_prefix;
        i++;
      } else {
        uint32_t reps = 1;
        for (uint32_t k = i + 1; k < context_map_size && rle_symbols[k] == 0; ++k) reps++;
        i += reps;
        while (reps != 0) {
          if (reps < (2u << max_prefix)) {
            const uint32_t run_length_prefix = br_log2_floor_nonzero(reps);
            const uint32_t extra_bits = reps - (1u << run_length_prefix);
            rle_symbols[o++] = run_length_prefix + (extra_bits << 9);
            break;
          } else {
            const uint32_t extra_bits = (1u << max_prefix) - 1;
            rle_symbols[o++] = max_prefix + (extra_bits << 9);
            reps -= (2u << max_prefix) - 1;
          }
        }
      }
    }
    num_rle_symbols = o;
  }
#if defined(BROTLI_HOST_EMU)
  uint32_t histogram[272];
  uint8_t depths[272];
  uint16_t bits[272];
#else
  __shared__ uint32_t histogram[272];
  __shared__ uint8_t depths[272];
  __shared__ uint16_t bits[272];
#endif
  for (int i = 0; i < 272; ++i) {
    histogram[i] = 0;
    depths[i] = 0;
    bits[i] = 0;
  }
  const uint32_t kSymbolMask = (1u << 9) - 1;
  for (uint32_t i = 0; i < num_rle_symbols; ++i) histogram[rle_symbols[i] & kSymbolMask]++;
  {
    const bool use_rle = max_run_length_prefix > 0;
    out.put(1, use_rle ? 1 : 0);
    if (use_rle) out.put(4, max_run_length_prefix - 1);
  }
  br_build_and_store_huffman_tree(histogram, num_clusters + max_run_length_prefix, num_clusters + max_run_length_prefix, sc,
                                  depths, bits, out);
  for (uint32_t i = 0; i < num_rle_symbols; ++i) {
    const uint32_t rle_symbol = rle_symbols[i] & kSymbolMask;
    const uint32_t extra_bits_val = rle_symbols[i] >> 9;
    out.put(depths[rle_symbol], bits[rle_symbol]);
    if (rle_symbol > 0 && rle_symbol <= max_run_length_prefix) out.put(rle_symbol, extra_bits_val);
  }
  out.put(1, 1);
}

// BrotliEncodeMlen + StoreCompressedMetaBlockHeader, brotli_bit_stream.rs:1272-1311
BR_DEV void br_store_compressed_meta_block_header(bool is_final_block, uint32_t length, BitSink& out) {
  out.put(1, is_final_block ? 1 : 0);
  if (is_final_block) out.put(1, 0);
  const uint32_t lg = length == 1 ? 1 : br_log2_floor_nonzero(length - 1) + 1;
  const uint32_t mnibbles = (lg < 16 ? 16 : lg + 3) / 4;
  out.put(2, mnibbles - 4);
  out.put(mnibbles * 4, length - 1);
  if (!is_final_block) out.put(1, 0);
}

// Context(), histogram.rs:448-463
BR_DEV uint32_t br_context(const uint8_t* utf8_lut, const uint8_t* signed_lut, uint8_t p1, uint8_t p2, uint32_t mode) {
  switch (mode) {
    case 3: return (uint32_t)((signed_lut[p1] << 3) + signed_lut[p2]);
    case 2: return (uint32_t)(utf8_lut[p1] | utf8_lut[256 + p2]);
    case 1: return (uint32_t)(p1 >> 2);
    default: return (uint32_t)(p1 & 0x3f);
  }
}

// static context maps, encode.rs:1723-1732, 1782-1798
BR_DEV uint32_t br_static_context_map(uint32_t map_id, uint32_t context) {
  const uint8_t kComplex[64] = {11, 11, 12, 12, 0, 0, 0, 0, 1, 1, 9, 9, 2, 2, 2, 2, 1, 1, 1, 1, 8, 3, 3, 3, 1, 1, 1, 1, 2, 2, 2, 2,
                                8,  4,  4,  4,  8, 7, 4, 4, 8, 0, 0, 0, 3, 3, 3, 3, 5, 5, 10, 5, 5, 5, 10, 5, 6, 6, 6, 6, 6, 6, 6, 6};
  switch (map_id) {
    case 1: return (context == 2 || context == 3) ? 1u : 0u;                       // kStaticContextMapSimpleUTF8
    case 2: return context < 2 ? 1u : (context < 4 ? 2u : 0u);                    // kStaticContextMapContinuation
    case 3: return kComplex[context];                                             // kStaticContextMapComplexUTF8
    default: return 0;
  }
}

// copy_len_code, brotli_bit_stream.rs:1923-1929
BR_DEV uint32_t br_copy_len_code(const Command& c) {
  const uint32_t modifier = c.copy_len_ >> 25;
  const int32_t delta = (int32_t)(int8_t)(uint8_t)(modifier | ((modifier & 0x40) << 1));
  return (uint32_t)((int32_t)(c.copy_len_ & 0x01ffffffu) + delta);
}

BR_DEV uint32_t br_ins_base(uint32_t code) {
  const uint32_t k[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
  return k[code];
}
BR_DEV uint32_t br_ins_extra(uint32_t code) {
  const uint8_t k[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
  return k[code];
}
BR_DEV uint32_t br_copy_base(uint32_t code) {
  const uint32_t k[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
  return k[code];
}
BR_DEV uint32_t br_copy_extra(uint32_t code) {
  const uint8_t k[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
  return k[code];
}

// StoreCommandExtra, brotli_bit_stream.rs:1947-1961: (nbits, bits)
BR_DEV uint64_t br_command_extra_bits(const Command& cmd, uint32_t* nbits) {
  const uint32_t copylen_code = br_copy_len_code(cmd);
  const uint32_t inscode = br_insert_length_code(cmd.insert_len_);
  const uint32_t copycode = br_copy_length_code(copylen_code);
  const uint32_t insnumextra = br_ins_extra(inscode);
  const uint64_t insextraval = cmd.insert_len_ - br_ins_base(inscode);
  const uint64_t copyextraval = copylen_code - br_copy_base(copycode);
  *nbits = insnumextra + br_copy_extra(copycode);
  return (copyextraval << insnumextra) | insextraval;
}

BR_DEV bool br_command_has_distance(const Command& c) { return (c.copy_len_ & 0x01ffffffu) != 0 && c.cmd_prefix_ >= 128; }

}  // namespace brotli_mi355x
#endif
