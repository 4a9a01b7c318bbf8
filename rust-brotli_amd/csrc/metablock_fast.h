// metablock_fast.h -- the meta-block writers of qualities 2 and 3 as device code, on top of the kernels of
// the greedy path: the histograms, code construction jobs, header pass and emission are the same kernels, told by
// MbDesc::simple / CodeJob::mode what is different:
//   quality 3, store_meta_block_trivial (brotli_bit_stream.rs:2345-2465): one block type per kind (no splitter), histograms
//     straight into BuildAndStoreHuffmanTree (no BrotliOptimizeHuffmanCountsForRle), context mode bits 0;
//   quality 2, store_meta_block_fast (brotli_bit_stream.rs:2578-2742): the same shape with
//     BrotliBuildAndStoreHuffmanTreeFast (:925-1121) -- depth limit 14, counts only in the sort, a fixed code-length code --
//     and, for at most 128 commands, the static command / distance codes (:2467-2474).
// Shared by the gfx950 kernels (metablock_kernels.hip) and the host emulation (tests/emu).
#ifndef BROTLI_MI355X_METABLOCK_FAST_H_
#define BROTLI_MI355X_METABLOCK_FAST_H_

#include "metablock_device.h"

#if defined(BROTLI_HOST_EMU)
#define BROTLI_FAST_TABLE __attribute__((unused)) static const
#else
#define BROTLI_FAST_TABLE __attribute__((unused)) static __device__ const
#endif
#include "../../tables/brotli_fast_tables.h"

namespace brotli_mi355x {

// MbDesc::simple
static constexpr uint32_t kMbGreedy = 0, kMbTrivial = 1, kMbFast = 2;
// CodeJob::mode
static constexpr uint32_t kCodeOptimized = 0;  // BrotliOptimizeHistograms + BuildAndStoreHuffmanTree (quality >= 4)
static constexpr uint32_t kCodePlain = 1;      // BuildAndStoreHuffmanTree on the raw counts (quality 3)
static constexpr uint32_t kCodeFast = 2;       // BrotliBuildAndStoreHuffmanTreeFast (quality 2)
static constexpr uint32_t kCodeStatic = 3;     // the static command / distance code (quality 2, <= 128 commands)

// SortHuffmanTreeItems with SimpleSortHuffmanTree (brotli_bit_stream.rs:917-923, entropy_encode.rs:71-116): by count only --
// what equal counts end up as hangs on the method, so it is the reference's: insertion sort below 13 items, else its shell sort
BR_DEV void br_sort_huffman_tree_items_by_count(HuffmanTree* items, uint32_t n) {
  const uint32_t gaps[6] = {132, 57, 23, 10, 4, 1};
  if (n < 13) {
    for (uint32_t i = 1; i < n; ++i) {
      HuffmanTree tmp = items[i];
      uint32_t k = i;
      uint32_t j = i - 1;
      while (tmp.total_count_ < items[j].total_count_) {
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
        for (; j >= gap && tmp.total_count_ < items[j - gap].total_count_; j -= gap) items[j] = items[j - gap];
        items[j] = tmp;
      }
    }
  }
}

// BrotliBuildAndStoreHuffmanTreeFast, brotli_bit_stream.rs:925-1121.  depth / bits: rows of the histogram's length, zeroed by
// the caller.  (Called by all lanes of a wavefront in lock step on the same buffers, like the rest of a code job: every lane
// redoes the same scalar work.)
BR_DEV void br_build_and_store_huffman_tree_fast(const uint32_t* histogram, uint32_t histogram_total, uint32_t max_bits, HuffmanScratch* sc,
                                                 uint8_t* depth, uint16_t* bits, BitSink& out) {
  uint32_t count = 0;
  uint32_t symbols[4] = {0, 0, 0, 0};
  uint32_t length = 0;
  uint32_t total = histogram_total;
  while (total != 0) {
    if (histogram[length] != 0) {
      if (count < 4) symbols[count] = length;
      ++count;
      total -= histogram[length];
    }
    ++length;
  }
  if (count <= 1) {
    out.put(4, 1);
    out.put(max_bits, symbols[0]);
    depth[symbols[0]] = 0;
    bits[symbols[0]] = 0;
    return;
  }
  for (uint32_t i = 0; i < length; ++i) depth[i] = 0;
  {
    HuffmanTree* tree = sc->tree;
    HuffmanTree sentinel;
    sentinel.total_count_ = 0xffffffffu;
    sentinel.index_left_ = -1;
    sentinel.index_right_or_value_ = -1;
    for (uint32_t count_limit = 1;; count_limit *= 2) {
      uint32_t node_index = 0;
      for (uint32_t l = length; l != 0;) {
        --l;
        if (histogram[l] != 0) {
          tree[node_index].total_count_ = histogram[l] >= count_limit ? histogram[l] : count_limit;
          tree[node_index].index_left_ = -1;
          tree[node_index].index_right_or_value_ = (int16_t)l;
          ++node_index;
        }
      }
      const int n = (int)node_index;
      int i = 0, j = n + 1;
      br_sort_huffman_tree_items_by_count(tree, (uint32_t)n);
      tree[node_index + 1] = sentinel;
      tree[node_index] = sentinel;
      node_index += 2;
      for (int k = n - 1; k > 0; --k) {
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
      if (br_set_depth(2 * n - 1, tree, depth, 14)) break;
    }
  }
  br_convert_bit_depths_to_symbols(depth, length, bits);
  if (count <= 4) {
    out.put(2, 1);
    out.put(2, count - 1);
    for (uint32_t i = 0; i < count; ++i)
      for (uint32_t j = i + 1; j < count; ++j)
        if (depth[symbols[j]] < depth[symbols[i]]) {
          const uint32_t t = symbols[j];
          symbols[j] = symbols[i];
          symbols[i] = t;
        }
    for (uint32_t i = 0; i < count; ++i) out.put(max_bits, symbols[i]);
    if (count == 4) out.put(1, depth[symbols[0]] == 1 ? 1 : 0);
  } else {
    uint8_t previous_value = 8;
    out.put(40, 0xff55555554ull);  // StoreStaticCodeLengthCode, :913-915
    for (uint32_t i = 0; i < length;) {
      const uint8_t value = depth[i];
      uint32_t reps = 1;
      for (uint32_t k = i + 1; k < length && depth[k] == value; ++k) ++reps;
      i += reps;
      if (value == 0) {
        out.put(kZeroRepsDepth[reps], kZeroRepsBits[reps]);
      } else {
        if (previous_value != value) {
          out.put(kCodeLengthDepth[value], kCodeLengthBits[value]);
          --reps;
        }
        if (reps < 3) {
          while (reps != 0) {
            --reps;
            out.put(kCodeLengthDepth[value], kCodeLengthBits[value]);
          }
        } else {
          reps -= 3;
          out.put(kNonZeroRepsDepth[reps], kNonZeroRepsBits[reps]);
        }
        previous_value = value;
      }
    }
  }
}

// StoreStaticCommandHuffmanTree / StoreStaticDistanceHuffmanTree (:2467-2474) and the codes that go with them
BR_DEV void br_store_static_code(uint32_t kind, uint8_t* depth, uint16_t* bits, BitSink& out) {
  if (kind == kSplitCommand) {
    for (uint32_t i = 0; i < 704; ++i) {
      depth[i] = kStaticCommandCodeDepth[i];
      bits[i] = kStaticCommandCodeBits[i];
    }
    out.put(56, 0x0092624416307003ull);
    out.put(3, 0);
  } else {
    for (uint32_t i = 0; i < 64; ++i) {
      depth[i] = kStaticDistanceCodeDepth[i];
      bits[i] = kStaticDistanceCodeBits[i];
    }
    out.put(28, 0x0369dc03ull);
  }
}

}  // namespace brotli_mi355x
#endif
