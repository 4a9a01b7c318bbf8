// metablock_types.h -- plain-old-data shared by the host driver and the kernels of the meta-block stage
// (greedy block splitting, Huffman code construction, header serialisation, symbol emission).
//
// Vocabulary follows the reference: "meta-block", "block split" (types/lengths), "literal context map",
// "histogram" (src/enc/metablock.rs, histogram.rs, brotli_bit_stream.rs).  New here: "granule" = the
// min_block_size_ symbols (512 literals / 1024 commands / 512 distances) between two possible split
// points of the greedy splitter (metablock.rs:885-928): every block boundary falls on a granule boundary.
#ifndef BROTLI_MI355X_METABLOCK_TYPES_H_
#define BROTLI_MI355X_METABLOCK_TYPES_H_

#include <stdint.h>

namespace brotli_mi355x {

static constexpr uint32_t kNumLiteralSymbols = 256;
static constexpr uint32_t kNumCommandSymbols = 704;
static constexpr uint32_t kNumDistanceHistoSymbols = 544;  // BROTLI_NUM_HISTOGRAM_DISTANCE_SYMBOLS
static constexpr uint32_t kMaxStaticContexts = 13;
static constexpr uint32_t kLiteralGranule = 512;
static constexpr uint32_t kCommandGranule = 1024;
static constexpr uint32_t kDistanceGranule = 512;
static constexpr uint32_t kMaxBlockTypes = 256;
static constexpr uint32_t kTreeBitsWords = 64;     // scratch for one serialised Huffman tree (<= 4096 bits)
static constexpr uint32_t kHeaderWords = 16384;    // scratch for one meta-block header (<= 1 Mi bits)
static constexpr uint32_t kHqHeaderWords = 131072; // quality >= 10: up to 3 x 256 trees of <= 4096 bits + two clustered context maps

enum SplitKind : uint32_t { kSplitLiteral = 0, kSplitCommand = 1, kSplitDistance = 2 };

// Everything the kernels need to know about one meta-block.  Filled by the host from the LZ77 plan
// and the context-modelling decision; offsets point into job-wide device arrays.
struct MbDesc {
  uint32_t start, end;          // text positions
  uint32_t cmd_offset, n_cmds;  // into the gathered command array
  uint32_t lit_base, n_lits;    // into the literal stream (lit_pos / lit_bits ...)
  uint32_t dist_base, n_dists;  // into the distance-symbol stream
  uint32_t prev_byte, prev_byte2;
  uint32_t num_contexts;        // 1, 2, 3 or 13 (encode.rs:1717-1927)
  uint32_t context_map_id;      // 0 none, 1 SimpleUTF8, 2 Continuation, 3 ComplexUTF8
  uint32_t context_mode;        // ContextType (histogram.rs:312-317), UTF8 = 2
  uint32_t is_last;             // ISLAST bit of the compressed meta-block header
  uint32_t uncompressed;
  uint32_t num_distance_symbols;    // params.dist.alphabet_size
  uint32_t dist_postfix_bits, num_direct_distance_codes;
  // granule bookkeeping (per split kind)
  uint32_t granule_base[3];     // first granule of this meta-block in the job-wide granule arrays
  uint32_t n_granules[3];
  uint32_t gran_row_base[3];    // first row of this meta-block in the granule histogram pools (literal rows: x num_contexts)
  uint32_t n_symbols[3];        // n_lits, n_cmds, n_dists
  // per split kind: where the chain writes its results
  uint32_t block_base[3];       // into block_types / block_lengths / block_switch arrays
  uint32_t max_blocks[3];       // capacity (n_symbols / granule + 1)
  uint32_t histo_base[3];       // first histogram slot (units of one histogram)
  uint32_t max_histos[3];       // capacity in histograms
  uint32_t header_word_base;    // into header scratch (uint64 words)
  // quality >= 10 (metablock_hq.h): splits with arbitrary block lengths (block_start instead of the granule index) and
  // clustered context maps instead of the static ones
  uint32_t hq;
  uint32_t hq_no_context;       // disable_literal_context_modeling: one literal histogram row per block type
  uint32_t hq_ctx_row_base[2];  // first row of this meta-block in the context histogram pools (literal, distance)
  uint32_t hq_ctx_map_base[2];  // first entry in the context map pools (num_types << 6 literal, << 2 distance)
  uint32_t simple;              // 0: greedy splitter (quality >= 4); 1 / 2: the one-block-type writers of quality 3 / 2 (metablock_fast.h)
};

// Results of the greedy splitters and the header pass, one per meta-block.
struct MbResult {
  uint32_t num_types[3];
  uint32_t num_blocks[3];
  uint32_t num_histos[3];      // literal: num_types * num_contexts
  uint32_t header_bits;        // length of the serialised header (incl. the meta-block header bits)
  uint32_t body_bits;          // commands + literals + distances
  uint32_t hq_postfix, hq_ndirect;  // distance parameters chosen by the quality >= 10 search
  uint32_t hq_mostly_utf8;          // BrotliIsMostlyUTF8 of the meta-block's bytes (ChooseContextMode)
  uint32_t pad[2];
};

}  // namespace brotli_mi355x
#endif
