// metablock_api.h -- device seam of the meta-block stage (see device_api.h for the conventions).
#ifndef BROTLI_MI355X_METABLOCK_API_H_
#define BROTLI_MI355X_METABLOCK_API_H_

#include <stddef.h>
#include <stdint.h>

#include "metablock_types.h"

namespace brotli_mi355x {

struct MbBuffers;  // metablock_items.h
struct HqSplitJob;    // metablock_hq.h
struct HqClusterJob;
struct HqBatchRef;

static constexpr uint32_t kContextStatsWords = 512;  // per meta-block: [0..9) bigram prefix histogram
                                                      // (encode.rs:1885-1918), [16..48) combined 5-bit histogram,
                                                      // [48..48+13*32) per-context histograms, [480] total (encode.rs:1802-1871)

struct CodeJob {
  uint32_t kind;
  uint32_t row_index;
  uint32_t num_distance_symbols;
  uint32_t mode;  // kCodeOptimized / kCodePlain / kCodeFast / kCodeStatic (metablock_fast.h)
};

size_t mb_scan_scratch_bytes(size_t n);
// per command insert_len / insert+copy / has-distance and their exclusive scans ([K] = totals)
void mb_command_scans(const MbBuffers& B, void* scan_scratch);
// out_dev[m] = src[first command of meta-block m], out_dev[n_mb] = src[n_cmds]  (descs must be on the device)
void mb_gather_at_metablock_starts(const MbBuffers& B, const uint32_t* src, uint32_t* out_dev);
// out[i] = text[positions[i]] (0 for position 0xffffffff): a handful of scattered bytes in one round trip
void mb_gather_bytes(const uint8_t* text, const uint32_t* positions_dev, uint32_t n, uint8_t* out_dev);
void mb_literal_map(const MbBuffers& B);
void mb_context_stats(const MbBuffers& B, uint32_t* stats_dev);
void mb_granule_histograms(const MbBuffers& B);
void mb_split_chains(const MbBuffers& B, bool wide = false);
void mb_build_codes(const MbBuffers& B, const CodeJob* jobs_dev, uint32_t n_jobs);
void mb_write_headers(const MbBuffers& B);
void mb_symbol_bits(const MbBuffers& B, void* scan_scratch);
void mb_emit(const MbBuffers& B);
void mb_copy_bits(uint64_t* out, uint64_t dst_bit, const uint64_t* src, uint64_t nbits);
// The tail of the emission in three launches, however many meta-blocks there are (an incompressible gigabyte is 700 stored
// meta-blocks, each with a handful of host-composed header pieces and one copy of its bytes):
// out |= bits [0, nbits) of src_words + src_word, placed at dst_bit, for every item (the headers of the compressed meta-blocks)
struct MbBitCopy {
  uint64_t dst_bit, src_word, nbits;
};
void mb_copy_bits_batch(uint64_t* out, const uint64_t* src_words, const MbBitCopy* items_dev, uint32_t n);
// out |= the low nbits of bits at bit position pos, for every piece (nbits <= 64): what the host composed
struct MbBitPiece {
  uint64_t pos;
  uint32_t nbits, pad;
  uint64_t bits;
};
void mb_place_pieces(uint64_t* out, const MbBitPiece* pieces_dev, uint32_t n);
// out_bytes[dst_byte, + bytes) = text[src_pos, + bytes) for every item: the bytes of the stored meta-blocks
struct MbRawCopy {
  uint64_t dst_byte;
  uint32_t src_pos, bytes;
};
void mb_raw_copies(uint8_t* out_bytes, const uint8_t* text, const MbRawCopy* items_dev, uint32_t n);

// ---- quality >= 10 (metablock_hq.h; row b10).  Order of a call: census + distance parameters (results: hq_*), symbol
// streams, FindBlocks iterations (jobs[].num_blocks), ClusterBlocks (the splits), context histograms, context-map
// clustering (histogram rows + maps + num_histos); the code / header / emission kernels above take it from there.
void mb_hq_utf8_census(const MbBuffers& B);
void mb_hq_distance_params(const MbBuffers& B);
void mb_hq_gather_symbols(const MbBuffers& B);
void mb_hq_find_blocks(const MbBuffers& B, HqSplitJob* jobs_dev, uint32_t n_jobs);
void mb_hq_cluster_blocks(const MbBuffers& B, const HqSplitJob* jobs_dev, uint32_t n_jobs, const HqBatchRef* batches_dev, uint32_t n_batches);
void mb_hq_context_histograms(const MbBuffers& B);
void mb_hq_cluster_histograms(const MbBuffers& B, const HqClusterJob* jobs_dev, uint32_t n_jobs, const HqBatchRef* batches_dev,
                              uint32_t n_batches);

}  // namespace brotli_mi355x
#endif
