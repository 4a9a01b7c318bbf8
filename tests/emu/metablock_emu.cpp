// tests/emu/metablock_emu.cpp -- serial CPU stand-in for the meta-block device seam (metablock_api.h).
// TEST INFRASTRUCTURE ONLY (see device_emu.cpp).
#define BROTLI_HOST_EMU 1
#include <string.h>

#include <vector>

#include "../../rust-brotli_amd/csrc/metablock_api.h"
#include "../../rust-brotli_amd/csrc/metablock_hq.h"

namespace brotli_mi355x {

size_t mb_scan_scratch_bytes(size_t) { return 64; }

static void exclusive_scan(uint32_t* a, size_t n) {
  uint32_t run = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t v = a[i];
    a[i] = run;
    run += v;
  }
}

void mb_gather_at_metablock_starts(const MbBuffers& B, const uint32_t* src, uint32_t* out) {
  for (uint32_t m = 0; m <= B.n_mb; ++m) out[m] = src[m < B.n_mb ? B.descs[m].cmd_offset : B.n_cmds];
}

void mb_gather_bytes(const uint8_t* text, const uint32_t* positions, uint32_t n, uint8_t* out) {
  for (uint32_t i = 0; i < n; ++i) out[i] = positions[i] == 0xffffffffu ? (uint8_t)0 : text[positions[i]];
}

void mb_command_scans(const MbBuffers& B, void*) {
  for (uint32_t c = 0; c < B.n_cmds; ++c) mb_item_command_counts(B, c);
  B.cmd_lit_start[B.n_cmds] = 0;
  B.cmd_pos[B.n_cmds] = 0;
  B.cmd_dist_index[B.n_cmds] = 0;
  exclusive_scan(B.cmd_lit_start, B.n_cmds + 1);
  exclusive_scan(B.cmd_pos, B.n_cmds + 1);
  exclusive_scan(B.cmd_dist_index, B.n_cmds + 1);
}

void mb_literal_map(const MbBuffers& B) {
  for (uint32_t i = 0; i < B.n_lits; ++i) mb_item_literal_map(B, i);
}

void mb_context_stats(const MbBuffers& B, uint32_t* stats) {
  for (uint32_t m = 0; m < B.n_mb; ++m) {
    uint32_t* s = stats + (size_t)m * kContextStatsWords;
    memset(s, 0, kContextStatsWords * 4);
    const MbDesc& d = B.descs[m];
    const uint32_t length = d.end - d.start;
    const uint32_t n_strides = length >= 64 ? (length - 64) / 4096 + 1 : 0;
    for (uint32_t t = 0; t < n_strides; ++t) {
      const uint8_t* p = B.text + d.start + t * 4096;
      const int lut[4] = {0, 0, 1, 2};
      int prev = lut[p[0] >> 6] * 3;
      for (uint32_t k = 1; k < 64; ++k) {
        s[prev + lut[p[k] >> 6]]++;
        prev = lut[p[k] >> 6] * 3;
      }
      uint8_t prev2 = p[0], prev1 = p[1];
      for (uint32_t k = 2; k < 64; ++k) {
        const uint8_t literal = p[k];
        const uint32_t context = br_static_context_map(3, br_context(B.utf8_lut, B.signed_lut, prev1, prev2, 2));
        s[480]++;
        s[16 + (literal >> 3)]++;
        s[48 + context * 32 + (literal >> 3)]++;
        prev2 = prev1;
        prev1 = literal;
      }
    }
  }
}

void mb_granule_histograms(const MbBuffers& B) {
  std::vector<uint32_t> lds(kMaxStaticContexts * 256);
  for (uint32_t g = 0; g < B.n_granules[kSplitLiteral]; ++g) mb_item_granule_histogram(B, kSplitLiteral, g, lds.data());
  for (uint32_t g = 0; g < B.n_granules[kSplitCommand]; ++g) mb_item_granule_histogram(B, kSplitCommand, g, lds.data());
  if (B.n_granules[kSplitDistance]) {
    memset(B.gran_hist[kSplitDistance], 0, (size_t)B.n_granules[kSplitDistance] * kNumDistanceHistoSymbols * 2);
    for (uint32_t c = 0; c < B.n_cmds; ++c) mb_item_distance_count(B, c);
  }
}

void mb_split_chains(const MbBuffers& B, bool /*wide*/) {
  static thread_local SplitScratch S;  // (BrotliEncoderCompressMulti runs chunks on several host threads)
  for (uint32_t m = 0; m < B.n_mb; ++m)
    for (uint32_t kind = 0; kind < 3; ++kind) mb_item_split_chain(B, m, kind, S);
}

void mb_build_codes(const MbBuffers& B, const CodeJob* jobs, uint32_t n_jobs) {
  for (uint32_t i = 0; i < n_jobs; ++i) mb_item_build_code(B, jobs[i].kind, jobs[i].row_index, jobs[i].num_distance_symbols, B.huff_scratch, jobs[i].mode);
}

// the form of k_write_headers: block-split histograms and switch commands a block at a time around the header's composition
// (BROTLI_EMU_PLAIN_HEADERS=1: the one-lane form, every block walked inside br_build_and_store_block_split_code)
void mb_write_headers(const MbBuffers& B) {
  static const bool plain = getenv("BROTLI_EMU_PLAIN_HEADERS") != nullptr;
  for (uint32_t m = 0; m < B.n_mb; ++m) {
    const MbDesc d = B.descs[m];
    if (plain || d.uncompressed) {
      mb_item_write_header(B, m, B.huff_scratch);
      continue;
    }
    static thread_local MbSplitPrepared prep;
    memset(&prep.histograms[0][0], 0, sizeof(prep.histograms));
    for (uint32_t kind = 0; kind < 3; ++kind) {
      const uint8_t* types = B.block_types[kind] + d.block_base[kind];
      const uint32_t* lengths = B.block_lengths[kind] + d.block_base[kind];
      const uint32_t nb = B.results[m].num_blocks[kind];
      for (uint32_t i = 0; i < nb; ++i) {
        if (i != 0) prep.histograms[kind][br_block_type_code_at(types, i)]++;
        prep.histograms[kind][258 + br_block_length_prefix_code(lengths[i])]++;
      }
    }
    mb_item_write_header(B, m, B.huff_scratch, nullptr, &prep);
    for (uint32_t kind = 0; kind < 3; ++kind) {
      if (B.results[m].num_types[kind] <= 1) continue;
      const uint8_t* types = B.block_types[kind] + d.block_base[kind];
      const uint32_t* lengths = B.block_lengths[kind] + d.block_base[kind];
      const uint32_t nb = B.results[m].num_blocks[kind];
      for (uint32_t i = 1; i < nb; ++i) {
        uint32_t nbits;
        B.switch_bits[kind][d.block_base[kind] + i] = br_block_switch_bits(prep.code[kind], br_block_type_code_at(types, i), lengths[i], false, &nbits);
        B.switch_nbits[kind][d.block_base[kind] + i] = (uint8_t)nbits;
      }
    }
  }
}

void mb_symbol_bits(const MbBuffers& B, void*) {
  for (uint32_t i = 0; i < B.n_lits; ++i) mb_item_literal_nbits(B, i);
  B.lit_nbits[B.n_lits] = 0;
  exclusive_scan(B.lit_nbits, B.n_lits + 1);
  for (uint32_t c = 0; c < B.n_cmds; ++c) mb_item_command_nbits(B, c);
  B.cmd_nbits[B.n_cmds] = 0;
  exclusive_scan(B.cmd_nbits, B.n_cmds + 1);
}

void mb_emit(const MbBuffers& B) {
  for (uint32_t c = 0; c < B.n_cmds; ++c) mb_item_emit_command(B, c);
  for (uint32_t i = 0; i < B.n_lits; ++i) mb_item_emit_literal(B, i);
}

void mb_hq_utf8_census(const MbBuffers& B) {
  HqCensusScratch S;
  for (uint32_t m = 0; m < B.n_mb; ++m)
    if (!B.descs[m].uncompressed) hq_item_utf8_census(B, m, S);
}
void mb_hq_distance_params(const MbBuffers& B) {
  HqWaveScratch S;
  for (uint32_t m = 0; m < B.n_mb; ++m) hq_item_distance_params(B, m, S);
}
void mb_hq_gather_symbols(const MbBuffers& B) {
  for (uint32_t i = 0; i < B.n_lits; ++i) hq_item_literal_symbol(B, i);
  for (uint32_t c = 0; c < B.n_cmds; ++c) hq_item_command_symbols(B, c);
}
void mb_hq_find_blocks(const MbBuffers& B, HqSplitJob* jobs, uint32_t n_jobs) {
  HqWaveScratch S;
  for (uint32_t i = 0; i < n_jobs; ++i) hq_item_find_blocks(B.et, jobs[i], S);
}
void mb_hq_cluster_blocks(const MbBuffers& B, const HqSplitJob* jobs, uint32_t n_jobs, const HqBatchRef* batches, uint32_t n_batches) {
  static thread_local HqWaveScratch S;
  static thread_local HqBatchPairs P;
  for (uint32_t i = 0; i < n_jobs; ++i) hq_item_blocks_prep(jobs[i]);
  for (uint32_t i = 0; i < n_batches; ++i) hq_item_cluster_blocks_batch(B.et, jobs[batches[i].job], batches[i].batch, S, P.pairs);
  for (uint32_t i = 0; i < n_jobs; ++i) hq_item_cluster_blocks(B, jobs[i], S);
}
void mb_hq_context_histograms(const MbBuffers& B) {
  for (uint32_t i = 0; i < B.n_lits; ++i) hq_item_literal_context_count(B, i);
  for (uint32_t c = 0; c < B.n_cmds; ++c) hq_item_command_context_count(B, c);
}
void mb_hq_cluster_histograms(const MbBuffers& B, const HqClusterJob* jobs, uint32_t n_jobs, const HqBatchRef* batches, uint32_t n_batches) {
  static thread_local HqWaveScratch S;
  static thread_local HqBatchPairs P;
  for (uint32_t i = 0; i < n_batches; ++i) hq_item_cluster_histograms_batch(B.et, jobs[batches[i].job], batches[i].batch, S, P.pairs);
  for (uint32_t i = 0; i < n_jobs; ++i) hq_item_cluster_histograms(B, jobs[i], S);
}

void mb_copy_bits(uint64_t* out, uint64_t dst_bit, const uint64_t* src, uint64_t nbits) {
  const uint64_t words = (nbits + 63) / 64;
  for (uint64_t w = 0; w < words; ++w) mb_item_copy_bits_word(out, dst_bit, src, nbits, w);
}

void mb_copy_bits_batch(uint64_t* out, const uint64_t* src_words, const MbBitCopy* items, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) mb_copy_bits(out, items[i].dst_bit, src_words + items[i].src_word, items[i].nbits);
}
void mb_place_pieces(uint64_t* out, const MbBitPiece* pieces, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t w = pieces[i].bits;
    mb_copy_bits(out, pieces[i].pos, &w, pieces[i].nbits);
  }
}
void mb_raw_copies(uint8_t* out_bytes, const uint8_t* text, const MbRawCopy* items, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) memcpy(out_bytes + items[i].dst_byte, text + items[i].src_pos, items[i].bytes);
}

}  // namespace brotli_mi355x
