// metablock_items.h -- the per-item bodies of the meta-block kernels.  Each function handles ONE work
// item (a command, a literal, a granule, a histogram, a meta-block ...); the HIP kernels in
// metablock_kernels.hip map them onto threads / workgroups, the CPU emulation in tests/emu loops over them.
#ifndef BROTLI_MI355X_METABLOCK_ITEMS_H_
#define BROTLI_MI355X_METABLOCK_ITEMS_H_

#include "metablock_fast.h"

#if defined(BROTLI_HOST_EMU)
#define BR_TID 0
#define BR_NT 1
#define BR_ATOMIC_ADD_U32(p, v) (*(p) += (v))
#define BR_ATOMIC_OR_U64(p, v) (*(p) |= (v))
#else
#define BR_TID ((int)threadIdx.x)
#define BR_NT ((int)blockDim.x)
#define BR_ATOMIC_ADD_U32(p, v) atomicAdd((p), (v))
#define BR_ATOMIC_OR_U64(p, v) atomicOr((unsigned long long*)(p), (unsigned long long)(v))
#endif

namespace brotli_mi355x {

struct MbBuffers {
  const uint8_t* text;
  const Command* cmds;
  uint32_t n_cmds, n_lits, n_dists, n_mb;
  uint32_t text_base;  // text position where the first command starts
  MbDesc* descs;
  MbResult* results;
  uint32_t* cmd_lit_start;   // [K+1] exclusive scan of insert_len
  uint32_t* cmd_pos;         // [K+1] exclusive scan of insert_len + copy_len
  uint32_t* cmd_dist_index;  // [K+1] exclusive scan of "has a distance symbol"
  uint32_t* lit_pos;         // [L] text position of every literal
  uint32_t* lit_cmd;         // [L] command that carries it
  uint32_t* gran_mb[3];      // per granule: meta-block index
  uint16_t* gran_hist[3];    // granule histograms, rows of 256 / 704 / 544 counters
  uint16_t* gran_block[3];   // per granule: index of the block (of its meta-block's split) it belongs to
  uint32_t n_granules[3];
  uint32_t* histo[3];        // block-type histograms produced by the splitters (rows of 256 / 704 / 544)
  uint8_t* depth[3];         // Huffman code lengths, same row layout
  uint16_t* bits[3];         // Huffman codes
  uint64_t* tree_bits[3];    // serialised tree per histogram row, kTreeBitsWords words each
  uint32_t* tree_nbits[3];
  uint8_t* block_types[3];
  uint32_t* block_lengths[3];
  uint64_t* switch_bits[3];
  uint8_t* switch_nbits[3];
  uint64_t* header_words;    // n_mb * kHeaderWords
  uint32_t* lit_nbits;       // [L+1] -> exclusive scan = bit offset of each literal inside the literal bit stream
  uint32_t* cmd_nbits;       // [K+1] -> exclusive scan = bit offset of each command inside the body
  uint32_t* cmd_own_bits;    // [K] bits emitted for the command itself before its literals
  HuffmanScratch* huff_scratch;
  uint32_t* ctxmap_scratch;  // n_mb * (256 * 64) words
  uint64_t* out_words;       // final stream (zero initialised)
  uint64_t* mb_out_bit;      // [n_mb] absolute bit offset of each meta-block's header in out_words
  const uint8_t* utf8_lut;
  const uint8_t* signed_lut;
  EntropyTables et;
  uint32_t header_stride;    // words of header scratch per meta-block (kHeaderWords / kHqHeaderWords)
  uint32_t pad_hq;
  // ---- quality >= 10 (metablock_hq.h)
  Command* cmds_rw;          // the stage's own copy of the commands (distance prefixes are re-coded per meta-block)
  uint16_t* hq_sym[3];       // symbol streams of the three splitters: [L] literals, [K] command prefixes, [D] distance codes
  uint32_t* block_start[3];  // per block: index of its first symbol inside the meta-block (one extra entry = the total)
  uint32_t* hq_ctx_histo[2]; // context histograms before clustering: literal rows of 256, distance rows of 544
  uint32_t* hq_ctx_map[2];   // clustered context maps
};

static constexpr uint32_t kRowLen[3] = {256, 704, 544};
static constexpr uint32_t kGranuleLen[3] = {kLiteralGranule, kCommandGranule, kDistanceGranule};

// ---- K1: per command prefix inputs
BR_DEV void mb_item_command_counts(const MbBuffers& B, uint32_t c) {
  const Command cmd = B.cmds[c];
  B.cmd_lit_start[c] = cmd.insert_len_;
  B.cmd_pos[c] = cmd.insert_len_ + (cmd.copy_len_ & 0x01ffffffu);
  B.cmd_dist_index[c] = br_command_has_distance(cmd) ? 1u : 0u;
}

// ---- K2: literal -> (command, text position)
BR_DEV void mb_item_literal_map(const MbBuffers& B, uint32_t i) {
  // last command c with cmd_lit_start[c] <= i and insert_len > 0 covering i
  uint32_t lo = 0, hi = B.n_cmds;  // invariant: cmd_lit_start[lo] <= i < cmd_lit_start[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (B.cmd_lit_start[mid] <= i) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  B.lit_cmd[i] = lo;
  B.lit_pos[i] = B.text_base + B.cmd_pos[lo] + (i - B.cmd_lit_start[lo]);
}

BR_DEV uint32_t mb_literal_context(const MbBuffers& B, const MbDesc& d, uint32_t pos) {
  const uint8_t p1 = pos >= d.start + 1 ? B.text[pos - 1] : (uint8_t)d.prev_byte;
  const uint8_t p2 = pos >= d.start + 2 ? B.text[pos - 2] : (pos == d.start + 1 ? (uint8_t)d.prev_byte : (uint8_t)d.prev_byte2);
  return br_context(B.utf8_lut, B.signed_lut, p1, p2, d.context_mode);
}

// ---- K3: granule histograms (one granule; threads of the workgroup stride over its symbols).  `lds` holds
// num_contexts * row counters (zeroed by the caller), results are written as u16 rows.
BR_DEV void mb_item_granule_histogram(const MbBuffers& B, uint32_t kind, uint32_t g, uint32_t* lds) {
  const uint32_t m = B.gran_mb[kind][g];
  const MbDesc d = B.descs[m];
  const uint32_t row = kRowLen[kind];
  const uint32_t nc = kind == kSplitLiteral ? d.num_contexts : 1;
  const uint32_t local_g = g - d.granule_base[kind];
  const uint32_t first = local_g * kGranuleLen[kind];
  uint32_t count = d.n_symbols[kind] - first;
  if (count > kGranuleLen[kind]) count = kGranuleLen[kind];
  for (uint32_t j = BR_TID; j < nc * row; j += BR_NT) lds[j] = 0;
  BR_SYNC();
  for (uint32_t j = BR_TID; j < count; j += BR_NT) {
    if (kind == kSplitLiteral) {
      const uint32_t pos = B.lit_pos[d.lit_base + first + j];
      const uint32_t lit = B.text[pos];
      uint32_t ctx = 0;
      if (nc > 1) ctx = br_static_context_map(d.context_map_id, mb_literal_context(B, d, pos));
      BR_ATOMIC_ADD_U32(&lds[ctx * row + lit], 1u);
    } else if (kind == kSplitCommand) {
      BR_ATOMIC_ADD_U32(&lds[B.cmds[d.cmd_offset + first + j].cmd_prefix_], 1u);
    }
  }
  BR_SYNC();
  if (kind != kSplitDistance) {
    uint16_t* out = B.gran_hist[kind] + ((size_t)d.gran_row_base[kind] + (size_t)local_g * nc) * row;
    for (uint32_t j = BR_TID; j < nc * row; j += BR_NT) out[j] = (uint16_t)lds[j];
  }
}

// distance symbols are scattered over the commands: one pass over the commands of the meta-block adds each
// distance code to its granule row (rows zeroed by the caller)
BR_DEV void mb_item_distance_count(const MbBuffers& B, uint32_t c) {
  const Command cmd = B.cmds[c];
  if (!br_command_has_distance(cmd)) return;
  // meta-block of the command
  uint32_t lo = 0, hi = B.n_mb;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (B.descs[mid].cmd_offset <= c) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  const MbDesc& d = B.descs[lo];
  if (d.uncompressed) return;
  const uint32_t local = B.cmd_dist_index[c] - d.dist_base;
  const uint32_t local_g = local / kDistanceGranule;
  uint16_t* rowp = B.gran_hist[kSplitDistance] + ((size_t)d.gran_row_base[kSplitDistance] + local_g) * kNumDistanceHistoSymbols;
  // u16 counters: use a 32-bit atomic on the containing word
  const uint32_t sym = cmd.dist_prefix_ & 0x3ffu;
  uint32_t* word = (uint32_t*)((uintptr_t)(rowp + sym) & ~(uintptr_t)3);
  const uint32_t shift = ((uintptr_t)(rowp + sym) & 2u) ? 16u : 0u;
  BR_ATOMIC_ADD_U32(word, 1u << shift);
}

// ---- K4: greedy block splitter chain for one (meta-block, kind).
// BlockSplitter / ContextBlockSplitter FinishBlock, metablock.rs:551-792, driven granule by granule.
static constexpr uint32_t kSplitRowMax = kMaxStaticContexts * 256 > 704 ? kMaxStaticContexts * 256 : 704;

struct SplitScratch {
  // workgroup shared memory
  uint32_t curr[kSplitRowMax];
  uint32_t comb[2][kSplitRowMax];
  uint32_t last[2][kSplitRowMax];  // the histograms of the last and the second last block type (copies of the rows in H)
  float terms[3][kSplitRowMax];    // non-zero entropy terms of curr / comb[0] / comb[1], compacted per context row
  float entropy[3 * kMaxStaticContexts];  // [0..nc) current, [nc..2nc) combined with last, [2nc..3nc) with second last
  float last_entropy[2 * kMaxStaticContexts];
  uint32_t ctl[16];
};

// BitsEntropy (bit_cost.rs:13-42) of the 3 * nc rows {curr, comb[0], comb[1]} x contexts, cooperatively.
// The reference accumulates `retval -= p * log2(p)` strictly left to right in f32; bins with p == 0 contribute an
// exact 0.0, so leaving them out does not change any intermediate value.  Each wavefront takes a row: the lanes compute
// the terms and compact the non-zero ones (ballot + popcount), then one lane adds them up in order -- the sequential
// part shrinks from `alphabet` dependent adds to the number of symbols that actually occur.
BR_DEV void mb_rows_entropy(const EntropyTables& et, SplitScratch& S, uint32_t nc, uint32_t row, uint32_t alphabet) {
#if defined(BROTLI_HOST_EMU)
  const uint32_t lane = 0, lanes = 1, wave = 0, waves = 1;
#else
  const uint32_t lane = threadIdx.x & 63u, lanes = 64, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
#endif
  for (uint32_t r = wave; r < 3 * nc; r += waves) {
    const uint32_t w = r / nc, i = r % nc;
    const uint32_t* src = (w == 0 ? S.curr : S.comb[w - 1]) + i * row;
    float* out = S.terms[w] + i * row;
    uint32_t cnt = 0, total = 0;
    for (uint32_t base = 0; base < alphabet; base += lanes) {
      const uint32_t j = base + lane;
      const uint32_t p = j < alphabet ? src[j] : 0u;
      total += p;
      const bool nz = p != 0;
#if defined(BROTLI_HOST_EMU)
      if (nz) out[cnt] = (float)p * et.logs_16[p & 0xffffu];
      cnt += nz ? 1u : 0u;
#else
      const unsigned long long mask = __ballot(nz);
      const uint32_t before = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (nz) out[cnt + before] = (float)p * et.logs_16[p & 0xffffu];
      cnt += (uint32_t)__popcll(mask);
#endif
    }
#if !defined(BROTLI_HOST_EMU)
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    __threadfence_block();
#endif
    if (lane == 0) {
      float retval = 0.0f;
      for (uint32_t n = 0; n < cnt; ++n) retval = retval - out[n];
      if (total != 0) retval = retval + (float)total * br_fast_log2(et, total);
      const float fsum = (float)total;
      S.entropy[r] = retval < fsum ? fsum : retval;
    }
  }
}

BR_DEV void mb_item_split_chain(const MbBuffers& B, uint32_t m, uint32_t kind, SplitScratch& S) {
  const MbDesc d = B.descs[m];
  if (d.uncompressed) return;
  const uint32_t row = kRowLen[kind];
  const uint32_t nc = kind == kSplitLiteral ? d.num_contexts : 1;
  const uint32_t entropy_alphabet = kind == kSplitDistance ? 64u : row;
  const uint32_t min_block = kGranuleLen[kind];
  const float threshold = kind == kSplitLiteral ? 400.0f : (kind == kSplitCommand ? 500.0f : 100.0f);
  const uint32_t max_types = kind == kSplitLiteral && nc > 1 ? 256u / nc : 256u;
  const uint32_t n_symbols = d.n_symbols[kind];
  const uint32_t n_gran = d.n_granules[kind];
  const uint16_t* G = B.gran_hist[kind] + (size_t)d.gran_row_base[kind] * row;
  uint32_t* H = B.histo[kind] + (size_t)d.histo_base[kind] * row;
  uint8_t* types = B.block_types[kind] + d.block_base[kind];
  uint32_t* lengths = B.block_lengths[kind] + d.block_base[kind];
  uint16_t* gran_block = B.gran_block[kind] + d.granule_base[kind];
  const uint32_t max_histos = d.max_histos[kind];
  const uint32_t W = nc * row;
  if (d.simple != 0) {
    // qualities 2 / 3: one block type per kind, its histogram is that of the whole meta-block (BuildHistograms,
    // brotli_bit_stream.rs:2263-2290)
    for (uint32_t j = BR_TID; j < row; j += BR_NT) {
      uint32_t acc = 0;
      for (uint32_t q = 0; q < n_gran; ++q) acc += G[(size_t)q * row + j];
      H[j] = acc;
    }
    for (uint32_t q = BR_TID; q < n_gran; q += BR_NT) gran_block[q] = 0;
    if (BR_TID == 0) {
      types[0] = 0;
      lengths[0] = n_symbols;
      MbResult& r = B.results[m];
      r.num_types[kind] = 1;
      r.num_blocks[kind] = 1;
      r.num_histos[kind] = 1;
    }
    return;
  }

  // uniform control state (every thread keeps a copy)
  uint32_t num_blocks = 0, num_types = 0, block_size = 0, target = min_block, curr_ix = 0, merge_count = 0;
  uint32_t last_ix[2] = {0, 0};
  uint32_t block_first_granule = 0;
  for (uint32_t j = BR_TID; j < W; j += BR_NT) {
    S.curr[j] = 0;
    S.last[0][j] = 0;
    S.last[1][j] = 0;
  }
  BR_SYNC();
  uint32_t g = 0;
  for (;;) {
    bool is_final = false;
    if (g < n_gran) {
      // all the granules up to the next decision point in one go (the target grows with every merge, so homogeneous
      // data asks for hundreds at a time): independent loads, instead of one memory round trip per granule
      uint32_t m = (target - block_size) / min_block;
      if (m == 0) m = 1;
      if (m > n_gran - g) m = n_gran - g;
      const uint16_t* gr = G + (size_t)g * W;
      for (uint32_t j = BR_TID; j < W; j += BR_NT) {
        uint32_t acc = 0;
        uint32_t q = 0;
        for (; q + 4 <= m; q += 4) {
          const uint32_t a0 = gr[(size_t)q * W + j], a1 = gr[(size_t)(q + 1) * W + j], a2 = gr[(size_t)(q + 2) * W + j],
                         a3 = gr[(size_t)(q + 3) * W + j];
          acc += a0 + a1 + a2 + a3;
        }
        for (; q < m; ++q) acc += gr[(size_t)q * W + j];
        S.curr[j] += acc;
      }
      const uint64_t upto = (uint64_t)(g + m) * min_block;
      block_size += (uint32_t)((upto < n_symbols ? upto : n_symbols) - (uint64_t)g * min_block);
      g += m;
      BR_SYNC();
      if (block_size != target) continue;
    } else {
      is_final = true;
    }
    // ---------------- FinishBlock
    if (block_size < min_block) block_size = min_block;
    if (num_blocks == 0) {
      for (uint32_t i = BR_TID; i < nc; i += BR_NT) {
        const float e = br_bits_entropy(B.et, S.curr + i * row, entropy_alphabet);
        S.last_entropy[i] = e;
        S.last_entropy[nc + i] = e;
      }
      BR_SYNC();  // the entropy threads must finish reading curr before it is cleared
      for (uint32_t j = BR_TID; j < W; j += BR_NT) {
        const uint32_t c = S.curr[j];
        H[j] = c;
        S.last[0][j] = c;  // last_histogram_ix = {0, 0}: both slots refer to block type 0
        S.last[1][j] = c;
        S.curr[j] = 0;
      }
      if (BR_TID == 0) {
        lengths[0] = block_size;
        types[0] = 0;
      }
      for (uint32_t q = block_first_granule + BR_TID; q < g; q += BR_NT) gran_block[q] = 0;
      block_first_granule = g;
      num_blocks = 1;
      num_types = 1;
      curr_ix += nc;
      block_size = 0;
      BR_SYNC();
    } else if (block_size > 0) {
      for (uint32_t j = BR_TID; j < W; j += BR_NT) {
        const uint32_t c = S.curr[j];
        S.comb[0][j] = c + S.last[0][j];
        S.comb[1][j] = c + S.last[1][j];
      }
      BR_SYNC();
      mb_rows_entropy(B.et, S, nc, row, entropy_alphabet);
      BR_SYNC();
      if (BR_TID == 0) {
        float diff[2] = {0.0f, 0.0f};
        for (uint32_t i = 0; i < nc; ++i) {
          for (uint32_t j = 0; j < 2; ++j) {
            const float v = S.entropy[(1 + j) * nc + i] - S.entropy[i] - S.last_entropy[j * nc + i];
            if (kind == kSplitLiteral && nc > 1) {
              diff[j] += v;  // ContextBlockSplitter accumulates over contexts (metablock.rs:725)
            } else {
              diff[j] = v;
            }
          }
        }
        uint32_t decision;
        if (num_types < max_types && diff[0] > threshold && diff[1] > threshold) {
          decision = 0;
        } else if (diff[1] < diff[0] - 20.0f) {
          decision = 1;
        } else {
          decision = 2;
        }
        S.ctl[0] = decision;
      }
      BR_SYNC();
      const uint32_t decision = S.ctl[0];
      BR_SYNC();
      if (decision == 0) {  // new block type
        uint32_t* Hn = H + (size_t)curr_ix * row;
        const bool room = curr_ix + nc <= max_histos;
        for (uint32_t j = BR_TID; j < W; j += BR_NT) {
          const uint32_t c = S.curr[j];
          if (room) Hn[j] = c;
          S.last[1][j] = S.last[0][j];
          S.last[0][j] = c;
          S.curr[j] = 0;
        }
        if (BR_TID == 0) {
          lengths[num_blocks] = block_size;
          types[num_blocks] = (uint8_t)num_types;
          for (uint32_t i = 0; i < nc; ++i) {
            S.last_entropy[nc + i] = S.last_entropy[i];
            S.last_entropy[i] = S.entropy[i];
          }
        }
        for (uint32_t q = block_first_granule + BR_TID; q < g; q += BR_NT) gran_block[q] = (uint16_t)num_blocks;
        last_ix[1] = last_ix[0];
        last_ix[0] = (kind == kSplitLiteral && nc > 1) ? num_types * nc : (num_types & 0xffu);
        num_blocks++;
        num_types++;
        curr_ix += nc;
        block_size = 0;
        merge_count = 0;
        target = min_block;
      } else if (decision == 1) {  // back to the second last type
        const uint32_t t = last_ix[0];
        last_ix[0] = last_ix[1];
        last_ix[1] = t;
        uint32_t* Hl = H + (size_t)last_ix[0] * row;
        for (uint32_t j = BR_TID; j < W; j += BR_NT) {
          const uint32_t merged = S.comb[1][j];
          Hl[j] = merged;
          S.last[1][j] = S.last[0][j];
          S.last[0][j] = merged;
          S.curr[j] = 0;
        }
        if (BR_TID == 0) {
          lengths[num_blocks] = block_size;
          types[num_blocks] = types[num_blocks - 2];
          for (uint32_t i = 0; i < nc; ++i) {
            S.last_entropy[nc + i] = S.last_entropy[i];
            S.last_entropy[i] = S.entropy[2 * nc + i];
          }
        }
        for (uint32_t q = block_first_granule + BR_TID; q < g; q += BR_NT) gran_block[q] = (uint16_t)num_blocks;
        num_blocks++;
        block_size = 0;
        merge_count = 0;
        target = min_block;
      } else {  // merge into the last block
        uint32_t* Hl = H + (size_t)last_ix[0] * row;
        const bool single_type = num_types == 1;
        for (uint32_t j = BR_TID; j < W; j += BR_NT) {
          const uint32_t merged = S.comb[0][j];
          Hl[j] = merged;
          S.last[0][j] = merged;
          if (single_type) S.last[1][j] = merged;  // both slots still name the only block type
          S.curr[j] = 0;
        }
        if (BR_TID == 0) {
          lengths[num_blocks - 1] += block_size;
          for (uint32_t i = 0; i < nc; ++i) {
            S.last_entropy[i] = S.entropy[nc + i];
            if (num_types == 1) S.last_entropy[nc + i] = S.last_entropy[i];
          }
        }
        for (uint32_t q = block_first_granule + BR_TID; q < g; q += BR_NT) gran_block[q] = (uint16_t)(num_blocks - 1);
        block_size = 0;
        if (++merge_count > 1) target += min_block;
      }
      block_first_granule = g;
      BR_SYNC();
    }
    if (is_final) break;
  }
  if (BR_TID == 0) {
    MbResult& r = B.results[m];
    r.num_types[kind] = num_types;
    r.num_blocks[kind] = num_blocks;
    r.num_histos[kind] = num_types * nc;
  }
}

// ---- K5: one histogram -> optimised counts, code lengths, codes and its serialised tree
// core of the job on explicit buffers (the device kernel stages them in LDS): h[row] in/out, depth[row], bits[row],
// words[kTreeBitsWords] out; returns the number of header bits
// coop: the call is made by all 64 lanes of a wavefront in lock step on the same (LDS) buffers -- every lane redoes the
// same scalar work (stores of identical values, ORs of identical bits), which costs nothing extra on SIMT hardware and
// lets the O(n^2 / 64) sort inside use all lanes.
BR_DEV uint32_t mb_build_code_core(uint32_t kind, uint32_t num_distance_symbols, uint32_t* h, uint8_t* depth, uint16_t* bits,
                                   uint64_t* words, HuffmanScratch* sc, bool coop = false, uint32_t mode = kCodeOptimized) {
  const uint32_t row = kRowLen[kind];
  if (mode == kCodeFast || mode == kCodeStatic) {
    // quality 2 (metablock_fast.h)
    for (uint32_t i = 0; i < row; ++i) {
      depth[i] = 0;
      bits[i] = 0;
    }
    for (uint32_t i = 0; i < kTreeBitsWords; ++i) words[i] = 0;
    BitSink sink;
    sink.words = words;
    sink.pos = 0;
    if (mode == kCodeStatic) {
      br_store_static_code(kind, depth, bits, sink);
    } else {
      uint32_t total = 0;
      for (uint32_t i = 0; i < row; ++i) total += h[i];
      const uint32_t max_bits = kind == kSplitLiteral ? 8u : (kind == kSplitCommand ? 10u : br_log2_floor_nonzero(num_distance_symbols - 1u) + 1u);
      br_build_and_store_huffman_tree_fast(h, total, max_bits, sc, depth, bits, sink);
    }
    return (uint32_t)sink.pos;
  }
  // BrotliOptimizeHistograms (metablock.rs:1076-1108): literal 256, command 704, distance min(alphabet, 544)
  uint32_t opt_len = row;
  if (kind == kSplitDistance) opt_len = num_distance_symbols < kNumDistanceHistoSymbols ? num_distance_symbols : kNumDistanceHistoSymbols;
  BR_PHASE_CLOCK();
  if (mode == kCodeOptimized) br_optimize_huffman_counts_for_rle(opt_len, h, sc->good_for_rle, coop);
  BR_PHASE(9);
  // build_and_store_entropy_codes (brotli_bit_stream.rs:1860-1889): histogram_length = row (distance:
  // num_effective_distance_symbols), alphabet_size = 256 / 704 / num_distance_symbols
  uint32_t hist_len = row, alphabet = row;
  if (kind == kSplitDistance) {
    hist_len = num_distance_symbols < kNumDistanceHistoSymbols ? num_distance_symbols : kNumDistanceHistoSymbols;
    alphabet = num_distance_symbols;
  }
  for (uint32_t i = 0; i < row; ++i) {
    depth[i] = 0;
    bits[i] = 0;
  }
  for (uint32_t i = 0; i < kTreeBitsWords; ++i) words[i] = 0;
  BitSink sink;
  sink.words = words;
  sink.pos = 0;
  BR_PHASE(10);
  br_build_and_store_huffman_tree(h, hist_len, alphabet, sc, depth, bits, sink, coop);
  BR_PHASE(11);
  return (uint32_t)sink.pos;
}

BR_DEV void mb_item_build_code(const MbBuffers& B, uint32_t kind, uint32_t row_index, uint32_t num_distance_symbols,
                               HuffmanScratch* sc, uint32_t mode = kCodeOptimized) {
  const uint32_t row = kRowLen[kind];
  B.tree_nbits[kind][row_index] =
      mb_build_code_core(kind, num_distance_symbols, B.histo[kind] + (size_t)row_index * row, B.depth[kind] + (size_t)row_index * row,
                         B.bits[kind] + (size_t)row_index * row, B.tree_bits[kind] + (size_t)row_index * kTreeBitsWords, sc, false, mode);
}

BR_DEV void mb_append_bits(BitSink& out, const uint64_t* words, uint32_t nbits) {
  uint32_t i = 0;
  while (nbits >= 32) {
    out.put(32, (words[i >> 1] >> ((i & 1) * 32)) & 0xffffffffull);
    nbits -= 32;
    i++;
  }
  if (nbits) out.put(nbits, (words[i >> 1] >> ((i & 1) * 32)) & ((1ull << nbits) - 1));
}

// ---- K6: header of one meta-block (store_meta_block up to the entropy codes, brotli_bit_stream.rs:2074-2191)
// staging != nullptr: the bits are composed there (zeroed by the caller, e.g. in LDS) instead of in B.header_words
// what the wave of k_write_headers counts in front of the header's composition and takes over behind it
struct MbSplitPrepared {
  uint32_t histograms[3][258 + 26];  // per kind: block type codes, then block length codes
  BlockSplitCode code[3];
};
BR_DEV void mb_item_write_header(const MbBuffers& B, uint32_t m, HuffmanScratch* sc, uint64_t* staging = nullptr, MbSplitPrepared* prepared = nullptr) {
  const MbDesc d = B.descs[m];
  MbResult& r = B.results[m];
  if (d.uncompressed) {
    r.header_bits = 0;
    return;
  }
  uint64_t* words = staging ? staging : B.header_words + (size_t)m * B.header_stride;
  if (!staging)
    for (uint32_t i = 0; i < B.header_stride; ++i) words[i] = 0;
  BitSink out;
  out.words = words;
  out.pos = 0;
  br_store_compressed_meta_block_header(d.is_last != 0, d.end - d.start, out);
  BlockSplitCode own_code;
  for (uint32_t kind = 0; kind < 3; ++kind) {
    BlockSplitCode& code = prepared ? prepared->code[kind] : own_code;
    for (int i = 0; i < 258; ++i) {
      code.type_depths[i] = 0;
      code.type_bits[i] = 0;
    }
    for (int i = 0; i < 26; ++i) {
      code.length_depths[i] = 0;
      code.length_bits[i] = 0;
    }
    br_build_and_store_block_split_code(B.block_types[kind] + d.block_base[kind], B.block_lengths[kind] + d.block_base[kind],
                                        r.num_blocks[kind], r.num_types[kind], sc, &code,
                                        B.switch_bits[kind] + d.block_base[kind], B.switch_nbits[kind] + d.block_base[kind], out,
                                        prepared ? prepared->histograms[kind] : nullptr);
  }
  out.put(2, d.dist_postfix_bits);
  out.put(4, d.num_direct_distance_codes >> d.dist_postfix_bits);
  for (uint32_t i = 0; i < r.num_types[kSplitLiteral]; ++i) out.put(2, d.context_mode);
  if (d.hq) {
    // BrotliBuildMetaBlock always hands over full maps (num_types << 6 and << 2 entries): EncodeContextMap for both
    uint32_t* scratch = B.ctxmap_scratch + (size_t)m * (2 * 256 * 64);
    br_encode_context_map(B.hq_ctx_map[0] + d.hq_ctx_map_base[0], r.num_types[kSplitLiteral] << 6, r.num_histos[kSplitLiteral], scratch, sc, out);
    br_encode_context_map(B.hq_ctx_map[1] + d.hq_ctx_map_base[1], r.num_types[kSplitDistance] << 2, r.num_histos[kSplitDistance], scratch, sc, out);
  } else {
    if (d.num_contexts <= 1) {
      br_store_trivial_context_map(r.num_histos[kSplitLiteral], 6, sc, out);
    } else {
      // MapStaticContexts, metablock.rs:832-857
      uint32_t* cm = B.ctxmap_scratch + (size_t)m * (2 * 256 * 64);
      const uint32_t n = r.num_types[kSplitLiteral] << 6;
      for (uint32_t i = 0; i < r.num_types[kSplitLiteral]; ++i)
        for (uint32_t j = 0; j < 64; ++j) cm[(i << 6) + j] = i * d.num_contexts + br_static_context_map(d.context_map_id, j);
      br_encode_context_map(cm, n, r.num_histos[kSplitLiteral], cm + 256 * 64, sc, out);
    }
    br_store_trivial_context_map(r.num_histos[kSplitDistance], 2, sc, out);
  }
  for (uint32_t kind = 0; kind < 3; ++kind) {
    for (uint32_t i = 0; i < r.num_histos[kind]; ++i) {
      const uint32_t row_index = d.histo_base[kind] + i;
      mb_append_bits(out, B.tree_bits[kind] + (size_t)row_index * kTreeBitsWords, B.tree_nbits[kind][row_index]);
    }
  }
  r.header_bits = (uint32_t)out.pos;
}

BR_DEV uint32_t mb_find_by_lit(const MbBuffers& B, uint32_t i) {
  uint32_t lo = 0, hi = B.n_mb;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (B.descs[mid].lit_base <= i) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  // skip meta-blocks without literals that share the same base
  while (lo + 1 < B.n_mb && B.descs[lo].n_lits == 0 && B.descs[lo + 1].lit_base <= i) lo++;
  return lo;
}
BR_DEV uint32_t mb_find_by_cmd(const MbBuffers& B, uint32_t c) {
  uint32_t lo = 0, hi = B.n_mb;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (B.descs[mid].cmd_offset <= c) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

struct SymbolCode {
  uint64_t bits;   // block switch (if any) followed by the symbol code
  uint32_t nbits;
};

// block (index inside its meta-block's split) of local symbol `local`: last start <= local
BR_DEV uint32_t hq_block_of(const uint32_t* starts, uint32_t num_blocks, uint32_t local) {
  uint32_t lo = 0, hi = num_blocks;  // starts[lo] <= local < starts[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (starts[mid] <= local) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

// Block of symbol `local` of (meta-block m, kind), and whether the block-switch command precedes it (the first symbol of
// every block but the first).  Greedy splits (quality < 10) end on granule boundaries and are looked up through the
// granule index; the quality >= 10 splits have arbitrary lengths and are searched in block_start.
BR_DEV uint32_t mb_block_of(const MbBuffers& B, const MbDesc& d, uint32_t m, uint32_t kind, uint32_t local, bool* at_switch) {
  if (d.hq) {
    const uint32_t* starts = B.block_start[kind] + d.block_base[kind];
    const uint32_t blk = hq_block_of(starts, B.results[m].num_blocks[kind], local);
    *at_switch = blk != 0 && starts[blk] == local;
    return blk;
  }
  const uint32_t gl = kGranuleLen[kind];
  const uint32_t blk = B.gran_block[kind][d.granule_base[kind] + local / gl];
  *at_switch = false;
  if ((local % gl) == 0 && blk != 0) *at_switch = B.gran_block[kind][d.granule_base[kind] + local / gl - 1] != blk;
  return blk;
}

// CommandDistanceContext, command.rs:28-36
BR_DEV uint32_t br_distance_context(const Command& c) {
  const uint32_t r = (uint32_t)(c.cmd_prefix_ >> 6);
  const uint32_t cc = (uint32_t)(c.cmd_prefix_ & 7);
  if ((r == 0 || r == 2 || r == 4 || r == 7) && cc <= 2) return cc;
  return 3;
}

// code of literal i (global literal index), store_symbol[_with_context] brotli_bit_stream.rs:1891-1920,1980-2020
BR_DEV SymbolCode mb_literal_code(const MbBuffers& B, uint32_t i) {
  const uint32_t m = mb_find_by_lit(B, i);
  const MbDesc& d = B.descs[m];
  SymbolCode sc;
  sc.bits = 0;
  sc.nbits = 0;
  if (d.uncompressed) return sc;
  const uint32_t local = i - d.lit_base;
  bool at_switch;
  const uint32_t blk = mb_block_of(B, d, m, kSplitLiteral, local, &at_switch);
  const uint32_t type = B.block_types[kSplitLiteral][d.block_base[kSplitLiteral] + blk];
  // first symbol of a block (other than the first one) is preceded by the block switch command
  if (at_switch) {
    sc.bits = B.switch_bits[kSplitLiteral][d.block_base[kSplitLiteral] + blk];
    sc.nbits = B.switch_nbits[kSplitLiteral][d.block_base[kSplitLiteral] + blk];
  }
  const uint32_t pos = B.lit_pos[i];
  const uint32_t lit = B.text[pos];
  uint32_t histo = type;
  if (d.hq) {
    histo = B.hq_ctx_map[0][d.hq_ctx_map_base[0] + (type << 6) + mb_literal_context(B, d, pos)];
  } else if (d.num_contexts > 1) {
    histo = type * d.num_contexts + br_static_context_map(d.context_map_id, mb_literal_context(B, d, pos));
  }
  const size_t ix = ((size_t)d.histo_base[kSplitLiteral] + histo) * 256 + lit;
  sc.bits |= (uint64_t)B.bits[kSplitLiteral][ix] << sc.nbits;
  sc.nbits += B.depth[kSplitLiteral][ix];
  return sc;
}

// the command's own symbols: [block switch] command code, insert/copy extra bits
BR_DEV SymbolCode mb_command_code(const MbBuffers& B, uint32_t c, const MbDesc& d, uint32_t m) {
  SymbolCode sc;
  sc.bits = 0;
  sc.nbits = 0;
  const Command cmd = B.cmds[c];
  const uint32_t local = c - d.cmd_offset;
  bool at_switch;
  const uint32_t blk = mb_block_of(B, d, m, kSplitCommand, local, &at_switch);
  const uint32_t type = B.block_types[kSplitCommand][d.block_base[kSplitCommand] + blk];
  if (at_switch) {
    sc.bits = B.switch_bits[kSplitCommand][d.block_base[kSplitCommand] + blk];
    sc.nbits = B.switch_nbits[kSplitCommand][d.block_base[kSplitCommand] + blk];
  }
  const size_t ix = ((size_t)d.histo_base[kSplitCommand] + type) * kNumCommandSymbols + cmd.cmd_prefix_;
  sc.bits |= (uint64_t)B.bits[kSplitCommand][ix] << sc.nbits;
  sc.nbits += B.depth[kSplitCommand][ix];
  // only the LENGTH of the insert/copy extra bits is added here: switch + code + extras can exceed 64 bits,
  // the emitter writes the extras as a separate piece
  uint32_t en;
  (void)br_command_extra_bits(cmd, &en);
  sc.nbits += en;
  return sc;
}

// distance symbol of command c: [block switch] distance code, extra bits
BR_DEV SymbolCode mb_distance_code(const MbBuffers& B, uint32_t c, const MbDesc& d, uint32_t m) {
  SymbolCode sc;
  sc.bits = 0;
  sc.nbits = 0;
  const Command cmd = B.cmds[c];
  if (!br_command_has_distance(cmd)) return sc;
  const uint32_t local = B.cmd_dist_index[c] - d.dist_base;
  bool at_switch;
  const uint32_t blk = mb_block_of(B, d, m, kSplitDistance, local, &at_switch);
  const uint32_t type = B.block_types[kSplitDistance][d.block_base[kSplitDistance] + blk];
  if (at_switch) {
    sc.bits = B.switch_bits[kSplitDistance][d.block_base[kSplitDistance] + blk];
    sc.nbits = B.switch_nbits[kSplitDistance][d.block_base[kSplitDistance] + blk];
  }
  const uint32_t dist_code = cmd.dist_prefix_ & 0x3ffu;
  const uint32_t histo = d.hq ? B.hq_ctx_map[1][d.hq_ctx_map_base[1] + (type << 2) + br_distance_context(cmd)] : type;
  const size_t ix = ((size_t)d.histo_base[kSplitDistance] + histo) * kNumDistanceHistoSymbols + dist_code;
  sc.bits |= (uint64_t)B.bits[kSplitDistance][ix] << sc.nbits;
  sc.nbits += B.depth[kSplitDistance][ix];
  return sc;
}

// ---- K7: bit length of every literal
BR_DEV void mb_item_literal_nbits(const MbBuffers& B, uint32_t i) { B.lit_nbits[i] = mb_literal_code(B, i).nbits; }

// ---- K8: bit length of every command (own symbols + its literals + its distance)
BR_DEV void mb_item_command_nbits(const MbBuffers& B, uint32_t c) {
  const uint32_t m = mb_find_by_cmd(B, c);
  const MbDesc& d = B.descs[m];
  if (d.uncompressed) {
    B.cmd_nbits[c] = 0;
    B.cmd_own_bits[c] = 0;
    return;
  }
  // command code + extras can exceed 64 bits together with a block switch: count the pieces separately
  const Command cmd = B.cmds[c];
  SymbolCode own = mb_command_code(B, c, d, m);
  const SymbolCode dist = mb_distance_code(B, c, d, m);
  const uint32_t dist_extra_n = br_command_has_distance(cmd) ? (uint32_t)(cmd.dist_prefix_ >> 10) : 0;
  const uint32_t lit_bits = B.lit_nbits[B.cmd_lit_start[c + 1]] - B.lit_nbits[B.cmd_lit_start[c]];  // after the scan
  B.cmd_own_bits[c] = own.nbits;
  B.cmd_nbits[c] = own.nbits + lit_bits + dist.nbits + dist_extra_n;
}

BR_DEV void mb_put_bits_atomic(uint64_t* words, uint64_t pos, uint32_t nbits, uint64_t bits) {
  if (nbits == 0) return;
  const uint32_t sh = (uint32_t)(pos & 63u);
  uint64_t* w = words + (pos >> 6);
  BR_ATOMIC_OR_U64(w, bits << sh);
  if (sh + nbits > 64) BR_ATOMIC_OR_U64(w + 1, bits >> (64 - sh));
}

// ---- K9: emission
BR_DEV void mb_item_emit_command(const MbBuffers& B, uint32_t c) {
  const uint32_t m = mb_find_by_cmd(B, c);
  const MbDesc& d = B.descs[m];
  if (d.uncompressed) return;
  const Command cmd = B.cmds[c];
  const uint64_t base = B.mb_out_bit[m] + B.results[m].header_bits + (B.cmd_nbits[c] - B.cmd_nbits[d.cmd_offset]);
  // own symbols, written in two pieces (switch + code, then extras) to stay below 64 bits per piece
  {
    SymbolCode sc;
    sc.bits = 0;
    sc.nbits = 0;
    const uint32_t local = c - d.cmd_offset;
    bool at_switch;
    const uint32_t blk = mb_block_of(B, d, m, kSplitCommand, local, &at_switch);
    const uint32_t type = B.block_types[kSplitCommand][d.block_base[kSplitCommand] + blk];
    if (at_switch) {
      sc.bits = B.switch_bits[kSplitCommand][d.block_base[kSplitCommand] + blk];
      sc.nbits = B.switch_nbits[kSplitCommand][d.block_base[kSplitCommand] + blk];
    }
    uint64_t pos = base;
    mb_put_bits_atomic(B.out_words, pos, sc.nbits, sc.bits);
    pos += sc.nbits;
    const size_t ix = ((size_t)d.histo_base[kSplitCommand] + type) * kNumCommandSymbols + cmd.cmd_prefix_;
    mb_put_bits_atomic(B.out_words, pos, B.depth[kSplitCommand][ix], B.bits[kSplitCommand][ix]);
    pos += B.depth[kSplitCommand][ix];
    uint32_t en;
    const uint64_t eb = br_command_extra_bits(cmd, &en);
    mb_put_bits_atomic(B.out_words, pos, en, eb);
  }
  if (br_command_has_distance(cmd)) {
    const uint32_t lit_bits = B.lit_nbits[B.cmd_lit_start[c + 1]] - B.lit_nbits[B.cmd_lit_start[c]];
    uint64_t pos = base + B.cmd_own_bits[c] + lit_bits;
    const SymbolCode dc = mb_distance_code(B, c, d, m);
    mb_put_bits_atomic(B.out_words, pos, dc.nbits, dc.bits);
    pos += dc.nbits;
    mb_put_bits_atomic(B.out_words, pos, (uint32_t)(cmd.dist_prefix_ >> 10), cmd.dist_extra_);
  }
}

BR_DEV void mb_item_emit_literal(const MbBuffers& B, uint32_t i) {
  const uint32_t m = mb_find_by_lit(B, i);
  const MbDesc& d = B.descs[m];
  if (d.uncompressed) return;
  const uint32_t c = B.lit_cmd[i];
  const uint64_t pos = B.mb_out_bit[m] + B.results[m].header_bits + (B.cmd_nbits[c] - B.cmd_nbits[d.cmd_offset]) + B.cmd_own_bits[c] +
                       (B.lit_nbits[i] - B.lit_nbits[B.cmd_lit_start[c]]);
  const SymbolCode sc = mb_literal_code(B, i);
  mb_put_bits_atomic(B.out_words, pos, sc.nbits, sc.bits);
}

// ---- K10: copy `nbits` bits from a word-aligned source to an arbitrary bit offset of the output
BR_DEV void mb_item_copy_bits_word(uint64_t* out, uint64_t dst_bit, const uint64_t* src, uint64_t nbits, uint64_t w) {
  // word w of the source (64 bits, the last one may be partial)
  const uint64_t first = w * 64;
  if (first >= nbits) return;
  uint32_t n = nbits - first >= 64 ? 64u : (uint32_t)(nbits - first);
  uint64_t v = src[w];
  if (n < 64) v &= (1ull << n) - 1;
  mb_put_bits_atomic(out, dst_bit + first, n, v);
}

}  // namespace brotli_mi355x
#endif
