// lz77_stage.cpp -- see lz77_stage.h
#include "lz77_stage.h"
#include "timeline.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <stdexcept>
#include <string>

#include "host_entropy.h"

#include <cmath>

namespace brotli_mi355x {

Lz77Stage::~Lz77Stage() { Release(); }

void Lz77Stage::Release() {
  if (owns_buffers_) {
    dev_free(B_.keys);
    dev_free(B_.by_key);
    dev_free(B_.sorted_keys);
    dev_free(B_.fbits);
    dev_free(B_.key_first);
    dev_free(B_.key_last);
    dev_free(B_.changed_keys);
    dev_free(B_.changed_count);
    dev_free(B_.info[0]);
    dev_free(B_.info[1]);
    dev_free(B_.sorted[0]);
    dev_free(B_.sorted[1]);
    dev_free(B_.sorted_tag[0]);
    dev_free(B_.sorted_tag[1]);
    B_.sorted_tag[0] = B_.sorted_tag[1] = nullptr;
    dev_free(B_.search_log);
    dev_free(B_.recheck_list);
    dev_free(B_.recheck_count);
    B_.search_log = B_.recheck_list = B_.recheck_count = nullptr;
    dev_free(B_.key_base);
    dev_free(B_.stag);
    dev_free(B_.rows);
    dev_free(B_.dict_items);
    dev_free(B_.checkpoints);
    dev_free(B_.rows_changed_lo);
    dev_free(B_.rows_changed_hi);
    dev_free(B_.changed_slot);
    dev_free(B_.row_ctl);
    dev_free(B_.big_tile);
    dev_free(B_.run_end);
    dev_free(L_.num);
    dev_free(L_.buckets);
    dev_free(L_.slot_of);
    dev_free(L_.rank[0]);
    dev_free(L_.rank[1]);
    dev_free(L_.entry[0]);
    dev_free(L_.entry[1]);
    dev_free(L_.changed_key);
    dev_free(L_.state);
    L_ = LiveBuffers{};
    dev_free(count_base_dev_);
    count_base_dev_ = nullptr;
    dev_free(B_.reset_counts);
    dev_free(B_.smask);
    dev_free(B_.gprev);
    dev_free(B_.pot);
    dev_free(B_.pot_state);
    dev_free(B_.pot_list);
    dev_free(B_.flip_cells);
    dev_free(Z_.buckets);
    dev_free(Z_.forest);
    dev_free(Z_.nodes);
    dev_free(Z_.literal_costs);
    dev_free(Z_.cost_dist);
    dev_free(Z_.cost_cmd);
    dev_free(Z_.matches);
    dev_free(Z_.num_matches);
    dev_free(Z_.tmp_cmds);
    dev_free(Z_.histo);
    dev_free(Z_.forest_new);
    dev_free(Z_.forest_bak);
    dev_free(Z_.buckets_bak);
    dev_free(Z_.rerooted);
    dev_free(Z_.ctl);
    Z_ = ZopfliJob{};
    dev_free(Q_.table);
    dev_free(S_.ev_slot);
    dev_free(S_.ev_id);
    dev_free(S_.ev_of);
    dev_free(S_.slot_first);
    dev_free(S_.qrank);
    dev_free(S_.act);
    dev_free(S_.val);
    dev_free(S_.cand);
    dev_free(S_.flags);
    dev_free(S_.actraw);
    dev_free(S_.flags_prev);
    dev_free(S_.chg_list);
    dev_free(S_.chg_range);
    dev_free(S_.chg_count);
    dev_free(S_.sort_tmp);
    dev_free(S_.sort_keys_tmp);
    dev_free(S_.sort_ids_tmp);
    dev_free(S_.scan_tmp);
    S_ = QuickSpec{};
    Q_ = QuickJob{};
    dev_free(qsnap_table_);
    qsnap_table_ = nullptr;
    dev_free(zsnap_buckets_);
    dev_free(zsnap_forest_);
    zsnap_buckets_ = zsnap_forest_ = nullptr;
    dev_free(B_.flags[0]);
    dev_free(B_.flags[1]);
    dev_free(B_.cmds);
    dev_free(B_.segments);
    dev_free(B_.entries);
    dev_free(B_.exits);
    dev_free(B_.sort_tmp);
    dev_free(gathered_cmds_);
    dev_free(histo_dev_);
    dev_free(gather_offsets_dev_);
    dev_free(gather_counts_dev_);
  }
  owns_buffers_ = false;
  gathered_cmds_ = nullptr;
}

void Lz77Stage::Setup(const EncoderParams& params, uint8_t* text_dev, uint32_t prefix_bytes, uint32_t input_bytes,
                      uint32_t raw_head_bytes, uint32_t segment_bytes) {
  Release();
  coarse_blocks_.clear();
  params_ = params;
  input_bytes_ = input_bytes;
  raw_head_bytes_ = raw_head_bytes;
  block_bytes_ = 1u << params.lgblock;
  segment_bytes_ = std::min(std::max(segment_bytes, 256u), block_bytes_);
  memset(&P_, 0, sizeof(P_));
  P_.total_bytes = prefix_bytes + input_bytes;
  P_.prefix_bytes = prefix_bytes;
  P_.dict_break = (carry_ && carry_->valid) ? carry_->dict_break : prefix_bytes;
  P_.ring_mask = (1u << ComputeRbBits(params)) - 1u;
  P_.max_backward_limit = (1u << params.lgwin) - 16u;
  P_.hasher_kind = params.hasher.type == 5 ? 5 : (params.hasher.type == 9 ? 9 : 6);
  P_.bucket_bits = (uint32_t)params.hasher.bucket_bits;
  P_.block_bits = (uint32_t)params.hasher.block_bits;
  P_.hash_len = (uint32_t)params.hasher.hash_len;
  P_.ndist = (uint32_t)params.hasher.num_last_distances_to_check;
  P_.htl = P_.hasher_kind == 6 ? 8 : 4;
  P_.literal_byte_score = (uint32_t)(params.hasher.literal_byte_score ? params.hasher.literal_byte_score : 540);
  P_.score_per_byte = P_.literal_byte_score >> 2;
  P_.use_dictionary = params.use_dictionary ? 1 : 0;
  P_.spree_window = params.quality < 9 ? 64 : 512;
  P_.dist_max_distance = (uint32_t)params.dist.max_distance;
  P_.quality = (uint32_t)params.quality;
  P_.dist_postfix_bits = params.dist.distance_postfix_bits;
  P_.num_direct_distance_codes = params.dist.num_direct_distance_codes;
  P_.cmd_slab_stride = segment_bytes_ / 2 + 8;
  P_.block_bytes = block_bytes_;
  P_.max_metablock_bytes = (uint32_t)MaxMetablockSize(params);
  BuildSegments();
  P_.num_segments = (uint32_t)segments_.size();

  const size_t M = P_.total_bytes;
  B_ = Lz77Buffers{};
  B_.text = text_dev;
  use_zopfli_ = params.hasher.type == 10;
  use_quick_ = params.hasher.type == 2 || params.hasher.type == 3 || params.hasher.type == 4 || params.hasher.type == 54;
  if (use_quick_) {
    // Qualities 2 .. 4 (quick_device.h): one chain per stream on the reference's own BasicHasher table, block by block, one
    // segment per input block; none of the sort / row / rank structures of the speculative path exist.
    use_live_ = use_rows_ = false;
    P_.htl = 8;
    // The speculative path (quick_spec.h): the segments of a block side by side on candidates derived from per-position flags, the
    // resolver over their exits as for qualities 5-9.  A later piece of a stream brings the table of the piece in front as what
    // every slot holds before this text files anything (QuickSpec::base); the table behind the piece -- or, for a partial piece,
    // in front of the block that opens the meta-block still open -- is derived from the flags (ExportQuick).  The walk over the
    // reference's own table block by block (quick_device.h) stays as the fall-back for inputs that do not settle, for inputs of
    // less than 64 bytes, and as a test aid (BROTLI_MI355X_QUICK_SERIAL=1).
    use_qspec_ = getenv("BROTLI_MI355X_QUICK_SERIAL") == nullptr && input_bytes >= 64 && (uint64_t)P_.total_bytes * 2 < 0xfffffff0ull;
    qspec_books_in_ = false;
    qspec_coarse_ = false;
    qspec_params_ = params;
    qspec_text_ = text_dev;
    qspec_prefix_ = prefix_bytes;
    qspec_input_ = input_bytes;
    qspec_raw_head_ = raw_head_bytes;
    if (!use_qspec_ && segment_bytes_ != block_bytes_) segment_bytes_ = block_bytes_;
    P_.cmd_slab_stride = segment_bytes_ / 2 + 8;
    BuildSegments();
    P_.num_segments = (uint32_t)segments_.size();
    Q_ = QuickJob{};
    Q_.kind = (uint32_t)params.hasher.type;
    Q_.bucket_bits = Q_.kind == 54 ? 20 : (Q_.kind == 4 ? 17 : 16);
    Q_.sweep = Q_.kind == 2 ? 1 : (Q_.kind == 3 ? 2 : 4);
    Q_.hash_len = Q_.kind == 54 ? 7 : 5;
    Q_.use_dictionary = (params.use_dictionary && (Q_.kind == 2 || Q_.kind == 4)) ? 1 : 0;
    // serial path: the throttle books travel with the table, not with the resolver; speculative path: the resolver keeps them
    P_.use_dictionary = use_qspec_ ? Q_.use_dictionary : 0;
    Q_.table = (uint32_t*)dev_alloc_uninit((size_t)quick_table_words(Q_) * 4 + 64);
    if (use_qspec_) {
      S_ = QuickSpec{};
      S_.n = P_.total_bytes;
      S_.events = Q_.sweep == 1 ? S_.n : 2u * S_.n;
      S_.slots = quick_slots(Q_);
      const size_t E = S_.events, N = S_.n;
      S_.ev_slot = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.ev_id = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.ev_of = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.slot_first = (uint32_t*)dev_alloc_uninit(((size_t)S_.slots + 2) * 4 + 64);
      S_.qrank = (uint32_t*)dev_alloc_uninit(N * Q_.sweep * 4 + 64);
      S_.actraw = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.act = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.val = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.cand = (uint32_t*)dev_alloc_uninit(N * Q_.sweep * 4 + 64);
      S_.flags = (uint8_t*)dev_alloc_uninit(N + 64);
      S_.flags_prev = (uint8_t*)dev_alloc_uninit(N + 64);
      S_.chg_cap = (uint32_t)std::max<size_t>(4096, E / (getenv("BROTLI_MI355X_QUICK_CAP_DIV") ? (size_t)atoi(getenv("BROTLI_MI355X_QUICK_CAP_DIV")) : 16));
      S_.chg_list = (uint32_t*)dev_alloc_uninit((size_t)S_.chg_cap * 4 + 64);
      S_.chg_range = (uint32_t*)dev_alloc_uninit((size_t)S_.chg_cap * 12 + 64);
      S_.chg_count = (uint32_t*)dev_alloc(64);
      S_.sort_tmp_bytes = lz77_qspec_sort_tmp_bytes(S_.events);
      S_.sort_tmp = dev_alloc_uninit(S_.sort_tmp_bytes);
      S_.sort_keys_tmp = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.sort_ids_tmp = (uint32_t*)dev_alloc_uninit(E * 4 + 64);
      S_.scan_tmp = (uint32_t*)dev_alloc_uninit((E / 1024 + E / (1024 * 1024) + 8192) * 4);
    }
    cmds_bytes_ = (size_t)total_cmd_slots_ * sizeof(Command) + 64;
    B_.cmds = (Command*)dev_alloc_uninit(cmds_bytes_);
    B_.segments = (Segment*)dev_alloc(segments_.size() * sizeof(Segment) + 64);
    B_.entries = (SegEntry*)dev_alloc(segments_.size() * sizeof(SegEntry) + 64);
    B_.exits = (SegExit*)dev_alloc(segments_.size() * sizeof(SegExit) + 64);
    histo_dev_ = (uint32_t*)dev_alloc(256 * 4);
    gather_offsets_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
    gather_counts_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
    owns_buffers_ = true;
    segments_upload_.resize_discard(segments_.size());
    memcpy(segments_upload_.data(), segments_.data(), segments_.size() * sizeof(Segment));
    dev_h2d(B_.segments, segments_upload_.data(), segments_.size() * sizeof(Segment));
    exits_.assign((uint32_t)segments_.size(), SegExit{});
    return;
  }
  if (use_zopfli_) {
    // Qualities 10 / 11 (zopfli_device.h): the H10 trees and the shortest-path parse go block by block, one segment per input
    // block; none of the sort / row / rank structures of the greedy path exist.
    use_live_ = use_rows_ = false;
    if (segment_bytes_ != block_bytes_) {
      segment_bytes_ = block_bytes_;
      P_.cmd_slab_stride = segment_bytes_ / 2 + 8;
      BuildSegments();
      P_.num_segments = (uint32_t)segments_.size();
    }
    P_.use_dictionary = 0;  // (no throttle books for the resolver to keep: H10 consults the dictionary at every position)
    // the positions sorted by a 16-bit hash key (two H10 keys -- 17 bits of the same product -- per group): the groups of a block
    // find their matches side by side (br_zopfli_matches_of_group)
    P_.hasher_kind = 5;
    P_.bucket_bits = 16;
    P_.htl = 4;
    B_.keys = (uint16_t*)dev_alloc_uninit(M * 2 + 64);
    B_.by_key = (uint32_t*)dev_alloc_uninit(M * 4 + 64);
    B_.sorted_keys = (uint16_t*)dev_alloc_uninit(M * 2 + 64);
    B_.key_first = (uint32_t*)dev_alloc((65536 + 1) * 4);
    B_.key_last = (uint32_t*)dev_alloc((65536 + 1) * 4);
    B_.changed_count = (uint32_t*)dev_alloc(64);
    B_.sort_tmp_bytes = lz77_sort_tmp_bytes(P_.total_bytes);
    B_.sort_tmp = dev_alloc_uninit(B_.sort_tmp_bytes);
    Z_ = ZopfliJob{};
    Z_.quality = (uint32_t)params.quality;
    Z_.lgwin = (uint32_t)params.lgwin;
    Z_.use_dictionary = params.use_dictionary ? 1 : 0;
    Z_.dist_alphabet_size = params.dist.alphabet_size;
    Z_.block_bytes = block_bytes_;
    Z_.buckets = (uint32_t*)dev_alloc_uninit(((size_t)1 << 17) * 4 + 64);
    Z_.forest = (uint32_t*)dev_alloc_uninit(((size_t)2 << params.lgwin) * 4 + 64);
    Z_.nodes = dev_alloc_uninit(((size_t)block_bytes_ + 2) * 20 + 64);
    Z_.literal_costs = (float*)dev_alloc_uninit(((size_t)block_bytes_ + 4) * 4 + 64);
    Z_.cost_dist = (float*)dev_alloc((size_t)(params.dist.alphabet_size + 64) * 4 + 64);
    Z_.cost_cmd = (float*)dev_alloc(704 * 4 + 64);
    Z_.histo = (uint32_t*)dev_alloc(2048 * 4 + 64);
    Z_.matches = (unsigned long long*)dev_alloc_uninit((size_t)128 * block_bytes_ * 8 + 64);
    Z_.num_matches = (uint32_t*)dev_alloc((size_t)block_bytes_ * 4 + 64);
    Z_.tmp_cmds = (Command*)dev_alloc_uninit(((size_t)block_bytes_ / 2 + 8) * sizeof(Command) + 64);
    Z_.forest_new = (uint32_t*)dev_alloc_uninit(((size_t)2 << params.lgwin) * 4 + 64);
    Z_.forest_bak = (uint32_t*)dev_alloc_uninit(((size_t)2 << params.lgwin) * 4 + 64);
    Z_.buckets_bak = (uint32_t*)dev_alloc_uninit(((size_t)1 << 17) * 4 + 64);
    Z_.rerooted = (uint8_t*)dev_alloc((size_t)block_bytes_ + 64);
    Z_.ctl = (uint32_t*)dev_alloc(64);
    cmds_bytes_ = (size_t)total_cmd_slots_ * sizeof(Command) + 64;
    B_.cmds = (Command*)dev_alloc_uninit(cmds_bytes_);
    B_.segments = (Segment*)dev_alloc(segments_.size() * sizeof(Segment) + 64);
    B_.entries = (SegEntry*)dev_alloc(segments_.size() * sizeof(SegEntry) + 64);
    B_.exits = (SegExit*)dev_alloc(segments_.size() * sizeof(SegExit) + 64);
    histo_dev_ = (uint32_t*)dev_alloc(256 * 4);
    gather_offsets_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
    gather_counts_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
    owns_buffers_ = true;
    segments_upload_.resize_discard(segments_.size());
    memcpy(segments_upload_.data(), segments_.data(), segments_.size() * sizeof(Segment));
    dev_h2d(B_.segments, segments_upload_.data(), segments_.size() * sizeof(Segment));
    exits_.assign((uint32_t)segments_.size(), SegExit{});
    return;
  }
  B_.keys = (uint16_t*)dev_alloc_uninit(M * 2 + 64);
  B_.by_key = (uint32_t*)dev_alloc_uninit(M * 4 + 64);
  B_.sorted_keys = (uint16_t*)dev_alloc_uninit(M * 2 + 64);
  B_.fbits = (uint8_t*)dev_alloc_uninit(M + 64);
  B_.key_first = (uint32_t*)dev_alloc((65536 + 1) * 4);
  B_.key_last = (uint32_t*)dev_alloc((65536 + 1) * 4);
  B_.changed_count = (uint32_t*)dev_alloc(64);
  // Ring depth 16 (quality 5): position-indexed candidate rows (lz77_chain.h); deeper rings keep the rank structures.
  substitute_inherited_pushes_ = getenv("BROTLI_MI355X_PUSH_SUBSTITUTION") != nullptr;
  use_rows_ = P_.hasher_kind != 9 && (1u << P_.block_bits) <= kRowEntries && getenv("BROTLI_MI355X_NO_ROWS") == nullptr;
  // Masked H5 ring entries (AdvHasher::StoreRangeOptBatch, mod.rs:1163-1232): from the stream position at which they start
  // to exist -- one ring-buffer size -- the candidates of a search hang on the exact parse in front of it, and the text is
  // parsed by live chains, one per input block, each on a private copy of the reference's bucket rings (lz77_live.h).
  // BROTLI_MI355X_LIVE=1 runs them on any AdvHasher input (test aid).
  P_.masked_from = kNeverMasked;
  if (P_.hasher_kind == 5) {
    const uint64_t ring = (uint64_t)P_.ring_mask + 1;
    const uint64_t base = (carry_ && carry_->valid) ? carry_->stream_base : 0;  // stream position of text position 0
    if (base >= ring) P_.masked_from = 0;
    else if (base + P_.total_bytes > ring) P_.masked_from = (uint32_t)(ring - base);
  }
  use_live_ = P_.hasher_kind != 9 && (P_.masked_from != kNeverMasked || getenv("BROTLI_MI355X_LIVE") != nullptr);
  if (use_live_) {
    use_rows_ = false;
    if (segment_bytes_ != block_bytes_) {  // one chain per input block
      segment_bytes_ = block_bytes_;
      P_.cmd_slab_stride = segment_bytes_ / 2 + 8;
      BuildSegments();
      P_.num_segments = (uint32_t)segments_.size();
    }
    // One chain walks through the whole text.  (Chains over spans of blocks, each on rings materialised from the flags of
    // the round before, were built and measured first: the parse heals from a wrong history within ~100 KB, but every change
    // of a flag has a small chance to change a search result up to a window further on, and with masked entries and a
    // 4 MiB window that cascade does not die out -- every span behind the frontier stayed dirty, one span settled per
    // round.  DESIGN.md section 10.)
    const size_t K = (size_t)1 << P_.bucket_bits;
    L_.span_blocks = (uint32_t)std::max<size_t>(1, segments_.size());
    L_.tables = 1;
    L_.num = (uint16_t*)dev_alloc_uninit(K * 2 + 64);
    L_.buckets = (uint32_t*)dev_alloc_uninit((K << P_.block_bits) * 4 + 64);
    L_.state = (LiveBlockState*)dev_alloc(segments_.size() * sizeof(LiveBlockState) + 64);
    live_verify_ = getenv("BROTLI_MI355X_LIVE_VERIFY") != nullptr || getenv("BROTLI_MI355X_SELFTEST") != nullptr;
    L_.slot_of = (uint32_t*)dev_alloc_uninit(M * 4 + 64);
    for (int i = 0; i < 2; ++i) {
      L_.rank[i] = (uint32_t*)dev_alloc_uninit((M + 1) * 4 + 64);
      L_.entry[i] = (uint32_t*)dev_alloc_uninit(M * 4 + 64);
    }
    if (live_verify_) {
      // every search is logged; the verification repeats them one by one against the final flags (lz77_live_verify)
      B_.search_log = (uint32_t*)dev_alloc(M * kSearchLogWords * 4 + 64);
      B_.recheck_cap = (uint32_t)(M + 4096);  // (every position can be a searched one)
      B_.recheck_list = (uint32_t*)dev_alloc_uninit((size_t)B_.recheck_cap * 4 + 64);
      B_.recheck_count = (uint32_t*)dev_alloc(64);
    }
    L_.changed_key = (uint8_t*)dev_alloc(65536 + 64);
    B_.changed_cap = kChangedCap;
    B_.changed_keys = (uint32_t*)dev_alloc((size_t)kChangedCap * 4);
  } else
  if (use_rows_) {
    B_.changed_cap = (uint32_t)std::max<size_t>(kChangedCap, M / 32);
    B_.changed_keys = (uint32_t*)dev_alloc_uninit((size_t)B_.changed_cap * 4 + 64);
    B_.changed_slot = (uint32_t*)dev_alloc_uninit((size_t)B_.changed_cap * 4 + 64);
    B_.row_ctl = (uint32_t*)dev_alloc(64);
    B_.big_tile = (uint8_t*)dev_alloc(M / 1024 + 128);
    B_.smask = (unsigned long long*)dev_alloc_uninit((M / 64 + 2) * 8 + 64);
    B_.gprev = (uint32_t*)dev_alloc_uninit((M / 64 + 2) * 4 + 64);
    if (getenv("BROTLI_MI355X_NO_POTENTIAL_MASK") == nullptr) {
      B_.pot = (unsigned long long*)dev_alloc_uninit((M / 64 + 2) * 8 + 64);
      B_.pot_state = (uint32_t*)dev_alloc(64);
      B_.pot_list_cap = (uint32_t)(M / 16 + 1024);  // (denser than that: the pass over all rows is the cheaper one)
      B_.pot_list = (uint32_t*)dev_alloc_uninit((size_t)B_.pot_list_cap * 4 + 64);
      if (getenv("BROTLI_MI355X_NO_FLIP_CELLS") == nullptr) {
        uint32_t shift = 8;
        while (shift < 31 && (1u << shift) < P_.max_backward_limit + 1u) ++shift;
        while (shift < 31 && (M >> shift) + 2 > 1024) ++shift;
        B_.cell_shift = shift;
        B_.cells_per_key = (uint32_t)(M >> shift) + 2;
        B_.flip_cells = (uint32_t*)dev_alloc((size_t)65536 * B_.cells_per_key / 8 + 64);
      }
    }
    B_.stag = (uint16_t*)dev_alloc_uninit(M * 2 + 64);
    B_.rows = (uint32_t*)dev_alloc_uninit(M * kRowEntries * 4 + 64);
    if (P_.use_dictionary) B_.dict_items = (uint32_t*)dev_alloc_uninit(M * 4 + 256);
    if (getenv("BROTLI_MI355X_NO_CHECKPOINTS") == nullptr) {
      B_.checkpoints = dev_alloc((M / kCheckpointStride + 2) * sizeof(Checkpoint));  // (zero: no record is valid)
      B_.rows_changed_lo = (uint32_t*)dev_alloc_uninit(segments_.size() * 4 + 64);
      B_.rows_changed_hi = (uint32_t*)dev_alloc_uninit(segments_.size() * 4 + 64);
    }
  } else {
    B_.changed_cap = kChangedCap;
    B_.changed_keys = (uint32_t*)dev_alloc((size_t)kChangedCap * 4);
    B_.info[0] = (uint32_t*)dev_alloc_uninit(M * 8 + 64);
    B_.info[1] = (uint32_t*)dev_alloc_uninit(M * 8 + 64);
    B_.sorted[0] = (uint32_t*)dev_alloc(M * 4 + 64);
    B_.sorted[1] = (uint32_t*)dev_alloc(M * 4 + 64);
    // tags of the ring entries (ChainTables::sorted_tag): candidates that start with other bytes are not fetched
    if (getenv("BROTLI_MI355X_NO_TAGS") == nullptr) {
      B_.stag = (uint16_t*)dev_alloc_uninit(M * 2 + 64);
      B_.sorted_tag[0] = (uint16_t*)dev_alloc(M * 2 + 64);
      B_.sorted_tag[1] = (uint16_t*)dev_alloc(M * 2 + 64);
    }
    // every search is logged so that a flag change can be answered by repeating single searches (lz77_recheck_searches)
    if (getenv("BROTLI_MI355X_NO_RECHECK") == nullptr) {
      B_.search_log = (uint32_t*)dev_alloc(M * kSearchLogWords * 4 + 64);
      B_.recheck_cap = (uint32_t)std::max<size_t>(4096, M / 8);
      B_.recheck_list = (uint32_t*)dev_alloc_uninit((size_t)B_.recheck_cap * 4 + 64);
      B_.recheck_count = (uint32_t*)dev_alloc(64);
    }
  }
  B_.key_base = (uint32_t*)dev_alloc((65536 + 1) * 4);
  B_.reset_counts = (uint32_t*)dev_alloc((65536 + 1) * 4);
  B_.flags[0] = (uint8_t*)dev_alloc(M + 64);
  B_.flags[1] = (uint8_t*)dev_alloc(M + 64);
  cmds_bytes_ = (size_t)total_cmd_slots_ * sizeof(Command) + 64;
  B_.cmds = (Command*)dev_alloc_uninit(cmds_bytes_);
  B_.segments = (Segment*)dev_alloc(segments_.size() * sizeof(Segment) + 64);
  B_.entries = (SegEntry*)dev_alloc(segments_.size() * sizeof(SegEntry) + 64);
  B_.exits = (SegExit*)dev_alloc(segments_.size() * sizeof(SegExit) + 64);
  B_.sort_tmp_bytes = lz77_sort_tmp_bytes(P_.total_bytes);
  B_.sort_tmp = dev_alloc_uninit(B_.sort_tmp_bytes);  // (every user writes what it reads: histograms, tile sums, ping-pong arrays)
  histo_dev_ = (uint32_t*)dev_alloc(256 * 4);
  gather_offsets_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
  gather_counts_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
  owns_buffers_ = true;
  segments_upload_.resize_discard(segments_.size());
  memcpy(segments_upload_.data(), segments_.data(), segments_.size() * sizeof(Segment));
  dev_h2d(B_.segments, segments_upload_.data(), segments_.size() * sizeof(Segment));
}

void Lz77Stage::BuildSegments() {
  segments_.clear();
  const uint32_t P0 = P_.prefix_bytes;
  const uint32_t end = P_.total_bytes;
  const uint32_t htl = P_.htl;
  // block geometry: compress_stream hands the encoder at most one full input block at a time
  // (encode.rs:2940-2964); the catable raw head belongs to the first block but is not searched.
  std::vector<std::pair<uint32_t, uint32_t>> blocks;
  for (uint64_t off = 0; off < input_bytes_; off += block_bytes_) {
    uint32_t bs = P0 + (uint32_t)off;
    uint32_t be = (uint32_t)std::min<uint64_t>(end, (uint64_t)P0 + off + block_bytes_);
    if (off == 0) bs = std::min(be, bs + raw_head_bytes_);
    blocks.emplace_back(bs, be);
  }
  uint32_t cmd_base = 0;
  block_first_segment_.clear();
  block_segment_bytes_.clear();
  for (size_t b = 0; b < blocks.size(); ++b) {
    const uint32_t bs = blocks[b].first, be = blocks[b].second;
    bool stitched = false;
    if (b + 1 < blocks.size()) {
      const uint32_t nbytes = blocks[b + 1].second - blocks[b + 1].first;
      stitched = nbytes >= htl - 1 && blocks[b + 1].first >= 3;  // StitchToPreviousBlockInternal, mod.rs:210-222
    }
    // a block can be marked for a single chain (coarse_blocks_, see RunRounds)
    const uint32_t seg_bytes = (b < coarse_blocks_.size() && coarse_blocks_[b]) ? block_bytes_ : segment_bytes_;
    block_first_segment_.push_back((uint32_t)segments_.size());
    block_segment_bytes_.push_back(seg_bytes);
    uint32_t s = bs;
    bool first = true;
    do {
      const uint32_t e = std::min(be, s + seg_bytes);
      Segment g;
      g.start = s;
      g.end = e;
      g.blk_start = bs;
      g.blk_end = be;
      g.flags = (first ? kSegFirstInBlock : 0u) | (e == be ? kSegLastInBlock : 0u) | (stitched ? kSegTailStitched : 0u);
      g.cmd_base = cmd_base;
      g.block_index = (uint32_t)b;
      g.cmd_cap = (e - s) / 2 + 8;
      cmd_base += g.cmd_cap;
      segments_.push_back(g);
      first = false;
      s = e;
    } while (s < be);
  }
  block_first_segment_.push_back((uint32_t)segments_.size());
  total_cmd_slots_ = cmd_base;
}

void Lz77Stage::InitFlags() {
  const bool cont = carry_ && carry_->valid;
  lz77_init_flags(P_, B_, segments_.empty() ? P_.total_bytes : segments_[0].blk_start, cont ? carry_->stored.data() : nullptr,
                  cont ? (uint32_t)std::min<size_t>(carry_->stored.size(), P_.prefix_bytes) : 0u);
}

// Static-dictionary throttle state (mod.rs:1957-1960) tracked along the segments.  While the dictionary is
// alive the exact lookup / match counters are known (alive chains report their deltas); once a chain turns it
// off under the exact counters it stays off for the rest of the stream, and the chains after it only need to
// know "dead" (they never consult it, so the counter values no longer matter).
struct DictTracker {
  // kAlive: on, exact counters.  kFuzzy: on as far as known, but a chain in front could not be accounted for exactly
  // (it has to be redone): counters are approximate, `slack` bounds the error.  kDead: off for good (exact).
  // kUnknown: probably off -- the chain that seems to trip the throttle has to be redone with exact counters first.
  enum State { kAlive, kFuzzy, kDead, kUnknown };
  bool use = false;
  State state = kAlive;
  uint32_t L = 0, M = 0;
  int64_t slack = 0;
  uint32_t left_alive_at = 0xffffffffu;  // segment at which the dictionary (probably) went off
  uint32_t flips = 0;        // chains behind that point that ran with a live dictionary and must be redone
  static constexpr uint32_t kDeadL = 0x40000000u, kDeadM = 0;
  // first guess for chains whose true counters are not known yet: alive with a margin no chain can use up, so
  // that the chain reports its own deficit (dict_maxdef) instead of tripping over a made-up start of stream
  static constexpr uint32_t kAliveL = 0, kAliveM = 0x01000000u;
  static constexpr int64_t kSlackPerChain = 256;
  void Hint(SegEntry* e) const {
    switch (state) {
      case kAlive:
        e->dict_lookups = L;
        e->dict_matches = M;
        e->dict_exact = 1;
        break;
      case kFuzzy:
        e->dict_lookups = kAliveL;
        e->dict_matches = kAliveM;
        e->dict_exact = 0;
        break;
      default:
        e->dict_lookups = kDeadL;
        e->dict_matches = kDeadM;
        e->dict_exact = state == kDead;
        break;
    }
  }
  // consumes the exit of segment k, parsed with entry `used`; returns whether that parse is valid here.
  // A chain that never got (mode 1) or never could have got (mode 2, see DictState in lz77_chain.h) a dictionary
  // match parses the same whether the dictionary is on or off, which makes most chains valid in either regime.
  bool Consume(uint32_t k, const SegEntry& used, const SegExit& x) {
    if (!use) return true;
    // the exact counters may already say "off" when the last lookup of the chain in front tipped the balance and
    // nothing was consulted after it: from here on the dictionary is off for good (mod.rs:1957-1960)
    if (state == kAlive && M < (L >> 7)) {
      state = kDead;
      left_alive_at = k;
    }
    const bool no_match = x.dict_matches == used.dict_matches;
    if (state == kDead || state == kUnknown) {
      const bool ok = x.dict_mode == 0 || x.dict_mode == 2 || x.dict_mode == 4 || (x.dict_mode == 1 && no_match);
      if (!ok) flips++;
      return ok;
    }
    const bool exact = state == kAlive && used.dict_lookups == L && used.dict_matches == M;
    const bool stays_alive = 128ll * (int64_t)M - (int64_t)L + 127 - slack >= (int64_t)x.dict_maxdef;
    bool ok = true;
    State next = state;
    switch (x.dict_mode) {
      case 0:
        break;
      case 4:  // parsed blind under an "off" that does not hold here (any more): redo it with the books open
        ok = false;
        next = kFuzzy;
        slack += kSlackPerChain;
        break;
      case 1:  // found the dictionary on at every consult under the counters it was given
        if (exact || stays_alive) {
          L += x.dict_lookups - used.dict_lookups;
          M += x.dict_matches - used.dict_matches;
        } else if (no_match) {
          // under the true counters it (probably, if they are approximate) dies inside this chain, whose parse does
          // not depend on it
          next = state == kAlive ? kDead : kUnknown;
        } else {
          ok = false;
          next = kUnknown;
        }
        break;
      case 2:  // ran with the dictionary off (a guess); the exit carries the virtual bookkeeping
        if (no_match) {
          if (stays_alive) {
            L += x.dict_lookups - used.dict_lookups;
          } else {
            next = state == kAlive ? kDead : kUnknown;
          }
        } else {
          // the dictionary is on here and this chain would have used it: redo it; the chains behind it are judged
          // with approximate counters meanwhile
          ok = false;
          next = kFuzzy;
          slack += kSlackPerChain;
        }
        break;
      default:  // switched off inside the chain under its own counters: only meaningful when those were exact
        if (exact) {
          next = kDead;
        } else {
          ok = false;
          next = kUnknown;
        }
        break;
    }
    if ((!ok || next != state) && getenv("BROTLI_MI355X_DEBUG_DICT"))
      fprintf(stderr, "  dict seg %u: state %d->%d L %u M %u slack %lld used (%u,%u,%u) exit (%u,%u) mode %u maxdef %d ok %d\n", k, (int)state,
              (int)next, L, M, (long long)slack, used.dict_lookups, used.dict_matches, used.dict_exact, x.dict_lookups, x.dict_matches,
              x.dict_mode, x.dict_maxdef, (int)ok);
    if (next != state) {
      if (next == kDead || next == kUnknown) left_alive_at = k;
      state = next;
    }
    return ok;
  }
};

// Chains the exits of the last parse into the entries of the next one, replaying the per-block
// control flow of encode_data (encode.rs:2214-2543).  Returns true when every entry that the last
// parse used equals the entry derived here (i.e. the parse is self-consistent).
// Where a chain that finds no match at all leaves segment `seg` when it enters it in state `e`: the stepping of the
// parse loop (br_parse_segment) through a literal spree -- every position, then every 9th, then every 17th
// (mod.rs:2529-2546) -- is plain arithmetic on (position, apply).  Used by the resolver to carry a corrected phase
// through a whole stretch of incompressible data in one pass; like every entry guess it is verified by the re-parse.
// (the loop of br_parse_segment, step by step: the definition.  PredictLiteralRun below takes the strides in closed form --
// incompressible input asks for every segment, dozens of times -- and BROTLI_MI355X_SELFTEST compares the two.)
static void PredictLiteralRunStepwise(const Lz77Params& P, const Segment& seg, const SegEntry& e, SegExit* x) {
  const uint32_t pos_end = seg.blk_end, htl = P.htl, window = P.spree_window;
  const uint32_t margin = htl - 1 > 4 ? htl - 1 : 4;
  uint32_t position = e.pos;
  const uint32_t apply = e.apply;
  uint32_t tail_kind = e.head_kind, tail_base = e.head_base;
  const uint32_t tail_p1 = e.head_p1;
  while (position + htl < pos_end && position < seg.end) {
    position++;
    if (position > apply) {
      if (position + 16 >= pos_end - margin) {
        tail_kind = kHeadUnstored;
        tail_base = position;
        position = pos_end;
      } else if (position > apply + 4 * window) {
        tail_kind = kHeadVec4;
        tail_base = position;
        position += 16;
      } else {
        tail_kind = kHeadEven4;
        tail_base = position;
        position += 8;
      }
    }
  }
  if (seg.flags & kSegLastInBlock) position = pos_end;
  x->pos = position;
  x->apply = apply;
  x->insert_len = position - e.pos;
  x->tail_kind = position > seg.end ? tail_kind : (uint32_t)kHeadNone;
  x->tail_base = position > seg.end ? tail_base : 0u;
  x->tail_p1 = position > seg.end ? tail_p1 : 0u;
}

static void PredictLiteralRun(const Lz77Params& P, const Segment& seg, const SegEntry& e, SegExit* x) {
  const uint64_t pos_end = seg.blk_end, htl = P.htl, window = P.spree_window;
  const uint64_t margin = htl - 1 > 4 ? htl - 1 : 4;
  uint64_t position = e.pos;
  const uint64_t apply = e.apply;
  uint32_t tail_kind = e.head_kind;
  uint64_t tail_base = e.head_base;
  // the loop runs while position < stop
  const uint64_t stop = std::min<uint64_t>(pos_end > htl ? pos_end - htl : 0, seg.end);
  // one position at a time up to `apply`
  if (position < stop && position < apply) position = std::min(stop, apply);
  // strides: a step from p looks at p' = p + 1.  p' + 16 >= pos_end - margin ends the block; p' <= apply + 4 * window
  // jumps 8 on (stride 9), anything later 16 (stride 17).
  const uint64_t block_tail = pos_end > margin + 16 ? pos_end - margin - 16 : 0;  // p' >= block_tail: the end-of-block step
  auto stride = [&](uint64_t step, uint64_t p_limit, uint32_t kind) {
    // steps from positions p = position, position + step, ... while p < stop, p + 1 < block_tail and p + 1 <= p_limit
    uint64_t bound = stop;                                         // p < bound
    bound = std::min(bound, block_tail > 0 ? block_tail - 1 : 0);  // p + 1 < block_tail
    bound = std::min(bound, p_limit);                              // p + 1 <= p_limit  <=>  p < p_limit
    if (position >= bound) return;
    const uint64_t n = (bound - position + step - 1) / step;
    tail_kind = kind;
    tail_base = position + (n - 1) * step + 1;
    position += n * step;
  };
  if (position < stop) stride(9, apply + 4 * window, kHeadEven4);
  if (position < stop && position + 1 > apply + 4 * window) stride(17, ~0ull >> 1, kHeadVec4);
  if (position < stop && position + 1 >= block_tail && position + 1 > apply) {
    tail_kind = kHeadUnstored;
    tail_base = position + 1;
    position = pos_end;
  }
  if (seg.flags & kSegLastInBlock) position = pos_end;
  x->pos = (uint32_t)position;
  x->apply = (uint32_t)apply;
  x->insert_len = (uint32_t)position - e.pos;
  x->tail_kind = position > seg.end ? tail_kind : (uint32_t)kHeadNone;
  x->tail_base = position > seg.end ? (uint32_t)tail_base : 0u;
  x->tail_p1 = position > seg.end ? e.head_p1 : 0u;
  static const bool selftest = getenv("BROTLI_MI355X_SELFTEST") != nullptr || getenv("BROTLI_MI355X_SELFTEST_RESOLVE") != nullptr;
  if (selftest) {
    SegExit y = *x;
    PredictLiteralRunStepwise(P, seg, e, &y);
    if (y.pos != x->pos || y.apply != x->apply || y.insert_len != x->insert_len || y.tail_kind != x->tail_kind || y.tail_base != x->tail_base ||
        y.tail_p1 != x->tail_p1)
      throw std::runtime_error("selftest: PredictLiteralRun differs from the stepwise loop (segment " + std::to_string(seg.start) + ", entry " +
                               std::to_string(e.pos) + " / " + std::to_string(e.apply) + ")");
  }
}

// should_compress (encode.rs:1325-1354) samples every 13th byte of a literal-only meta-block.  Incompressible input has
// hundreds of such meta-blocks and the resolver meets them in the middle of its pass: the pass notes the meta-blocks it
// has no answer for yet (wanted_histograms_), goes on with a provisional answer, and Resolve() then asks the device for
// all of them in one launch and one copy and runs the pass again -- where a meta-block starts and ends hangs on the
// counts of commands and literals alone, not on the answers, so the second pass meets the same meta-blocks.
// returns whether every provisional answer of the pass (wanted_guesses_) was the right one
bool Lz77Stage::FetchShouldCompress() {
  const uint32_t count = (uint32_t)wanted_histograms_.size();
  if (count == 0) return true;
  bool guessed_right = true;
  std::vector<uint32_t> ranges((size_t)count * 2);
  for (uint32_t i = 0; i < count; ++i) {
    ranges[2 * i] = wanted_histograms_[i].first;
    ranges[2 * i + 1] = wanted_histograms_[i].second;
  }
  std::vector<uint32_t> histos((size_t)count * 256);
  lz77_sample_histograms(B_.text, ranges.data(), count, histos.data());
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t bytes = wanted_histograms_[i].second;
    const float threshold = (float)bytes * 7.92f / 13.0f;
    const bool c = !(HostBitsEntropy(histos.data() + (size_t)i * 256, 256) > threshold);
    should_compress_cache_.emplace(wanted_histograms_[i], c);
    guessed_right = guessed_right && i < wanted_guesses_.size() && (wanted_guesses_[i] != 0) == c;
    if (getenv("BROTLI_MI355X_DEBUG")) fprintf(stderr, "  should_compress sampled [%u,+%u) -> %d\n", wanted_histograms_[i].first, bytes, (int)c);
  }
  wanted_histograms_.clear();
  return guessed_right;
}

bool Lz77Stage::Resolve(bool final_pass) {
  wanted_histograms_.clear();
  wanted_guesses_.clear();
  should_compress_guess_ = true;
  static const bool never_incremental = getenv("BROTLI_MI355X_NO_INCREMENTAL_RESOLVE") != nullptr;
  static const bool check_incremental = getenv("BROTLI_MI355X_SELFTEST") != nullptr || getenv("BROTLI_MI355X_SELFTEST_RESOLVE") != nullptr;
  const bool incremental = resolve_incremental_ && !never_incremental && !use_live_ && block_records_valid_ && !touch_all_;
  resolve_blocks_skipped_ = 0;
  bool consistent = ResolvePass(final_pass, incremental);
  if (!wanted_histograms_.empty()) {
    if (!FetchShouldCompress()) {  // (a guess was wrong: the pass again, every block of it)
      wanted_guesses_.clear();
      should_compress_guess_ = true;
      consistent = ResolvePass(final_pass, false);
      if (!wanted_histograms_.empty()) throw std::runtime_error("brotli_mi355x: the resolver asked for a meta-block twice");
    }
  }
  if (incremental && check_incremental && resolve_blocks_skipped_ != 0) {
    // selftest: the pass over every block must come out the same
    const std::vector<MetaBlockPlan> mb = metablocks_;
    const std::vector<Patch> pa = patches_;
    const std::vector<TrailingInsert> tr = trailing_;
    const std::vector<Carry> ca = carries_;
    const std::vector<SegEntry> ne = next_entries_;
    const std::vector<uint8_t> de = dirty_entry_, er = entry_reason_, pe = predicted_entry_;
    const uint32_t fd = first_dirty_, pr = predicted_runs_, dd = dict_death_seg_, df = dict_flips_, rp = resume_pos_;
    const bool nel = needs_empty_last_;
    const bool full = ResolvePass(final_pass, false);
    auto same_bytes = [](const void* a, const void* b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; };
    const bool same = full == consistent && mb.size() == metablocks_.size() && same_bytes(mb.data(), metablocks_.data(), mb.size() * sizeof(MetaBlockPlan)) &&
                      pa.size() == patches_.size() && same_bytes(pa.data(), patches_.data(), pa.size() * sizeof(Patch)) && tr.size() == trailing_.size() &&
                      same_bytes(tr.data(), trailing_.data(), tr.size() * sizeof(TrailingInsert)) && ca.size() == carries_.size() &&
                      same_bytes(ca.data(), carries_.data(), ca.size() * sizeof(Carry)) && same_bytes(ne.data(), next_entries_.data(), ne.size() * sizeof(SegEntry)) &&
                      de == dirty_entry_ && er == entry_reason_ && pe == predicted_entry_ && fd == first_dirty_ && pr == predicted_runs_ &&
                      dd == dict_death_seg_ && df == dict_flips_ && rp == resume_pos_ && nel == needs_empty_last_;
    if (!same) throw std::runtime_error("selftest: the block-by-block resolver pass differs from the pass over every block");
  }
  touch_all_ = false;
  if (!block_touched_.empty()) std::fill(block_touched_.begin(), block_touched_.end(), (uint8_t)0);
  return consistent;
}

bool Lz77Stage::ResolvePass(bool final_pass, bool incremental) {
  const uint32_t nseg = (uint32_t)segments_.size();
  next_entries_.resize(nseg);  // (every element is written below)
  dbg_mismatch_ = dbg_counts_;
  memset(dbg_counts_, 0, sizeof(dbg_counts_));
  memset(dbg_first_, 0, sizeof(dbg_first_));
  const uint32_t nblocks = (uint32_t)block_segment_bytes_.size();
  if (incremental && (block_records_.size() != (size_t)nblocks + 1 || block_touched_.size() != nblocks || dirty_entry_.size() != nseg)) incremental = false;
  if (incremental) {
    // (what the last pass appended, block by block: copied for the blocks that are not looked at again)
    metablocks_prev_.swap(metablocks_);
    trailing_prev_.swap(trailing_);
    carries_prev_.swap(carries_);
  }
  std::vector<ResolveBlockRecord> records((size_t)nblocks + 1);
  metablocks_.clear();
  patches_.clear();
  trailing_.clear();
  carries_.clear();
  bool consistent = true;
  predicted_runs_ = 0;
  first_dirty_ = nseg;
  if (!incremental) {
    dirty_entry_.assign(nseg, 0);
    entry_reason_.assign(nseg, 0);
    predicted_entry_.assign(nseg, 0);
  }
  if (use_live_) live_state_.assign(nseg, LiveBlockState{});
  uint32_t block_marks = 0, block_first_dirty = nseg;  // of the block being processed
  // reason: bit 1 = the distance cache at the entry differs, bit 0 = anything else
  auto mark = [&](uint32_t k, bool same, uint8_t reason = 1) {
    if (!same) {
      consistent = false;
      dirty_entry_[k] = 1;
      entry_reason_[k] |= reason;
      if (k < first_dirty_) first_dirty_ = k;
      block_marks++;
      if (k < block_first_dirty) block_first_dirty = k;
    }
  };

  int32_t cache[4] = {4, 11, 15, 16};
  if (params_.catable) {
    for (int i = 0; i < 4; ++i) cache[i] = 0x7ffffff0;  // encode.rs:693-703
  }
  if (carry_ && carry_->valid) memcpy(cache, carry_->dist_cache, sizeof(cache));
  int32_t saved_cache[4];
  memcpy(saved_cache, cache, sizeof(cache));
  uint32_t last_insert_len = 0;
  uint64_t num_commands = 0, num_literals = 0;
  uint32_t last_flush_pos = P_.prefix_bytes + raw_head_bytes_;
  resume_pos_ = last_flush_pos;
  needs_empty_last_ = false;
  DictTracker dict;
  dict.use = P_.use_dictionary != 0;
  if (carry_ && carry_->valid) {
    dict.L = carry_->dict_lookups;
    dict.M = carry_->dict_matches;
    if (carry_->dict_dead) dict.state = DictTracker::kDead;
  }
  if (use_qspec_ && qspec_books_in_) {
    // qualities 2 .. 4: the books travel behind the slots of the table (the piece in front may have walked it the serial way, with no
    // tracker); whether the throttle has tripped is a function of the two counters (mod.rs:1957-1960)
    dict.L = qspec_lookups_;
    dict.M = qspec_matches_;
    dict.state = dict.M < (dict.L >> 7) ? DictTracker::kDead : DictTracker::kAlive;
  }
  struct LastCmd {
    bool valid = false;
    uint32_t seg = 0, idx = 0, dist_code = 0, copy_len = 0;
  } last_cmd;
  uint32_t mb_first_seg = 0;
  uint64_t mb_cmds = 0;
  const size_t max_mb = MaxMetablockSize(params_);
  const size_t max_literals = max_mb / 8, max_commands = max_mb / 8;
  (void)final_pass;
  std::map<std::pair<uint32_t, uint32_t>, bool>& should_compress_cache = should_compress_cache_;

  auto flow_now = [&]() {
    ResolveFlow f;
    memset(&f, 0, sizeof(f));  // (compared as bytes)
    memcpy(f.cache, cache, sizeof(cache));
    memcpy(f.saved_cache, saved_cache, sizeof(saved_cache));
    f.last_insert_len = last_insert_len;
    f.last_flush_pos = last_flush_pos;
    f.mb_first_seg = mb_first_seg;
    f.resume_pos = resume_pos_;
    f.num_commands = num_commands;
    f.num_literals = num_literals;
    f.mb_cmds = mb_cmds;
    f.dict_state = (uint32_t)dict.state;
    f.dict_L = dict.L;
    f.dict_M = dict.M;
    f.dict_left_alive_at = dict.left_alive_at;
    f.dict_flips = dict.flips;
    f.dict_slack = dict.slack;
    f.last_valid = last_cmd.valid ? 1u : 0u;
    f.last_seg = last_cmd.seg;
    f.last_idx = last_cmd.idx;
    f.last_dist_code = last_cmd.dist_code;
    f.last_copy_len = last_cmd.copy_len;
    f.compress_guess = should_compress_guess_ ? 1u : 0u;
    f.needs_empty_last = needs_empty_last_ ? 1u : 0u;
    return f;
  };
  auto flow_set = [&](const ResolveFlow& f) {
    memcpy(cache, f.cache, sizeof(cache));
    memcpy(saved_cache, f.saved_cache, sizeof(saved_cache));
    last_insert_len = f.last_insert_len;
    last_flush_pos = f.last_flush_pos;
    mb_first_seg = f.mb_first_seg;
    resume_pos_ = f.resume_pos;
    num_commands = f.num_commands;
    num_literals = f.num_literals;
    mb_cmds = f.mb_cmds;
    dict.state = (DictTracker::State)f.dict_state;
    dict.L = f.dict_L;
    dict.M = f.dict_M;
    dict.left_alive_at = f.dict_left_alive_at;
    dict.flips = f.dict_flips;
    dict.slack = f.dict_slack;
    last_cmd.valid = f.last_valid != 0;
    last_cmd.seg = f.last_seg;
    last_cmd.idx = f.last_idx;
    last_cmd.dist_code = f.last_dist_code;
    last_cmd.copy_len = f.last_copy_len;
    should_compress_guess_ = f.compress_guess != 0;
    needs_empty_last_ = f.needs_empty_last != 0;
  };
  uint32_t cur_block = 0xffffffffu, runs_at_entry = 0, dbg_c0[4] = {0, 0, 0, 0}, dbg_f0[4] = {0, 0, 0, 0};
  // the books of the block just left: what it marked and counted, and the state it is left in (= the next one's entry)
  auto leave_block = [&]() {
    ResolveBlockRecord& r = records[cur_block];
    r.marks = block_marks;
    r.first_dirty = block_first_dirty;
    r.predicted_runs = predicted_runs_ - runs_at_entry;
    for (int i = 0; i < 4; ++i) {
      r.dbg_counts[i] = dbg_counts_[i] - dbg_c0[i];
      r.dbg_first[i] = dbg_first_[i] - dbg_f0[i];
    }
    ResolveBlockRecord& n = records[cur_block + 1];
    n.in = flow_now();
    n.n_metablocks = (uint32_t)metablocks_.size();
    n.n_patches = (uint32_t)patches_.size();
    n.n_trailing = (uint32_t)trailing_.size();
    n.n_carries = (uint32_t)carries_.size();
  };

  uint32_t k = 0;
  while (k < nseg) {
    const uint32_t k0 = k;
    const Segment& g0 = segments_[k0];
    const uint32_t bs = g0.blk_start, be = g0.blk_end;
    uint32_t k1 = k0;
    cur_block = g0.block_index;
    {
      ResolveBlockRecord& r = records[cur_block];
      r.in = flow_now();
      r.n_metablocks = (uint32_t)metablocks_.size();
      r.n_patches = (uint32_t)patches_.size();
      r.n_trailing = (uint32_t)trailing_.size();
      r.n_carries = (uint32_t)carries_.size();
      r.merged_ext = 0;
      if (incremental && !block_touched_[cur_block]) {
        const ResolveBlockRecord& o = block_records_[cur_block];
        const ResolveBlockRecord& on = block_records_[cur_block + 1];
        // (the number of the meta-block decides whether it is one the encoder forces to be stored: part of the state)
        if (memcmp(&o.in, &r.in, sizeof(ResolveFlow)) == 0 && o.n_metablocks == r.n_metablocks) {
          // ---- the block comes out as it did in the last pass
          if (o.merged_ext != 0) {
            // (extend_last_command at its start, as below: onto the patch of the same command if there is one)
            if (!patches_.empty() && patches_.back().segment == last_cmd.seg && patches_.back().index == last_cmd.idx) {
              patches_.back().ext += o.merged_ext;
            } else {
              patches_.push_back({last_cmd.seg, last_cmd.idx, o.merged_ext});
            }
          }
          metablocks_.insert(metablocks_.end(), metablocks_prev_.begin() + o.n_metablocks, metablocks_prev_.begin() + on.n_metablocks);
          trailing_.insert(trailing_.end(), trailing_prev_.begin() + o.n_trailing, trailing_prev_.begin() + on.n_trailing);
          carries_.insert(carries_.end(), carries_prev_.begin() + o.n_carries, carries_prev_.begin() + on.n_carries);
          if (o.marks != 0) {
            consistent = false;
            if (o.first_dirty < first_dirty_) first_dirty_ = o.first_dirty;
          }
          predicted_runs_ += o.predicted_runs;
          for (int i = 0; i < 4; ++i) {
            dbg_counts_[i] += o.dbg_counts[i];
            dbg_first_[i] += o.dbg_first[i];
          }
          r.merged_ext = o.merged_ext;
          r.marks = o.marks;
          r.first_dirty = o.first_dirty;
          r.predicted_runs = o.predicted_runs;
          memcpy(r.dbg_counts, o.dbg_counts, sizeof(r.dbg_counts));
          memcpy(r.dbg_first, o.dbg_first, sizeof(r.dbg_first));
          flow_set(on.in);
          ResolveBlockRecord& n = records[cur_block + 1];
          n.in = on.in;
          n.n_metablocks = (uint32_t)metablocks_.size();
          n.n_patches = (uint32_t)patches_.size();
          n.n_trailing = (uint32_t)trailing_.size();
          n.n_carries = (uint32_t)carries_.size();
          resolve_blocks_skipped_++;
          k = block_first_segment_[cur_block + 1];
          continue;
        }
      }
    }
    block_marks = 0;
    block_first_dirty = nseg;
    runs_at_entry = predicted_runs_;
    memcpy(dbg_c0, dbg_counts_, sizeof(dbg_c0));
    memcpy(dbg_f0, dbg_first_, sizeof(dbg_f0));
    while (!(segments_[k1].flags & kSegLastInBlock)) ++k1;
    if (incremental) {
      // (a block that is looked at again starts from clean marks)
      memset(dirty_entry_.data() + k0, 0, k1 - k0 + 1);
      memset(entry_reason_.data() + k0, 0, k1 - k0 + 1);
      memset(predicted_entry_.data() + k0, 0, k1 - k0 + 1);
    }
    // ---- entry of the block
    SegEntry E{};
    E.pos = bs;
    E.apply = 0;
    memcpy(E.cache, cache, sizeof(cache));
    E.insert_len = last_insert_len;
    E.ext_allowed = 0;
    if (num_commands != 0 && last_insert_len == 0 && last_cmd.valid) {
      const uint64_t cmd_dist = (uint64_t)(int64_t)cache[0];
      if (last_cmd.dist_code < 16 || (uint64_t)last_cmd.dist_code - 15 == cmd_dist) {
        const uint64_t lpp = (uint64_t)bs - last_cmd.copy_len;
        const uint64_t max_distance = std::min<uint64_t>(lpp, P_.max_backward_limit);
        if (cmd_dist <= max_distance) E.ext_allowed = 1;
      }
    }
    dict.Hint(&E);
    {
      const SegEntry& u = entries_[k0];
      const bool cache_same = memcmp(u.cache, E.cache, sizeof(E.cache)) == 0;
      const bool ext_same = u.ext_allowed == E.ext_allowed && !E.ext_allowed;  // (an extension runs at distance cache[0])
      if (dbg_mismatch_) {
        dbg_first_[0] += !cache_same;
        dbg_first_[1] += u.ext_allowed != E.ext_allowed;
      }
      mark(k0, cache_same && u.ext_allowed == E.ext_allowed, (uint8_t)((ext_same ? 0 : 1) | (cache_same ? 0 : 2)));
    }
    next_entries_[k0] = E;
    if (use_live_) {
      LiveBlockState& st = live_state_[k0];
      st.mb_start = last_flush_pos;
      st.mb_cmds = (uint32_t)num_commands;
      st.mb_lits = (uint32_t)num_literals;
      st.last_valid = last_cmd.valid ? 1u : 0u;
      st.last_dist_code = last_cmd.dist_code;
      st.last_copy_len = last_cmd.copy_len;
      memcpy(st.saved_cache, saved_cache, sizeof(st.saved_cache));
    }
    // ---- chain through the segments of the block
    uint32_t carry = last_insert_len;  // literals pending when the segment is entered
    int32_t cur_cache[4];              // dist cache at the entry of segment j (as derived in this pass)
    memcpy(cur_cache, cache, sizeof(cache));
    for (uint32_t j = k0; j <= k1; ++j) {
      const SegEntry& used = entries_[j];
      mark(j, dict.Consume(j, used, exits_[j]));
      // A segment that was parsed from the wrong position and found nothing to copy sits in a literal spree: where
      // the chain leaves it from the right position is arithmetic.  Chain that prediction on instead of the stale exit.
      SegExit predicted;
      const SegExit* chained = &exits_[j];
      if (j > k0 && exits_[j].n_cmds == 0) {
        const SegEntry& d = next_entries_[j];
        // (a segment that a copy of the previous chain covered entirely searched nothing; if the new entry covers it as
        // well, the state just passes through, otherwise the old exit is all there is to go by)
        const bool moved = used.pos != d.pos || used.apply != d.apply || used.head_kind != d.head_kind || used.head_base != d.head_base;
        if (moved && (exits_[j].n_searches != 0 || d.pos >= segments_[j].end)) {
          predicted = exits_[j];
          PredictLiteralRun(P_, segments_[j], d, &predicted);
          chained = &predicted;
          if (exits_[j].n_searches != 0) {
            predicted_runs_++;
            if (j < k1) predicted_entry_[j + 1] = 1;
          }
        }
      }
      const SegExit& X = *chained;
      // Exit cache = the segment's own pushes on top of its entry cache.  When the entry cache derived in
      // this pass differs from the one the chain used, keep the pushes and swap the inherited tail: the
      // chain is re-run with the new entry anyway, this only lets a change travel through segments that
      // merely pass the cache along in one round instead of one segment per round.
      int32_t out_cache[4];
      for (uint32_t i = 0; i < 4; ++i) out_cache[i] = i < X.n_pushes ? X.cache[i] : cur_cache[i - X.n_pushes];
      // The pushes themselves can be inherited too: a copy found through the last-distance codes 1..3 pushes a value it
      // took FROM the cache (mod.rs:1751-1794, command.rs:48-68), and text that repeats an earlier passage at distance A
      // keeps A alive that way through hundreds of commands.  Experiment (BROTLI_MI355X_PUSH_SUBSTITUTION=1, off by
      // default): when the entry cache of this pass holds A where the chain had A', guess that an A' among the chain's
      // pushes follows.  By value alone an inherited A' cannot be told from one the hash search found afresh, and the
      // guess is wrong for the latter: measured on the emulation build it takes a round off synth.mixed and adds two to
      // finely mixed Silesia-like input.  What is needed is the provenance from the chain itself (a note per cache slot of
      // the exit: inherited from entry slot s / fresh); until then the moving front is followed by a chain instead
      // ("cache front" in RunRounds).  Once the entries agree nothing is substituted, so the fixed point is untouched.
      if (substitute_inherited_pushes_ && X.n_pushes != 0 && memcmp(used.cache, cur_cache, sizeof(cur_cache)) != 0) {
        for (uint32_t i = 0; i < 4 && i < X.n_pushes; ++i) {
          int hit = -1;
          bool ambiguous = false;
          for (int c = 0; c < 4; ++c) {
            if (used.cache[c] != out_cache[i]) continue;
            if (hit < 0) hit = c;
            else if (cur_cache[c] != cur_cache[hit]) ambiguous = true;
          }
          if (hit >= 0 && !ambiguous) out_cache[i] = cur_cache[hit];
        }
      }
      if (j == k0 && X.ext_len > 0 && last_cmd.valid) {
        if (!patches_.empty() && patches_.back().segment == last_cmd.seg && patches_.back().index == last_cmd.idx) {
          patches_.back().ext += X.ext_len;
        } else {
          patches_.push_back({last_cmd.seg, last_cmd.idx, X.ext_len});
        }
        records[cur_block].merged_ext = X.ext_len;  // (a block that is skipped next time repeats this)
        last_cmd.copy_len += X.ext_len;
      }
      num_commands += X.n_cmds;
      mb_cmds += X.n_cmds;
      num_literals += X.n_lits;
      if (X.n_cmds > 0) {
        if (carry != 0) {
          carries_.push_back({j, carry});
          num_literals += carry;
        }
        carry = X.insert_len;
      } else {
        carry += X.insert_len;
      }
      if (X.n_cmds > 0) {
        last_cmd.valid = true;
        last_cmd.seg = j;
        last_cmd.idx = X.n_cmds - 1;
        last_cmd.dist_code = X.last_dist_code;
        last_cmd.copy_len = X.last_copy_len;
      }
      if (j < k1) {
        SegEntry N{};
        N.pos = X.pos;
        N.apply = X.apply;
        memcpy(N.cache, out_cache, sizeof(N.cache));
        N.insert_len = carry;
        N.ext_allowed = 0;
        N.head_kind = X.tail_kind;
        N.head_base = X.tail_base;
        N.head_p1 = X.tail_p1;
        dict.Hint(&N);
        const SegEntry& u = entries_[j + 1];
        const bool cache_same = memcmp(u.cache, N.cache, sizeof(N.cache)) == 0;
        const bool rest_same = u.pos == N.pos && u.apply == N.apply && u.head_kind == N.head_kind && u.head_base == N.head_base && u.head_p1 == N.head_p1;
        if (dbg_mismatch_) {
          dbg_first_[2] += u.head_kind != N.head_kind || u.head_base != N.head_base || u.head_p1 != N.head_p1;
          dbg_mismatch_[0] += u.pos != N.pos;
          dbg_mismatch_[1] += u.pos == N.pos && u.apply != N.apply;
          dbg_mismatch_[2] += u.pos == N.pos && memcmp(u.cache, N.cache, sizeof(N.cache)) != 0;
          dbg_mismatch_[3] += u.pos == N.pos && u.apply != N.apply && memcmp(u.cache, N.cache, sizeof(N.cache)) == 0;
        }
        mark(j + 1, cache_same && rest_same, (uint8_t)((rest_same ? 0 : 1) | (cache_same ? 0 : 2)));
        next_entries_[j + 1] = N;
      }
      memcpy(cur_cache, out_cache, sizeof(cur_cache));
    }
    memcpy(cache, cur_cache, sizeof(cache));
    last_insert_len = carry;
    // ---- meta-block flush rule, encode.rs:2454-2483
    const bool batch_end = (k1 + 1 == nseg);
    bool is_last = batch_end && !partial_;
    const size_t processed_bytes = be - last_flush_pos;
    // (below MIN_QUALITY_FOR_BLOCK_SPLIT a meta-block is also closed once it holds 0x2fff literals + commands, encode.rs:2458-2459)
    const bool should_flush = params_.quality < 4 && num_literals + num_commands >= 0x2fff;
    const bool next_fits = processed_bytes + block_bytes_ <= max_mb && !should_flush;
    if (is_last && early_last_ && !(next_fits && num_literals < max_literals && num_commands < max_commands)) {
      // the flush rule had closed this meta-block before anybody knew that the stream ends here
      is_last = false;
      needs_empty_last_ = true;
    }
    if (!is_last && next_fits && num_literals < max_literals && num_commands < max_commands) {
      // (partial piece: the meta-block that is still open when the input runs out is left to the next piece)
      leave_block();
      if (batch_end) break;
      k = k1 + 1;
      continue;
    }
    MetaBlockPlan mb{};
    if (last_insert_len > 0) {
      trailing_.push_back({k1, last_insert_len});
      num_commands++;
      mb_cmds++;
      num_literals += last_insert_len;
      last_insert_len = 0;
    }
    mb.start = last_flush_pos;
    mb.end = be;
    mb.n_cmds = (uint32_t)mb_cmds;
    mb.n_literals = (uint32_t)num_literals;
    mb.is_last = is_last && stream_is_last_;
    mb.cmd_offset = mb_first_seg;  // temporarily: first segment; turned into a command offset by Gather()
    memcpy(mb.saved_dist_cache, saved_cache, sizeof(saved_cache));
    // should_compress, encode.rs:1325-1354
    const uint32_t bytes = mb.end - mb.start;
    bool compress = true;
    if (num_commands < (uint64_t)(bytes >> 8) + 2 && (float)num_literals > 0.99f * (float)bytes) {
      auto key = std::make_pair(mb.start, bytes);
      auto it = should_compress_cache.find(key);
      if (it == should_compress_cache.end()) {
        // answered by FetchShouldCompress() after the pass; until then: what the last literal-only meta-block was told (one
        // incompressible meta-block is followed by another).  The pass is run again only if a guess turns out wrong.
        wanted_histograms_.push_back(key);
        compress = should_compress_guess_;
        wanted_guesses_.push_back(compress ? 1 : 0);
      } else {
        compress = it->second;
      }
      should_compress_guess_ = compress;
    }
    if (std::find(forced_uncompressed_.begin(), forced_uncompressed_.end(), (uint32_t)metablocks_.size()) !=
        forced_uncompressed_.end())
      compress = false;
    mb.uncompressed = !compress;
    if (!compress) memcpy(cache, saved_cache, sizeof(cache));  // encode.rs:1994, 2142
    memcpy(mb.dist_cache_after, cache, sizeof(cache));
    mb.dict_lookups_after = dict.L;
    mb.dict_matches_after = dict.M;
    mb.dict_dead_after = dict.state == DictTracker::kDead || dict.state == DictTracker::kUnknown || (dict.use && dict.M < (dict.L >> 7));
    metablocks_.push_back(mb);
    resume_pos_ = be;
    memcpy(saved_cache, cache, sizeof(cache));
    num_commands = 0;
    num_literals = 0;
    mb_cmds = 0;
    last_cmd.valid = false;
    last_flush_pos = be;
    mb_first_seg = k1 + 1;
    k = k1 + 1;
    leave_block();
  }
  block_records_.swap(records);
  block_records_valid_ = true;
  if (block_touched_.size() != nblocks) block_touched_.assign(nblocks, 0);
  dict_death_seg_ = dict.left_alive_at;
  dict_flips_ = dict.flips;
  final_dict_lookups_ = dict.L;
  final_dict_matches_ = dict.M;
  final_dict_dead_ = dict.state == DictTracker::kDead || dict.state == DictTracker::kUnknown;
  return consistent;
}

// Dry run over the last warmup_bytes_ of segments [first_seg, nseg): fills entries_[k + 1] (position, spree
// state, distance cache) with the state in which the dry run left segment k.  dict_dead selects the static
// dictionary regime the dry run assumes.
void Lz77Stage::Warmup(uint32_t first_seg, bool dict_dead, int which, int rbuf, const std::vector<uint8_t>* only_after_dirty) {
  WarmupJob job;
  WarmupBegin(&job, first_seg, (uint32_t)segments_.size(), dict_dead, which, rbuf, only_after_dirty, 3);
  WarmupEnd(&job);
}

// The dry runs of segments [first_seg, end_seg) are queued (WarmupBegin) and their results turned into entries (WarmupEnd) in
// two steps: the host works on one half of the input's dry runs while the device is busy with the other half's (RunRounds).
void Lz77Stage::WarmupBegin(WarmupJob* job, uint32_t first_seg, uint32_t end_seg, bool dict_dead, int which, int rbuf,
                            const std::vector<uint8_t>* only_after_dirty, int mark) {
  const uint32_t nseg = (uint32_t)segments_.size();
  job->count = 0;
  job->dict_dead = dict_dead;
  job->whole_input = only_after_dirty == nullptr;
  job->mark = mark;
  if (first_seg + 1 >= nseg) return;
  // dry-run segment k to get a guess for the entry of k + 1; with only_after_dirty, just where both k and k + 1 are
  // about to be re-parsed (otherwise the resolver already chained the exact entry from a valid parse of k)
  std::vector<uint32_t>& ks = job->ks;
  ks.clear();
  for (uint32_t k = first_seg; k + 1 < nseg && k < end_seg; ++k)
    if (!only_after_dirty || ((*only_after_dirty)[k] && (*only_after_dirty)[k + 1])) ks.push_back(k);
  const uint32_t count = (uint32_t)ks.size();
  job->count = count;
  if (count == 0) return;
  // (page-locked like everything that moves every call: copies from pageable memory go through the runtime's staging)
  PinnedArray<Segment>& wsegs = job->wsegs;
  PinnedArray<SegEntry>& wentries = job->wentries;
  PinnedArray<SegExit>& wexits = job->wexits;
  wsegs.resize_discard(count);
  wentries.resize_discard(count);
  wexits.resize_discard(count);
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t k = ks[i];
    Segment g = segments_[k];
    const uint32_t len = g.end - g.start;
    if (len > warmup_bytes_) g.start = g.end - warmup_bytes_;
    g.flags = (g.flags & kSegTailStitched) | kSegWarmup;
    wsegs[i] = g;
    SegEntry e = entries_[k];
    e.pos = g.start;
    e.apply = g.start + P_.spree_window;
    e.dict_lookups = dict_dead ? DictTracker::kDeadL : DictTracker::kAliveL;
    e.dict_matches = dict_dead ? DictTracker::kDeadM : DictTracker::kAliveM;
    wentries[i] = e;
  }
  timeline().stamp("wu-prepared");
  Segment* wsegs_dev = (Segment*)dev_alloc(count * sizeof(Segment));
  SegEntry* wentries_dev = (SegEntry*)dev_alloc(count * sizeof(SegEntry));
  SegExit* wexits_dev = (SegExit*)dev_alloc(count * sizeof(SegExit));
  dev_h2d(wsegs_dev, wsegs.data(), count * sizeof(Segment));
  dev_h2d(wentries_dev, wentries.data(), count * sizeof(SegEntry));
  lz77_parse_custom(P_, B_, which, rbuf, wsegs_dev, wentries_dev, wexits_dev, count);
  timeline().stamp("wu-queued");
  dev_d2h_async(wexits.data(), wexits_dev, count * sizeof(SegExit));
  dev_mark_n(mark);
  job->wsegs_dev = wsegs_dev;
  job->wentries_dev = wentries_dev;
  job->wexits_dev = wexits_dev;
}

void Lz77Stage::WarmupEnd(WarmupJob* job) {
  const uint32_t nseg = (uint32_t)segments_.size();
  const uint32_t count = job->count;
  if (count == 0) return;
  const std::vector<uint32_t>& ks = job->ks;
  PinnedArray<Segment>& wsegs = job->wsegs;
  PinnedArray<SegEntry>& wentries = job->wentries;
  PinnedArray<SegExit>& wexits = job->wexits;
  const bool dict_dead = job->dict_dead;
  const bool only_after_dirty = !job->whole_input;
  dev_wait_mark_n(job->mark);
  timeline().stamp("wu-exits");
  dev_free(job->wsegs_dev);
  dev_free(job->wentries_dev);
  dev_free(job->wexits_dev);
  if (!dict_dead && !only_after_dirty) {
    // lookups / matches the dry runs saw, scaled to the whole segment: a forecast of where the throttle trips
    if (warm_lookups_.size() != nseg || ks[0] == 0) {
      warm_lookups_.assign(nseg, 0.0);
      warm_matches_.assign(nseg, 0.0);
    }
    for (uint32_t i = 0; i < count; ++i) {
      const double scale = (double)(segments_[ks[i]].end - segments_[ks[i]].start) / (double)(wsegs[i].end - wsegs[i].start);
      warm_lookups_[ks[i]] = scale * (double)(wexits[i].dict_lookups - wentries[i].dict_lookups);
      warm_matches_[ks[i]] = scale * (double)(wexits[i].dict_matches - wentries[i].dict_matches);
    }
  }
  for (uint32_t i = 0; i < count; ++i) {
    SegEntry& e = entries_[ks[i] + 1];
    // distance cache guess: the pushes the dry run made, on top of the guess for the segment it ran in (a dry run that
    // pushed fewer than 4 distances still carries its cold start values behind them)
    const uint32_t np = std::min<uint32_t>(wexits[i].n_pushes, 4);
    int32_t guess[4];
    for (uint32_t c = 0; c < 4; ++c) guess[c] = c < np ? wexits[i].cache[c] : entries_[ks[i]].cache[c - np];
    memcpy(e.cache, guess, sizeof(e.cache));
    if (!(segments_[ks[i] + 1].flags & kSegFirstInBlock)) {
      e.pos = wexits[i].pos;
      e.apply = wexits[i].apply;
      e.head_kind = wexits[i].tail_kind;
      e.head_base = wexits[i].tail_base;
      e.head_p1 = wexits[i].tail_p1;
    } else {
      // will extend_last_command run at the start of the next block (encode.rs:2435-2437, 360-400)?  Same test as in
      // Resolve(), on the dry run's last command.  A wrong guess is caught there.
      const SegExit& x = wexits[i];
      e.ext_allowed = 0;
      // (the dry run is not told that its segment ends the block, so it does not add the unsearched tail of the block to
      // its pending literals: only a copy that reaches the very end of the block leaves nothing pending)
      if (x.n_cmds > 0 && x.insert_len == 0 && x.pos >= segments_[ks[i]].blk_end) {
        const uint64_t cmd_dist = (uint64_t)(int64_t)x.cache[0];
        if (x.last_dist_code < 16 || (uint64_t)x.last_dist_code - 15 == cmd_dist) {
          const uint64_t lpp = (uint64_t)segments_[ks[i] + 1].blk_start - x.last_copy_len;
          if (cmd_dist <= std::min<uint64_t>(lpp, P_.max_backward_limit)) e.ext_allowed = 1;
        }
      }
    }
  }
  static const bool no_spree_guess = getenv("BROTLI_MI355X_NO_SPREE_GUESS") != nullptr;
  if (!only_after_dirty && !no_spree_guess) {
    // A dry run that found nothing to copy in the tail of its segment says "incompressible here", but the state it ends in
    // is that of a spree begun at the tail's start; in the real parse the spree began wherever the last copy ended --
    // blocks away in random data -- and where it steps (every 9th, every 17th position, mod.rs:2529-2546) is a matter of
    // that phase.  Carry the phase along instead: from the entry guessed for segment k (the block start, the end of a
    // copy the previous dry run saw, or this same arithmetic one segment earlier) through a segment k without copies.
    // Like every entry this is a guess that the first parse verifies; on incompressible input it spares the round that
    // parses every segment a second time, and the flag changes (row rebuilds) that go with it.
    for (uint32_t i = 0; i < count; ++i) {
      const uint32_t k = ks[i];
      if (wexits[i].n_cmds != 0) continue;
      if (segments_[k + 1].flags & kSegFirstInBlock) continue;
      SegEntry from = entries_[k];
      if (segments_[k].flags & kSegFirstInBlock) {
        if (from.ext_allowed) continue;  // (how far extend_last_command gets is not known yet)
        from.pos = segments_[k].blk_start;
        from.apply = from.pos + P_.spree_window;
        from.head_kind = kHeadNone;
        from.head_base = from.head_p1 = 0;
      }
      SegExit x{};
      PredictLiteralRun(P_, segments_[k], from, &x);
      SegEntry& e = entries_[k + 1];
      e.pos = x.pos;
      e.apply = x.apply;
      e.head_kind = x.tail_kind;
      e.head_base = x.tail_base;
      e.head_p1 = x.tail_p1;
    }
  }
  stats_.segments_parsed += (uint64_t)count * warmup_bytes_ / segment_bytes_;
  touch_all_ = true;  // (entries behind every dry run)
}

// Segments whose entry changed in the distance cache only and whose parse found nothing to copy: instead of parsing
// them again, ask the device whether any of the new cache distances would match at a searched position.  If not, the
// old parse holds for the new entry as it stands (no command to re-code, the cache passes through).
uint32_t Lz77Stage::RecheckCacheOnly(int which, std::vector<uint32_t>* accepted) {
  if (P_.hasher_kind == 9) return 0;
  const uint32_t nseg = (uint32_t)segments_.size();
  std::vector<CacheCheck> items;
  for (uint32_t k = 0; k < nseg; ++k) {
    if (!dirty_entry_[k] || entry_reason_[k] != 2) continue;
    const SegExit& x = exits_[k];
    if (x.n_cmds != 0 || x.ext_len != 0 || x.n_pushes != 0) continue;
    CacheCheck c;
    c.segment = k;
    memcpy(c.cache, next_entries_[k].cache, sizeof(c.cache));
    items.push_back(c);
  }
  if (items.empty()) return 0;
  const uint32_t count = (uint32_t)items.size();
  CacheCheck* items_dev = (CacheCheck*)dev_alloc(count * sizeof(CacheCheck) + 64);
  uint8_t* ok_dev = (uint8_t*)dev_alloc(count + 64);
  dev_h2d(items_dev, items.data(), count * sizeof(CacheCheck));
  lz77_check_cache(P_, B_, which, items_dev, count, ok_dev);
  std::vector<uint8_t> ok(count);
  dev_d2h(ok.data(), ok_dev, count);
  dev_free(items_dev);
  dev_free(ok_dev);
  uint32_t cleared = 0;
  for (uint32_t i = 0; i < count; ++i) {
    if (!ok[i]) continue;
    const uint32_t k = items[i].segment;
    memcpy(entries_[k].cache, items[i].cache, sizeof(items[i].cache));
    TouchSegment(k);
    if (accepted) accepted->push_back(k);  // (the device's copy of this entry is stale now)
    dirty_entry_[k] = 0;
    ++cleared;
  }
  return cleared;
}

namespace {
struct Timer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit Timer(bool enabled) : on(enabled) {
    if (on) {
      dev_sync();
      t0 = std::chrono::steady_clock::now();
    }
  }
  void stop(double* acc) {
    if (!on) return;
    dev_sync();
    auto t1 = std::chrono::steady_clock::now();
    *acc += std::chrono::duration<double, std::milli>(t1 - t0).count();
    t0 = t1;
  }
};
}  // namespace

void Lz77Stage::Run() {
  stats_ = Lz77Stats{};
  const bool prof = getenv("BROTLI_MI355X_PROFILE") != nullptr;
  Timer total(true);
  Timer tm(prof);
  should_compress_cache_.clear();
  block_records_valid_ = false;  // (Resolve() block by block: nothing of an earlier call, another forced meta-block, another cut)
  touch_all_ = true;
  resolve_incremental_ = false;
  const uint32_t nseg = (uint32_t)segments_.size();
  if (nseg == 0) {
    metablocks_.clear();
    total_cmds_ = 0;
    return;
  }
  if (use_zopfli_ || use_quick_) {
    if (use_quick_) { if (use_qspec_) RunQuickSpec(); else RunQuick(); } else RunZopfli();
    tm.stop(&stats_.ms_parse);
    Gather();
    tm.stop(&stats_.ms_gather);
    total.stop(&stats_.ms_total);
    return;
  }
  lz77_compute_keys(P_, B_);
  tm.stop(&stats_.ms_keys);
  lz77_sort_by_key(P_, B_);
  key_first_.resize_discard(65537);
  key_last_.resize_discard(65537);
  dev_d2h_async(key_first_.data(), B_.key_first, 65537 * 4);
  dev_d2h_async(key_last_.data(), B_.key_last, 65537 * 4);
  uint32_t run_samples = 0;
  dev_d2h_async(&run_samples, B_.changed_count + 8, 4);
  timeline().stamp("sort-queued");
  dev_sync();
  timeline().stamp("sorted");
  // long runs of one byte (zero fill ...): every candidate of every position inside a run matches to the end of the
  // block; the run table lets the chains jump over a run instead of comparing it 32 bytes at a time (lz77_chain.h).
  // One sample in 64 positions: 64 samples ~ 4 KiB of runs.
  if (run_samples >= 64 && getenv("BROTLI_MI355X_NO_RUN_TABLE") == nullptr) {
    if (!B_.run_end) B_.run_end = (uint32_t*)dev_alloc_uninit((size_t)P_.total_bytes * 4 + 64);
    lz77_run_table(P_, B_);
    if (getenv("BROTLI_MI355X_SELFTEST")) {
      const uint32_t n = P_.total_bytes;
      std::vector<uint32_t> got(n);
      std::vector<uint8_t> text(n);
      dev_d2h(got.data(), B_.run_end, (size_t)n * 4);
      dev_d2h(text.data(), B_.text, n);
      uint32_t end = n;
      for (uint32_t p = n; p-- > 0;) {
        if (!(p + 1 < n && text[p + 1] == text[p])) end = p + 1;
        if (got[p] != end) throw std::runtime_error("selftest: run table wrong at position " + std::to_string(p));
      }
    }
  } else if (B_.run_end) {
    dev_free(B_.run_end);
    B_.run_end = nullptr;
  }
  has_big_keys_ = false;
  for (uint32_t key = 0; key < 65536 && !has_big_keys_; ++key) has_big_keys_ = key_last_[key] - key_first_[key] >= 65536u;
  // a later piece of a stream: the ring counters of the reference have been running since the start of the stream
  if (carry_ && carry_->valid && carry_->key_counts.size() == 65536) {
    if (!count_base_dev_) count_base_dev_ = (uint32_t*)dev_alloc(65536 * 4);
    dev_h2d(count_base_dev_, carry_->key_counts.data(), 65536 * 4);
    B_.count_base = count_base_dev_;
    has_big_keys_ = true;
  } else {
    B_.count_base = nullptr;
  }
  if (P_.reset_pos) has_big_keys_ = true;  // (the ring counters start over inside this text: wrap marks for every key)
  tm.stop(&stats_.ms_sort);
  timeline().stamp("rounds");
  RunRounds(true);
  tm.stop(&stats_.ms_resolve);
  timeline().stamp("gather");
  Gather();
  timeline().stamp("gathered");
  tm.stop(&stats_.ms_gather);
  total.stop(&stats_.ms_total);
  if (prof)
    fprintf(stderr, "lz77 stage: keys %.2f sort %.2f init %.2f warmup %.2f rank %.2f parse %.2f resolve %.2f gather %.2f | total %.2f ms, rounds %u\n",
            stats_.ms_keys, stats_.ms_sort, stats_.ms_init, stats_.ms_warmup, stats_.ms_rank, stats_.ms_parse, stats_.ms_resolve, stats_.ms_gather,
            stats_.ms_total, (unsigned)stats_.rounds);
  if (prof) fprintf(stderr, "  host: Resolve() %.2f ms, recheck + scheduling %.2f ms\n", host_resolve_ms_, host_schedule_ms_);
  host_resolve_ms_ = host_schedule_ms_ = 0;
}

// Re-cut the input into segments of a different size (the sort by key stays valid).
void Lz77Stage::Resegment(uint32_t segment_bytes) {
  segment_bytes_ = std::min(std::max(segment_bytes, 256u), block_bytes_);
  P_.cmd_slab_stride = segment_bytes_ / 2 + 8;
  BuildSegments();
  P_.num_segments = (uint32_t)segments_.size();
  const size_t need = (size_t)total_cmd_slots_ * sizeof(Command) + 64;
  if (need > cmds_bytes_) {
    dev_free(B_.cmds);
    B_.cmds = (Command*)dev_alloc_uninit(need);
    cmds_bytes_ = need;
  }
  segments_upload_.resize_discard(segments_.size());
  memcpy(segments_upload_.data(), segments_.data(), segments_.size() * sizeof(Segment));
  dev_h2d(B_.segments, segments_upload_.data(), segments_.size() * sizeof(Segment));
  if (B_.checkpoints) dev_memset(B_.checkpoints, 0, ((size_t)P_.total_bytes / kCheckpointStride + 2) * sizeof(Checkpoint));  // (other segments: no record holds)
}

// first guess of the entries: every chain starts at its segment start with the default cache
void Lz77Stage::InitEntries() {
  const uint32_t nseg = (uint32_t)segments_.size();
  entries_.assign(nseg, SegEntry{});
  for (uint32_t k = 0; k < nseg; ++k) {
    SegEntry& e = entries_[k];
    e.pos = segments_[k].start;
    e.apply = segments_[k].start + P_.spree_window;
    const int32_t d[4] = {4, 11, 15, 16};
    for (int i = 0; i < 4; ++i) e.cache[i] = (carry_ && carry_->valid) ? carry_->dist_cache[i] : (params_.catable ? 0x7ffffff0 : d[i]);
    if (k != 0) {
      e.dict_lookups = DictTracker::kAliveL;
      e.dict_matches = DictTracker::kAliveM;
    }
  }
  exits_.assign(nseg, SegExit{});
}

// Live chains (lz77_live.h): ONE chain walks through all input blocks on a copy of the reference's bucket rings, which is
// materialised for its first block from the stored / masked flags of the prefix (custom dictionary, or the window of the
// stream so far).  The chain enters every further block by itself (flush rule and extend_last_command on its own books);
// the host resolver then replays the exits like after any other round.  Where it derives another entry than the chain
// used -- a meta-block that is stored uncompressed hands on the distance cache of its start -- everything in front of that
// block is final: the rings are materialised again for its start and the chain walks on from there.
// BROTLI_MI355X_LIVE_VERIFY=1: every search is logged and, at the end, repeated on its own against the ring that the final
// flags imply for its position (a check of the chain against the definition of the rings, all searches in parallel).
void Lz77Stage::RunLive() {
  const bool prof = getenv("BROTLI_MI355X_PROFILE") != nullptr;
  const bool debug = getenv("BROTLI_MI355X_DEBUG") != nullptr;
  Timer tm(prof);
  const uint32_t nseg = (uint32_t)segments_.size();
  InitFlags();
  InitEntries();
  entries_[0].dict_exact = 1;  // (the chain starts with the true counters of the stream and keeps them)
  if (P_.use_dictionary) {
    entries_[0].dict_lookups = (carry_ && carry_->valid) ? carry_->dict_lookups : 0u;
    entries_[0].dict_matches = (carry_ && carry_->valid) ? carry_->dict_matches : 0u;
    if (carry_ && carry_->valid && carry_->dict_dead) {
      entries_[0].dict_lookups = DictTracker::kDeadL;
      entries_[0].dict_matches = DictTracker::kDeadM;
    }
  }
  tm.stop(&stats_.ms_init);
  uint32_t* first_dev = (uint32_t*)dev_alloc(64);
  uint32_t* start_dev = (uint32_t*)dev_alloc(64);
  PinnedArray<uint32_t> first_start;
  PinnedArray<LiveBlockState> books;
  first_start.resize_discard(2);
  books.resize_discard(nseg);
  for (uint32_t k = 0; k < nseg; ++k) books[k] = LiveBlockState{};
  books[0].mb_start = P_.prefix_bytes + raw_head_bytes_;
  memcpy(books[0].saved_cache, entries_[0].cache, sizeof(books[0].saved_cache));
  int which = 0;
  uint32_t from = 0;
  const uint32_t max_rounds = getenv("BROTLI_MI355X_MAX_ROUNDS") ? (uint32_t)atoi(getenv("BROTLI_MI355X_MAX_ROUNDS")) : nseg + 8;
  bool done = false;
  for (uint32_t round = 0; round < max_rounds && !done; ++round) {
    stats_.rounds++;
    if (round != 0) {
      entries_[from] = next_entries_[from];
      books[from] = live_state_[from];
    }
    first_start[0] = from;
    first_start[1] = segments_[from].blk_start;
    dev_h2d(B_.entries, entries_.data(), (size_t)nseg * sizeof(SegEntry));
    dev_h2d(L_.state, books.data(), (size_t)nseg * sizeof(LiveBlockState));
    dev_h2d(first_dev, first_start.data(), 4);
    dev_h2d(start_dev, first_start.data() + 1, 4);
    lz77_live_index(P_, B_, L_, which);
    lz77_live_materialise(P_, B_, L_, which, first_dev, start_dev, 1);
    dev_d2d(B_.flags[which ^ 1], B_.flags[which], (size_t)P_.total_bytes + 64);
    tm.stop(&stats_.ms_rank);
    lz77_live_parse(P_, B_, L_, which, first_dev, 1);
    stats_.segments_parsed += nseg - from;
    dev_d2h_async(exits_.data(), B_.exits, (size_t)nseg * sizeof(SegExit));
    dev_d2h_async(entries_.data(), B_.entries, (size_t)nseg * sizeof(SegEntry));  // (the chain chose the entries behind its first block)
    dev_sync();
    tm.stop(&stats_.ms_parse);
    which ^= 1;
    const auto host_t0 = std::chrono::steady_clock::now();
    Resolve(false);
    host_resolve_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    uint32_t wrong = nseg;
    for (uint32_t k = 0; k < nseg && wrong == nseg; ++k)
      if (dirty_entry_[k]) wrong = k;
    if (debug) fprintf(stderr, "live chain from block %u of %u: first block entered in another state than the resolver derives: %d\n", from, nseg, wrong == nseg ? -1 : (int)wrong);
    if (wrong != nseg && wrong <= from && round != 0) throw std::runtime_error("brotli_mi355x: live chain and resolver disagree about an entry the resolver chose");
    tm.stop(&stats_.ms_resolve);
    from = wrong;
    done = wrong == nseg;
  }
  dev_free(first_dev);
  dev_free(start_dev);
  if (!done) throw std::runtime_error("brotli_mi355x: backward-reference search (live chain) did not reach a fixed point");
  final_flags_ = which;
  if (live_verify_) {
    SegGeometry geo{};
    geo.prefix_bytes = P_.prefix_bytes;
    geo.first_block_start = segments_[0].blk_start;
    geo.block_bytes = block_bytes_;
    geo.num_blocks = nseg;
    geo.num_segments = nseg;
    geo.block_size = 1u << P_.block_bits;
    uint8_t* dirty_dev = (uint8_t*)dev_alloc(nseg + 64);
    std::vector<uint8_t> failed(nseg);
    lz77_live_slots(P_, B_, L_);
    lz77_live_index(P_, B_, L_, which);
    lz77_live_verify(P_, B_, L_, -1, which, geo, nullptr, dirty_dev);
    dev_d2h(failed.data(), dirty_dev, nseg);
    dev_free(dirty_dev);
    for (uint32_t k = 0; k < nseg; ++k)
      if (failed[k]) throw std::runtime_error("brotli_mi355x: live chain verification failed in block " + std::to_string(k));
  }
  for (uint32_t k = 0; k < nseg; ++k) stats_.searches += exits_[k].n_searches;
}

ZopfliCarry::~ZopfliCarry() {
  dev_free(buckets);
  dev_free(forest);
}

void Lz77Stage::ExportZopfli(StreamCarry* co, bool partial) {
  auto zc = std::make_shared<ZopfliCarry>();
  zc->lgwin = Z_.lgwin;
  zc->text_base = (carry_ && carry_->valid) ? carry_->stream_base : 0;
  if (partial) {
    if (!zsnap_buckets_) throw std::runtime_error("brotli_mi355x: quality 10 / 11: no snapshot of the trees at the resume point");
    zc->buckets = zsnap_buckets_;
    zc->forest = zsnap_forest_;
    zsnap_buckets_ = zsnap_forest_ = nullptr;
  } else {
    zc->buckets = Z_.buckets;
    zc->forest = Z_.forest;
    Z_.buckets = Z_.forest = nullptr;
  }
  co->zopfli = std::move(zc);
}

QuickCarry::~QuickCarry() { dev_free(table); }

void Lz77Stage::ExportQuick(StreamCarry* co, bool partial) {
  auto qc = std::make_shared<QuickCarry>();
  qc->text_base = (carry_ && carry_->valid) ? carry_->stream_base : 0;
  if (use_qspec_) {
    // The speculative path kept no table: the slots as the reference's hold them behind the text -- or, for a partial piece, in front
    // of the block at the resume point, before that block's StitchToPreviousBlock files the three positions in front of it (the next
    // piece does that itself) -- from the final flags.  The books of the throttle behind the slots: the resolver's, as the encoder
    // has put them into the carry (once the throttle has tripped only "matches < lookups >> 7" matters: nothing is looked up any more).
    uint32_t* table = (uint32_t*)dev_alloc_uninit((size_t)quick_table_words(Q_) * 4 + 64);
    const uint32_t first = segments_.empty() ? P_.total_bytes : segments_[0].blk_start;
    uint32_t upto = 0xffffffffu;
    if (partial) upto = resume_pos_ >= 3 ? resume_pos_ - 3 : 0;
    lz77_qspec_table(P_, B_, Q_, S_, upto, table);
    uint32_t books[16] = {0};
    if (partial && resume_pos_ <= first) {
      // (nothing of this piece lies in front of the resume point: the books it came in with)
      books[0] = qspec_books_in_ ? qspec_lookups_ : 0u;
      books[1] = qspec_books_in_ ? qspec_matches_ : 0u;
    } else {
      books[0] = co->dict_dead ? DictTracker::kDeadL : co->dict_lookups;
      books[1] = co->dict_dead ? DictTracker::kDeadM : co->dict_matches;
    }
    if (!Q_.use_dictionary) books[0] = books[1] = 0;
    dev_h2d(table + quick_books_at(Q_), books, sizeof(books));
    dev_sync();
    qc->table = table;
    co->quick = std::move(qc);
    return;
  }
  if (partial) {
    if (!qsnap_table_) throw std::runtime_error("brotli_mi355x: qualities 2 .. 4: no snapshot of the hash table at the resume point");
    qc->table = qsnap_table_;
    qsnap_table_ = nullptr;
  } else {
    qc->table = Q_.table;
    Q_.table = nullptr;
  }
  co->quick = std::move(qc);
}

// Qualities 2 .. 4: like RunZopfli -- the blocks go through the device one after the other on the stream's BasicHasher table, the
// host resolver replays the exits in between (flush rule incl. the 0x2fff rule of these qualities, should_compress,
// extend_last_command) and hands the next block its entry.
void Lz77Stage::RunQuick() {
  const bool debug = getenv("BROTLI_MI355X_DEBUG") != nullptr;
  const uint32_t nseg = (uint32_t)segments_.size();
  if (P_.reset_pos != 0) throw std::runtime_error("brotli_mi355x: qualities 2 .. 4 across the reference's 32-bit position wrap are not supported");
  InitEntries();
  const size_t table_bytes = (size_t)quick_table_words(Q_) * 4;
  if (carry_ && carry_->valid && carry_->quick) {
    const QuickCarry& qc = *carry_->quick;
    if (carry_->stream_base < qc.text_base) throw std::runtime_error("brotli_mi355x: qualities 2 .. 4: the carried hash table does not fit this piece");
    lz77_quick_import(Q_, qc.table, (uint32_t)(carry_->stream_base - qc.text_base));
  } else {
    lz77_quick_init(Q_);
    if (P_.prefix_bytes > 1) lz77_quick_prepend(P_, B_, Q_, P_.prefix_bytes);  // custom dictionary, encode.rs:1163-1194
  }
  auto snapshot = [&]() {
    if (!qsnap_table_) qsnap_table_ = (uint32_t*)dev_alloc_uninit(table_bytes + 64);
    dev_d2d(qsnap_table_, Q_.table, table_bytes);
  };
  bool starts_metablock = true;
  for (uint32_t from = 0; from < nseg; ++from) {
    stats_.rounds++;
    if (from != 0) entries_[from] = next_entries_[from];
    if (partial_ && starts_metablock) snapshot();  // (the next piece starts again at the first block of the meta-block still open)
    dev_h2d(B_.entries + from, entries_.data() + from, sizeof(SegEntry));
    lz77_quick_block(P_, B_, Q_, from);
    stats_.segments_parsed++;
    dev_d2h(exits_.data() + from, B_.exits + from, sizeof(SegExit));
    Resolve(false);
    for (uint32_t k = 0; k <= from; ++k)
      if (dirty_entry_[k]) throw std::runtime_error("brotli_mi355x: resolver and quick parse disagree about the entry of a finished block");
    if (debug) fprintf(stderr, "quick block %u of %u done: %u commands, %u bytes pending\n", from, nseg, exits_[from].n_cmds, exits_[from].insert_len);
    starts_metablock = false;
    for (const MetaBlockPlan& mb : metablocks_)
      if (mb.end == segments_[from].blk_end) starts_metablock = true;
  }
  if (partial_ && starts_metablock) snapshot();
  final_flags_ = 0;
  for (uint32_t k = 0; k < nseg; ++k) stats_.searches += exits_[k].n_searches;
}

// Qualities 2 .. 4 on the speculative path (quick_spec.h, quick_api.h).  Round 0 parses every segment from a cold guess (its own
// start, the default distance cache) on the candidates of "every position filed"; after every launch the resolver chains the exits
// into entries and the candidates are derived again from the flags the chains wrote; a segment is parsed again when its entry
// changed or a candidate of a position it searched did.  A candidate of p hangs on the flags in front of p only, so the prefix of
// segments that are final grows by at least one per round; text settles in a handful of rounds.  An input that does not (more than
// kMaxRounds launches) is handed to the serial path.
void Lz77Stage::RunQuickSpec() {
  const bool debug = getenv("BROTLI_MI355X_DEBUG") != nullptr;
  auto stamp = [&](const char* what) { timeline().stamp(what); };
  const uint32_t nseg = (uint32_t)segments_.size();
  if (P_.reset_pos != 0) throw std::runtime_error("brotli_mi355x: qualities 2 .. 4 across the reference's 32-bit position wrap are not supported");
  resolve_incremental_ = false;
  const auto t_begin = std::chrono::steady_clock::now();
  InitEntries();
  const bool continuing = carry_ && carry_->valid;
  S_.base = nullptr;
  qspec_books_in_ = false;
  if (continuing && carry_->quick) {
    // a later piece: the table of the piece in front, moved to this piece's text positions, is what the slots hold to begin with; the
    // books of the throttle sit behind its slots
    const QuickCarry& qc = *carry_->quick;
    if (carry_->stream_base < qc.text_base) throw std::runtime_error("brotli_mi355x: qualities 2 .. 4: the carried hash table does not fit this piece");
    lz77_quick_import(Q_, qc.table, (uint32_t)(carry_->stream_base - qc.text_base));
    S_.base = Q_.table;
    uint32_t books[2] = {0, 0};
    dev_d2h(books, Q_.table + quick_books_at(Q_), sizeof(books));
    qspec_books_in_ = true;
    qspec_lookups_ = books[0];
    qspec_matches_ = books[1];
    if (P_.use_dictionary) {
      entries_[0].dict_lookups = books[0];
      entries_[0].dict_matches = books[1];
      entries_[0].dict_exact = 1;
    }
  }
  const bool prof_sync = debug && getenv("BROTLI_MI355X_PROFILE") != nullptr;
  auto lap = [&](const char* what) {
    if (!prof_sync) return;
    dev_sync();
    fprintf(stderr, "quick %s: %.2f ms into the parse\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  if (!qspec_coarse_) lz77_qspec_index(P_, B_, Q_, S_);  // (parse-independent: a restart with other segments keeps it)
  if (qspec_coarse_) {
    // what the pass with small segments said about the state at every block start
    for (uint32_t k = 0; k < nseg; ++k) {
      auto it = saved_block_guess_.find(segments_[k].blk_start);
      if (it == saved_block_guess_.end() || !(segments_[k].flags & kSegFirstInBlock)) continue;
      memcpy(entries_[k].cache, it->second.cache, sizeof(entries_[k].cache));
      entries_[k].ext_allowed = it->second.ext_allowed;
    }
  }
  lap("index");
  lz77_qspec_init_flags(P_, B_, Q_, S_, segments_[0].blk_start, !continuing);
  lz77_qspec_candidates(P_, B_, Q_, S_, nullptr, nullptr);
  lap("first candidates");
  SegGeometry geo{};
  geo.prefix_bytes = P_.prefix_bytes;
  geo.first_block_start = segments_[0].blk_start;
  geo.block_bytes = block_bytes_;
  geo.num_blocks = (uint32_t)block_segment_bytes_.size();
  uint32_t* geo_tables = (uint32_t*)dev_alloc((block_first_segment_.size() + block_segment_bytes_.size()) * 4 + 64);
  dev_h2d(geo_tables, block_first_segment_.data(), block_first_segment_.size() * 4);
  dev_h2d(geo_tables + block_first_segment_.size(), block_segment_bytes_.data(), block_segment_bytes_.size() * 4);
  geo.block_first_segment = geo_tables;
  geo.block_segment_bytes = geo_tables + block_first_segment_.size();
  geo.num_segments = nseg;
  geo.block_size = 1;
  uint8_t* dirty_dev = (uint8_t*)dev_alloc(nseg + 64);
  uint32_t* list_dev = (uint32_t*)dev_alloc((size_t)nseg * 4 + 64);
  SegEntry* up_entries_dev = (SegEntry*)dev_alloc_uninit((size_t)nseg * sizeof(SegEntry) + 64);
  SegExit* got_exits_dev = (SegExit*)dev_alloc_uninit((size_t)nseg * sizeof(SegExit) + 64);
  PinnedArray<uint8_t> dirty;
  dirty.resize_discard(nseg);
  // One chain per block (the restart below): every chain on a table of its own, read and filed into as the reference does
  // (QsTables::own), so that what a chain finds inside its own block is exact in the round it is parsed.  A block is parsed again
  // when its entry changed or a filing changed anywhere in front of it (lz77_qspec_first_change); the candidates are only kept up
  // for the tables (the scan they are read from) and for the table handed to the next piece of the stream.
  const uint32_t own_stride = (quick_slots(Q_) + 63u) & ~63u;
  // (It is the LAST resort in front of the serial walk: a launch of chains on candidates is cheaper -- no tables to derive, nothing but
  // the chains that have a reason is parsed again -- and most inputs that are re-cut into blocks settle on them; the blocks switch to
  // tables of their own where that iteration stops making progress, from the first block that is not final.)
  const uint32_t own_window = getenv("BROTLI_MI355X_QUICK_OWN_WINDOW") ? (uint32_t)atoi(getenv("BROTLI_MI355X_QUICK_OWN_WINDOW")) : 128u;
  const bool can_own = qspec_coarse_ && getenv("BROTLI_MI355X_QUICK_NO_OWN_TABLES") == nullptr;
  bool own = can_own && getenv("BROTLI_MI355X_QUICK_OWN_TABLES_FIRST") != nullptr;  // (tests: the tables from round 0 on)
  const uint32_t own_slots = own ? nseg : std::min(nseg, own_window);
  uint32_t* own_tables = can_own ? (uint32_t*)dev_alloc_uninit((size_t)own_slots * own_stride * 4 + 64) : nullptr;
  uint32_t own_from = 0, own_upto = nseg;  // (with the tables from round 0 on, that round parses every block)
  uint32_t* first_change_dev = (uint32_t*)dev_alloc(64);
  PinnedArray<uint32_t> first_change;
  first_change.resize_discard(16);
  first_change[0] = 0xffffffffu;
  // segment that holds text position q (segments_ are in text order)
  auto segment_of = [&](uint32_t q) {
    uint32_t lo = 0, hi = nseg;
    while (lo + 1 < hi) {
      const uint32_t mid = (lo + hi) / 2;
      if (segments_[mid].start <= q) lo = mid; else hi = mid;
    }
    return lo;
  };
  PinnedArray<uint32_t> chg_count;
  chg_count.resize_discard(16);
  chg_count[0] = chg_count[1] = 0;
  RoundBuffers& R = round_buffers_;
  R.up_index.resize_discard(nseg);
  R.up_entries.resize_discard(nseg);
  R.got_exits.resize_discard(nseg);
  stamp("qs-index-queued");
  // ---- warm-up: a dry run over the tail of every segment from a cold state (nothing written but the exit) guesses the state in which
  // the parse enters the next one -- greedy parses fall into step within a few commands -- so that round 0 already starts almost every
  // chain where the true parse does, instead of spending a launch over everything on finding that out (RunRounds does the same)
  const uint32_t warm = getenv("BROTLI_MI355X_QUICK_WARMUP") ? (uint32_t)atoi(getenv("BROTLI_MI355X_QUICK_WARMUP")) : std::min(192u, segment_bytes_ / 2u);
  if (warm != 0 && nseg > 1) {
    PinnedArray<Segment> wsegs;
    PinnedArray<SegEntry> wentries;
    PinnedArray<SegExit> wexits;
    wsegs.resize_discard(nseg);
    wentries.resize_discard(nseg);
    wexits.resize_discard(nseg);
    for (uint32_t k = 0; k < nseg; ++k) {
      Segment g = segments_[k];
      if (g.end - g.start > warm) {
        g.start = g.end - warm;
        g.flags &= ~(uint32_t)kSegFirstInBlock;
      }
      g.flags |= kSegWarmup;
      wsegs[k] = g;
      SegEntry e = entries_[k];
      e.pos = g.start;
      e.apply = g.start + P_.spree_window;
      e.ext_allowed = 0;
      e.head_kind = kHeadNone;
      wentries[k] = e;
    }
    Segment* wsegs_dev = (Segment*)dev_alloc_uninit((size_t)nseg * sizeof(Segment) + 64);
    dev_h2d(wsegs_dev, wsegs.data(), (size_t)nseg * sizeof(Segment));
    dev_h2d(up_entries_dev, wentries.data(), (size_t)nseg * sizeof(SegEntry));
    lz77_qspec_parse_custom(P_, B_, Q_, S_, wsegs_dev, up_entries_dev, got_exits_dev, nseg);
    dev_d2h(wexits.data(), got_exits_dev, (size_t)nseg * sizeof(SegExit));
    dev_free(wsegs_dev);
    for (uint32_t k = 0; k + 1 < nseg; ++k) {
      const SegExit& x = wexits[k];
      SegEntry& n = entries_[k + 1];
      int32_t out_cache[4];
      for (uint32_t i = 0; i < 4; ++i) out_cache[i] = i < x.n_pushes ? x.cache[i] : entries_[k].cache[i - x.n_pushes];
      memcpy(n.cache, out_cache, sizeof(out_cache));
      if (segments_[k + 1].flags & kSegFirstInBlock) continue;  // (a block is entered at its start)
      n.pos = x.pos;
      n.apply = x.apply;
      n.head_kind = x.tail_kind;
      n.head_base = x.tail_base;
      n.head_p1 = x.tail_p1;
    }
    stamp("qs-warmed-up");
  }
  // ---- round 0: every segment
  dev_h2d(B_.entries, entries_.data(), (size_t)nseg * sizeof(SegEntry));
  if (own) lz77_qspec_block_tables(P_, B_, Q_, S_, nullptr, nseg, own_tables, own_stride);
  lz77_qspec_parse(P_, B_, Q_, S_, nullptr, nseg, own ? own_tables : nullptr, own_stride);
  if (own) {
    lz77_qspec_first_change(B_, S_, nullptr, nseg, first_change_dev);
    dev_d2h_async(first_change.data(), first_change_dev, 4);
  }
  stats_.rounds++;
  stats_.segments_parsed += nseg;
  lap("round 0 parsed");
  dev_d2h(exits_.data(), B_.exits, (size_t)nseg * sizeof(SegExit));
  stamp("qs-round0");
  // (read per call: the tests switch them)
  const uint32_t kMaxRounds = getenv("BROTLI_MI355X_QUICK_ROUNDS") ? (uint32_t)atoi(getenv("BROTLI_MI355X_QUICK_ROUNDS")) : 64u;
  const bool never_incremental = getenv("BROTLI_MI355X_QUICK_NO_INCREMENTAL") != nullptr;
  const bool selftest = getenv("BROTLI_MI355X_SELFTEST") != nullptr;
  // the candidates follow the flags in proportion to the changes when a launch was short and changed few filings (lz77_qspec_diff /
  // _repair), else by the pass over everything
  bool settled = false, restart_coarse = false, coarse_candidate = false;
  std::vector<uint32_t> dirty_history;  // segments listed in round 1, 2, ...
  bool diffed = false;  // the last launch was followed by lz77_qspec_diff: chg_count[0] holds the number of listed events
  uint32_t incremental_rounds = 0;
  for (uint32_t round = 1; round <= kMaxRounds + (can_own ? nseg + 8u : 0u); ++round) {
    // candidates of the flags as they are now; the chains that searched a position whose candidates changed (device), beside the
    // resolver pass over the exits (host)
    dev_memset(dirty_dev, 0, nseg);
    bool repaired = false;
    if (own) {
      lz77_qspec_candidates(P_, B_, Q_, S_, nullptr, nullptr);
    } else if (diffed && chg_count[0] <= S_.chg_cap) {
      lz77_qspec_repair(P_, B_, Q_, S_, chg_count[0], geo, dirty_dev);
      dev_d2h_async(chg_count.data(), S_.chg_count, 64);
      repaired = true;
    } else {
      lz77_qspec_candidates(P_, B_, Q_, S_, &geo, dirty_dev);
    }
    dev_d2h_async(dirty.data(), dirty_dev, nseg);
    touch_all_ = true;
    Resolve(false);
    dev_sync();
    if (repaired && chg_count[1] != 0) {
      // a walk gave up (a long stretch of inactive filings in one slot): the pass over everything, which starts from the flags
      // (the marks of the repairs that were made stay: those candidates are up to date already and will not be seen to change)
      lz77_qspec_candidates(P_, B_, Q_, S_, &geo, dirty_dev);
      dev_d2h(dirty.data(), dirty_dev, nseg);
      repaired = false;
    }
    if (repaired) incremental_rounds++;
    if (own) {
      // Everything in front of block own_from is final: parsed from its true entry on a table that stood for the final filings of all
      // that lies in front of it.  The launch before this one parsed [own_from, own_upto): final now are the blocks up to the one in
      // which a filing changed (the blocks behind it looked into tables that stood for the old one), or up to the first block whose
      // entry the resolver has just found to be another.  From there the next own_window blocks are parsed again -- each launch moves
      // the frontier by at least one block, and by as many as entered in their true state.  (dirty[] is zero here: nothing was marked
      // by candidates.)
      uint32_t from = first_change[0] != 0xffffffffu ? segment_of(first_change[0]) + 1u : own_upto;
      for (uint32_t k = own_from; k < from && k < nseg; ++k)
        if (dirty_entry_[k]) from = k;
      own_from = from < nseg ? from : nseg;
      own_upto = own_from + own_window < nseg ? own_from + own_window : nseg;
      for (uint32_t k = 0; k < nseg; ++k) dirty_entry_[k] = 0;
      for (uint32_t k = own_from; k < own_upto; ++k) dirty[k] = 1;
    }
    uint32_t count = 0, by_entry = 0;
    for (uint32_t k = 0; k < nseg; ++k) {
      if (!dirty_entry_[k] && !dirty[k]) continue;
      by_entry += dirty_entry_[k] ? 1u : 0u;
      entries_[k] = next_entries_[k];
      R.up_index[count] = k;
      R.up_entries[count] = entries_[k];
      ++count;
    }
    if (debug) {
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
      uint32_t by_cache = 0, by_rest = 0, first = nseg;
      for (uint32_t k = 0; k < nseg; ++k)
        if (dirty_entry_[k]) {
          by_cache += (entry_reason_[k] & 2) ? 1u : 0u;
          by_rest += (entry_reason_[k] & 1) ? 1u : 0u;
          if (first == nseg) first = k;
        }
      if (getenv("BROTLI_MI355X_DEBUG_FIRST") && first < nseg) {
        for (uint32_t k = first > 0 ? first - 1 : 0; k < std::min(nseg, first + 3); ++k) {
          const SegEntry& u = entries_[k];
          const SegEntry& w = next_entries_[k];
          const SegExit& x = exits_[k];
          fprintf(stderr, "  seg %u [%u,%u) dirty %d reason %d: used pos %u apply %u cache %d %d %d %d head %u | new pos %u apply %u cache %d %d %d %d head %u | exit pos %u cmds %u pushes %u cache %d %d %d %d ins %u searches %u\n",
                  k, segments_[k].start, segments_[k].end, (int)dirty_entry_[k], (int)entry_reason_[k], u.pos, u.apply, u.cache[0], u.cache[1], u.cache[2], u.cache[3], u.head_kind,
                  w.pos, w.apply, w.cache[0], w.cache[1], w.cache[2], w.cache[3], w.head_kind, x.pos, x.n_cmds, x.n_pushes, x.cache[0], x.cache[1], x.cache[2], x.cache[3], x.insert_len, x.n_searches);
        }
      }
      fprintf(stderr, "quick round %u: %u of %u segments to parse again (%u for their entry: %u distance cache, %u position / spree / dictionary, first %u; %u predicted runs), %.2f ms into the parse%s\n",
              round, count, nseg, by_entry, by_cache, by_rest, first, predicted_runs_, ms, repaired ? " [repaired]" : " [all]");
    }
    if (count == 0) {
      settled = true;
      break;
    }
    // Most chains were entered in another state than the warm-up guessed: the input does not fall into step inside its blocks
    // (literal sprees file every second or fourth position, and from WHICH one on is a matter of where the spree was entered: in
    // incompressible stretches the phase of the chain in front never washes out before the block ends, and every change of it
    // changes the filings, hence the candidates, of everything behind).  It is re-cut into one chain per input block -- nothing is
    // guessed inside a block then -- and the iteration starts over on the same index (RunRounds does the same for qualities 5-9).
    // No progress to speak of over eight rounds (the cascade of changed candidates does not die out: structured binary data at four
    // slots per key can need more rounds than it has segments' worth of positions): the serial walk takes over now rather than after
    // all kMaxRounds launches.
    dirty_history.push_back(count);
    if (!own && (round >= kMaxRounds || (round >= 16 && count > std::max<uint32_t>(8u, nseg / 64u) && (uint64_t)count * 10 > (uint64_t)dirty_history[round - 9] * 8))) {
      if (!qspec_coarse_ && segment_bytes_ < block_bytes_ && nseg >= 64 && getenv("BROTLI_MI355X_QUICK_NO_COARSE") == nullptr) {
        restart_coarse = true;
        break;
      }
      if (!can_own) break;
      // the blocks in front of the first one with a reason to be parsed again are final; from there on tables of their own
      uint32_t first = 0;
      while (first < nseg && !dirty_entry_[first] && !dirty[first]) ++first;
      if (debug) fprintf(stderr, "quick: no progress to speak of, the blocks from %u on go on tables of their own\n", first);
      own = true;
      own_from = own_upto = first;
      first_change[0] = 0xffffffffu;
      stats_.coarse_restarts++;
      continue;
    }
    // (decided behind the second launch: Silesia-like pieces enter 60 % of their chains in another state than guessed and are down
    // to 1-2 % of the segments one launch later; the inputs meant here still have 90 % of them to parse again)
    if (round == 1) coarse_candidate = !qspec_coarse_ && segment_bytes_ < block_bytes_ && nseg >= 64 && (uint64_t)by_entry * 3 > nseg && getenv("BROTLI_MI355X_QUICK_NO_COARSE") == nullptr;
    if (round == 2 && coarse_candidate && (uint64_t)count * 10 > (uint64_t)dirty_history[0] * 6) {
      restart_coarse = true;
      break;
    }
    stats_.rounds++;
    stats_.segments_parsed += count;
    dev_h2d(list_dev, R.up_index.data(), (size_t)count * 4);
    dev_h2d(up_entries_dev, R.up_entries.data(), (size_t)count * sizeof(SegEntry));
    lz77_scatter_entries(B_, list_dev, up_entries_dev, count);
    if (own) lz77_qspec_block_tables(P_, B_, Q_, S_, list_dev, count, own_tables, own_stride);
    lz77_qspec_parse(P_, B_, Q_, S_, list_dev, count, own ? own_tables : nullptr, own_stride);
    lz77_qspec_gather_exits(B_, list_dev, count, got_exits_dev);
    if (own) {
      lz77_qspec_first_change(B_, S_, list_dev, count, first_change_dev);
      dev_d2h_async(first_change.data(), first_change_dev, 4);
    }
    diffed = !never_incremental && !own;
    if (diffed) {
      lz77_qspec_diff(P_, B_, Q_, S_, list_dev, count);
      dev_d2h_async(chg_count.data(), S_.chg_count, 64);
    }
    dev_d2h(R.got_exits.data(), got_exits_dev, (size_t)count * sizeof(SegExit));
    for (uint32_t i = 0; i < count; ++i) exits_[R.up_index[i]] = R.got_exits[i];
  }
  stats_.incremental_ranks += incremental_rounds;
  stamp("qs-settled");
  if (settled && selftest) {
    // the candidates as the repairs of the rounds left them against the pass over everything from the final flags
    const size_t words = (size_t)S_.n * Q_.sweep;
    std::vector<uint32_t> kept(words), fresh(words);
    dev_d2h(kept.data(), S_.cand, words * 4);
    lz77_qspec_candidates(P_, B_, Q_, S_, nullptr, nullptr);
    dev_d2h(fresh.data(), S_.cand, words * 4);
    for (size_t i = 0; i < words; ++i)
      if (kept[i] != fresh[i])
        throw std::runtime_error("selftest: candidate " + std::to_string(i % Q_.sweep) + " of position " + std::to_string(i / Q_.sweep) + " is " + std::to_string(kept[i]) +
                                 " after the repairs of " + std::to_string(incremental_rounds) + " rounds, " + std::to_string(fresh[i]) + " from the final flags");
  }
  dev_free(geo_tables);
  dev_free(dirty_dev);
  dev_free(own_tables);
  dev_free(first_change_dev);
  dev_free(list_dev);
  dev_free(up_entries_dev);
  dev_free(got_exits_dev);
  if (restart_coarse) {
    saved_block_guess_.clear();
    for (uint32_t k = 0; k < nseg; ++k)
      if (segments_[k].flags & kSegFirstInBlock) saved_block_guess_[segments_[k].blk_start] = next_entries_[k];
    if (debug) fprintf(stderr, "quick: the parse does not fall into step inside the blocks, one chain per block from here\n");
    coarse_blocks_.assign(block_segment_bytes_.size(), 1);
    Resegment(segment_bytes_);
    qspec_coarse_ = true;
    stats_.coarse_restarts++;
    RunQuickSpec();
    return;
  }
  if (!settled) {
    // the serial path from the start: one segment per block on the reference's own table
    if (debug) fprintf(stderr, "quick: not settled after %u rounds, the serial path takes over\n", (unsigned)dirty_history.size());
    use_qspec_ = false;
    P_.use_dictionary = 0;
    Resegment(block_bytes_);
    dev_free(B_.entries);
    dev_free(B_.exits);
    dev_free(gather_offsets_dev_);
    dev_free(gather_counts_dev_);
    B_.entries = (SegEntry*)dev_alloc(segments_.size() * sizeof(SegEntry) + 64);
    B_.exits = (SegExit*)dev_alloc(segments_.size() * sizeof(SegExit) + 64);
    gather_offsets_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
    gather_counts_dev_ = (uint32_t*)dev_alloc(segments_.size() * 4 + 64);
    stats_.coarse_restarts++;
    RunQuick();
    return;
  }
  final_flags_ = 0;
  for (uint32_t k = 0; k < nseg; ++k) stats_.searches += exits_[k].n_searches;
}

// Qualities 10 / 11: the blocks go through the device one after the other (the H10 trees and the dynamic programme are a
// function of everything in front); between two of them the host resolver replays the exits -- flush rule, should_compress,
// extend_last_command -- exactly as behind a round of the greedy path, and hands the next block its entry.
void Lz77Stage::RunZopfli() {
  const bool debug = getenv("BROTLI_MI355X_DEBUG") != nullptr;
  const uint32_t nseg = (uint32_t)segments_.size();
  if (P_.reset_pos != 0) throw std::runtime_error("brotli_mi355x: quality 10 / 11 across the reference's 32-bit position wrap is not supported");
  InitEntries();
  lz77_compute_keys(P_, B_);
  lz77_sort_by_key(P_, B_);
  const size_t forest_bytes = ((size_t)2 << Z_.lgwin) * 4, bucket_bytes = ((size_t)1 << 17) * 4;
  if (carry_ && carry_->valid && carry_->zopfli) {
    // a later piece of a stream: the trees of the piece in front, moved to this piece's text positions
    const ZopfliCarry& zc = *carry_->zopfli;
    if (zc.lgwin != Z_.lgwin || carry_->stream_base < zc.text_base) throw std::runtime_error("brotli_mi355x: quality 10 / 11: the carried trees do not fit this piece");
    lz77_zopfli_import(Z_, zc.buckets, zc.forest, (uint32_t)(carry_->stream_base - zc.text_base));
  } else {
    lz77_zopfli_init(Z_);
    // a custom dictionary (the prefix of a compress_multi shard) goes into the trees first (encode.rs:1163-1194); nothing of the
    // stream itself can lie in front of a piece that brings no trees
    if (P_.prefix_bytes > 1) lz77_zopfli_prepend(P_, B_, Z_, P_.prefix_bytes);
  }
  bool starts_metablock = true;  // (the block about to be parsed is the first of its meta-block)
  uint32_t from = 0;
  bool done = false;
  for (uint32_t round = 0; round < nseg + 8 && !done; ++round) {
    stats_.rounds++;
    if (round != 0) entries_[from] = next_entries_[from];
    if (partial_ && starts_metablock) {
      // the next piece starts again at the first block of the meta-block that is still open when this one ends: the trees as
      // they are in front of every block that opens a meta-block (the last such copy is the one that travels)
      if (!zsnap_buckets_) {
        zsnap_buckets_ = (uint32_t*)dev_alloc_uninit(bucket_bytes + 64);
        zsnap_forest_ = (uint32_t*)dev_alloc_uninit(forest_bytes + 64);
      }
      dev_d2d(zsnap_buckets_, Z_.buckets, bucket_bytes);
      dev_d2d(zsnap_forest_, Z_.forest, forest_bytes);
    }
    dev_h2d(B_.entries + from, entries_.data() + from, sizeof(SegEntry));
    if (lz77_zopfli_block(P_, B_, Z_, from)) stats_.coarse_restarts++;  // (parsed the sequential way)
    stats_.segments_parsed++;
    dev_d2h(exits_.data() + from, B_.exits + from, sizeof(SegExit));
    if (exits_[from].bad_commands != 0)
      throw std::runtime_error("brotli_mi355x: the reference encoder fails on this input at quality 10 / 11 (it indexes past the end of an array)");
    Resolve(false);
    // Everything up to `from` is final: its entries came out of exits that were final already (the trees do not hang on the
    // parse, and what a meta-block in front decides -- stored uncompressed: it hands on the distance cache of its start,
    // encode.rs:1994 -- was decided from the same exits).  The blocks behind are unparsed, whatever their empty exit records
    // make the resolver think of them.
    for (uint32_t k = 0; k <= from; ++k)
      if (dirty_entry_[k]) throw std::runtime_error("brotli_mi355x: resolver and Zopfli parse disagree about the entry of a finished block");
    if (debug) fprintf(stderr, "zopfli block %u of %u done: %u commands, %u bytes pending\n", from, nseg, exits_[from].n_cmds, exits_[from].insert_len);
    starts_metablock = false;
    for (const MetaBlockPlan& mb : metablocks_)
      if (mb.end == segments_[from].blk_end) starts_metablock = true;  // (the flush rule closed a meta-block behind this block)
    ++from;
    done = from == nseg;
  }
  if (!done) throw std::runtime_error("brotli_mi355x: quality 10 / 11 parse did not finish");
  if (partial_ && starts_metablock) {  // (the flush rule closed a meta-block with the last block: the next piece starts behind it)
    if (!zsnap_buckets_) {
      zsnap_buckets_ = (uint32_t*)dev_alloc_uninit(bucket_bytes + 64);
      zsnap_forest_ = (uint32_t*)dev_alloc_uninit(forest_bytes + 64);
    }
    dev_d2d(zsnap_buckets_, Z_.buckets, bucket_bytes);
    dev_d2d(zsnap_forest_, Z_.forest, forest_bytes);
  }
  final_flags_ = 0;
  for (uint32_t k = 0; k < nseg; ++k) stats_.searches += exits_[k].n_searches;
}

void Lz77Stage::RunRounds(bool allow_restart) {
  if (use_live_) {
    RunLive();
    return;
  }
  const bool prof = getenv("BROTLI_MI355X_PROFILE") != nullptr;
  Timer tm(prof);
  const uint32_t nseg = (uint32_t)segments_.size();
  // the resolver passes of this loop look only at the blocks whose segments were handed new entries or exits (TouchSegment)
  resolve_incremental_ = true;
  block_records_valid_ = false;
  touch_all_ = true;
  block_touched_.assign(block_segment_bytes_.size(), 0);
  InitFlags();
  tm.stop(&stats_.ms_init);
  const bool selftest = getenv("BROTLI_MI355X_SELFTEST") != nullptr;
  // (the candidate rows against a rebuild from the flags after EVERY update, not only the first build: minutes per MiB)
  const bool selftest_rows_every_update = getenv("BROTLI_MI355X_SELFTEST_ROWS") != nullptr;
  if (selftest) SelfTestSort();
  int which = 0, rbuf = 0;
  // (the index build is queued first: the device works on it while the host prepares the entries)
  {
    RankInitialHint hint{segments_[0].blk_start, block_bytes_, (carry_ && carry_->valid) ? 0u : 1u};
    if (use_rows_) {
      lz77_rows_init(P_, B_, which, &hint, has_big_keys_);
    } else {
      lz77_rank_flags(P_, B_, which, rbuf, &hint);
    }
  }
  InitEntries();
  tm.stop(&stats_.ms_rank);
  timeline().stamp("index-queued");
  if (selftest) {
    if (use_rows_) SelfTestRows(which); else SelfTestRank(which, rbuf);
  }
  // ---- warm-up: a dry run over the tail of every segment gives a good first guess of the state in which the
  // parse leaves it (greedy parses re-synchronise quickly), so that the first full round already starts
  // almost every chain from its true entry.
  if (const char* w = getenv("BROTLI_MI355X_WARMUP")) warmup_bytes_ = (uint32_t)atoi(w);
  double fc_L = 0, fc_M = 0;  // the forecast's running sums
  uint32_t fc_death = nseg;
  bool fc_found = false;
  // forecast of the segment in which the static dictionary gets switched off (matches < lookups >> 7,
  // mod.rs:1957-1960); chains behind it start with the "off" guess.  A wrong forecast only costs re-parses of
  // the chains that actually used / could have used a dictionary match (DictTracker::Consume).
  // Guessing "off" too early is the expensive mistake (a chain that ran without the dictionary cannot tell what
  // it would have counted, so the exact counters can only advance one redone chain per round), guessing "on" too
  // long only costs the re-parse of the chains that really used a dictionary match.  Hence a forecast three
  // standard deviations of the sampled match count on the late side.
  auto forecast = [&](uint32_t from, uint32_t upto) {  // segments [from, upto), in order
    if (!(P_.use_dictionary && warm_lookups_.size() == nseg)) return;
    const double sample = (double)warmup_bytes_ / (double)segment_bytes_;
    for (uint32_t k = from; k < upto && !fc_found; ++k) {
      fc_L += warm_lookups_[k];
      fc_M += warm_matches_[k];
      const double sigma = std::sqrt(std::max(1.0, fc_M * sample)) / sample;
      if (fc_L >= 256.0 && fc_M + 3.0 * sigma < fc_L / 128.0) {
        fc_death = k;
        fc_found = true;
      }
    }
    if (upto == nseg || fc_found) {
      for (uint32_t k = fc_death + 1; k < nseg; ++k) {
        entries_[k].dict_lookups = DictTracker::kDeadL;
        entries_[k].dict_matches = DictTracker::kDeadM;
      }
      touch_all_ = true;
      predicted_death_ = fc_death;
    }
  };
  if (nseg > 1 && warmup_bytes_ > 0) {
    Warmup(0, false, which, rbuf, nullptr);
    timeline().stamp("warmed-up");
    forecast(0, nseg);
    tm.stop(&stats_.ms_warmup);
  }
  if (!allow_restart && !saved_block_guess_.empty()) {
    // second pass after a coarse restart: the first pass chained the distance cache through every segment of the
    // input, which beats the dry run's view of the last few hundred bytes of each block
    for (uint32_t k = 0; k < nseg; ++k) {
      if (!(segments_[k].flags & kSegFirstInBlock)) continue;
      auto it = saved_block_guess_.find(segments_[k].blk_start);
      if (it == saved_block_guess_.end()) continue;
      memcpy(entries_[k].cache, it->second.cache, sizeof(entries_[k].cache));
      entries_[k].ext_allowed = it->second.ext_allowed;
      touch_all_ = true;
    }
  }
  // ---- rounds.  Round 0 parses every segment; later rounds re-parse only the segments whose entry state
  // changed or that searched a position whose candidate list changed with the flags (lz77_validate).
  SegGeometry geo{};
  geo.prefix_bytes = P_.prefix_bytes;
  geo.first_block_start = segments_[0].blk_start;
  geo.block_bytes = block_bytes_;
  geo.num_blocks = (uint32_t)block_segment_bytes_.size();
  uint32_t* geo_tables = (uint32_t*)dev_alloc((block_first_segment_.size() + block_segment_bytes_.size()) * 4 + 64);
  dev_h2d(geo_tables, block_first_segment_.data(), block_first_segment_.size() * 4);
  dev_h2d(geo_tables + block_first_segment_.size(), block_segment_bytes_.data(), block_segment_bytes_.size() * 4);
  geo.block_first_segment = geo_tables;
  geo.block_segment_bytes = geo_tables + block_first_segment_.size();
  geo.num_segments = nseg;
  geo.block_size = 1u << P_.block_bits;
  uint8_t* dirty_dev = (uint8_t*)dev_alloc(nseg + 64);
  uint32_t* list_dev = (uint32_t*)dev_alloc((size_t)nseg * 8 + 64);  // (listed segments + entries accepted by RecheckCacheOnly)
  std::vector<uint8_t> dirty(nseg, 0);
  std::vector<uint32_t> list(nseg);
  // list rounds: compact transfers (lz77_scatter_entries / lz77_gather_results)
  static constexpr uint32_t kContFirst = 64;
  const uint32_t cont_cap = std::max(nseg, kContFirst);
  SegEntry* up_entries_dev = (SegEntry*)dev_alloc_uninit((size_t)nseg * 2 * sizeof(SegEntry) + 64);
  SegExit* got_exits_dev = (SegExit*)dev_alloc_uninit((size_t)nseg * sizeof(SegExit) + 64);
  uint32_t* cont_count_dev = (uint32_t*)dev_alloc(64);
  uint32_t* cont_index_dev = (uint32_t*)dev_alloc((size_t)cont_cap * 4 + 64);
  SegExit* cont_exits_dev = (SegExit*)dev_alloc_uninit((size_t)cont_cap * sizeof(SegExit) + 64);
  SegEntry* cont_entries_dev = (SegEntry*)dev_alloc_uninit((size_t)cont_cap * sizeof(SegEntry) + 64);
  std::vector<uint32_t> also_upload;
  // bursts (device_api.h): up to burst_max list launches per pass of the host resolver, scheduled on the device in between
  static const uint32_t burst_env = getenv("BROTLI_MI355X_BURST") ? (uint32_t)atoi(getenv("BROTLI_MI355X_BURST")) : 8u;
  static const uint32_t splice_max_share = getenv("BROTLI_MI355X_SPLICE_SHARE") ? (uint32_t)atoi(getenv("BROTLI_MI355X_SPLICE_SHARE")) : 8u;
  static const uint32_t burst_shrink = getenv("BROTLI_MI355X_BURST_SHRINK") ? (uint32_t)atoi(getenv("BROTLI_MI355X_BURST_SHRINK")) : 4u;
  // 0: one launch per pass, scheduled by the host -- rank-structure chains, and inputs with long runs of one byte (zero fill:
  // every launch there is followed by passes over all rows, and the device's plain scheduling needs more of them: 1 GiB of
  // zeros 6 rounds / 736 ms with bursts, 4 rounds / 467 ms without)
  const uint32_t burst_max = (use_rows_ && B_.run_end == nullptr) ? burst_env : 0u;
  BurstBuffers U;
  bool stale_marks_valid = false;  // U.stale says where the two flag arrays differ (from the first launch of the first burst on)
  if (burst_max != 0) {
    U.sched = (uint8_t*)dev_alloc(nseg + 64);
    U.cand_dirty = dirty_dev;
    U.entry_dirty = (uint8_t*)dev_alloc(nseg + 64);
    U.touched = (uint8_t*)dev_alloc(nseg + 64);
    U.stale = (uint8_t*)dev_alloc(nseg + 64);
    U.new_entries = (SegEntry*)dev_alloc_uninit((size_t)nseg * sizeof(SegEntry) + 64);
    U.list = (uint32_t*)dev_alloc_uninit((size_t)nseg * 4 + 64);
    U.counters = (uint32_t*)dev_alloc(64);
  }
  RoundBuffers& rb = round_buffers_;  // page-locked, kept from call to call
  rb.up_index.resize_discard((size_t)nseg * 2);
  rb.up_entries.resize_discard((size_t)nseg * 2);
  rb.got_exits.resize_discard(nseg);
  rb.cont_index.resize_discard(cont_cap);
  rb.cont_exits.resize_discard(cont_cap);
  rb.cont_entries.resize_discard(cont_cap);
  rb.counts.resize_discard(4);
  uint32_t* up_index = rb.up_index.data();
  SegEntry* up_entries = rb.up_entries.data();
  SegExit* got_exits = rb.got_exits.data();
  uint32_t* cont_index = rb.cont_index.data();
  SegExit* cont_exits = rb.cont_exits.data();
  SegEntry* cont_entries = rb.cont_entries.data();
  volatile uint32_t* counts = rb.counts.data();  // [0] flag changes, [1] segments continued into
  const bool debug = getenv("BROTLI_MI355X_DEBUG") != nullptr;
  std::vector<uint32_t> changed_all(kChangedCap);
  std::vector<uint8_t> entry_streak(nseg, 0), was_dirty, cand_dirty, pending(nseg, 0), sched(nseg, 0);
  // cache_wave[k]: segment k was parsed in the last launch from an entry that differed from its previous one in the
  // distance cache alone (by a chain of its own, or by a chain that walked into it) -- see "cache front" below
  std::vector<uint8_t> cache_wave(nseg, 0), cache_wave_next(nseg, 0);
  const bool follow_cache_fronts = getenv("BROTLI_MI355X_NO_CACHE_FRONTS") == nullptr;
  for (uint32_t k = 0; k < nseg; ++k) list[k] = k;
  uint32_t count = nseg;
  // (every round fixes at least the first segment that was still wrong, so nseg rounds always suffice; the override
  // makes an experiment that does not converge fail fast)
  const uint32_t max_rounds = getenv("BROTLI_MI355X_MAX_ROUNDS") ? (uint32_t)atoi(getenv("BROTLI_MI355X_MAX_ROUNDS")) : nseg + 8;
  bool done = false;
  bool restart = false;
  bool full_round = true;
  uint32_t last_death_seg = 0xffffffffu;
  auto stamp = [&](const char* what) { timeline().stamp(what); };  // BROTLI_MI355X_TIMELINE
  for (uint32_t round = 0; round < max_rounds && !done; ++round) {
    stats_.rounds++;
    stamp("| round");
    tm.stop(&stats_.ms_resolve);
    // upload the entries of the segments to parse, run them, fetch their exits
    if (full_round) {
      dev_h2d(B_.entries, entries_.data(), (size_t)nseg * sizeof(SegEntry));
    } else {
      // only the entries the host changed since the last launch: those of the listed segments and those whose distance
      // cache RecheckCacheOnly() accepted
      const uint32_t n_up = count + (uint32_t)also_upload.size();
      for (uint32_t i = 0; i < n_up; ++i) {
        up_index[i] = i < count ? list[i] : also_upload[i - count];
        up_entries[i] = entries_[up_index[i]];
      }
      dev_h2d(list_dev, up_index, (size_t)n_up * 4);
      dev_h2d(up_entries_dev, up_entries, (size_t)n_up * sizeof(SegEntry));
      lz77_scatter_entries(B_, list_dev, up_entries_dev, n_up);
      // (segments whose parse was accepted for another distance cache: their checkpoints hold the old one)
      if (n_up > count) lz77_drop_checkpoints(P_, B_, list_dev + count, n_up - count);
      dev_h2d(burst_max != 0 ? U.sched : dirty_dev, sched.data(), nseg);
    }
    also_upload.clear();
    const bool burst = !full_round && burst_max != 0;
    uint32_t n_touched = 0;
    if (burst) {
      // launch, bring the rows up to date, chain the exits on the device, schedule the next launch there; the host only
      // waits for the length of the next list
      const uint32_t* launch_list = list_dev;
      uint32_t launch_count = count;
      for (uint32_t it = 0;; ++it) {
        // Chains that restart from / stop at checkpoints carry more state (the kernel keeps ~180 more scalars in spill slots):
        // worth it where most of a re-parse is saved -- launches of a few per cent of the segments, as on text -- and a loss
        // where a fifth of the input is parsed again every round by chains that never fall back into step (pieces of a mix).
        B_.splice_lists = (uint64_t)launch_count * splice_max_share <= nseg ? 1u : 0u;
        // flags[which ^ 1] := flags[which]: everything after a full round, afterwards only what the last launch parsed (U.stale)
        if (stale_marks_valid) {
          lz77_flags_catch_up(P_, B_, U, which, which ^ 1);
        } else {
          dev_d2d(B_.flags[which ^ 1], B_.flags[which], (size_t)P_.total_bytes + 64);
          dev_memset(U.stale, 0, nseg);
          stale_marks_valid = true;
        }
        lz77_parse_list(P_, B_, which, rbuf, launch_list, U.sched, launch_count);
        stats_.segments_parsed += launch_count;
        lz77_chain_check(P_, B_, U);  // (also resets the rows-changed marks of what was parsed: before the marks of this launch's flips)
        lz77_diff_flags_touched(P_, B_, U, which, which ^ 1);
        dev_memset(dirty_dev, 0, nseg);
        lz77_rows_update(P_, B_, which, which ^ 1, geo, dirty_dev, has_big_keys_);
        which ^= 1;
        stats_.burst_launches++;
        if (selftest_rows_every_update) SelfTestRows(which);
        if (it + 1 >= burst_max) break;
        lz77_burst_count(P_, B_, U);
        dev_d2h((void*)counts, U.counters, 4);
        const uint32_t next_count = counts[0];
        if (debug) fprintf(stderr, "  burst launch %u: %u segments next\n", it, next_count);
        // The device goes on by itself only while the parse is settling fast (text: 2 900 -> 7 -> 0 segments).  Where every
        // launch leaves about as many segments dirty as it parsed (pieces of a mix that never fall back into step), parsing
        // all of them again from entries chained out of parses that are themselves about to be redone only adds noise: the
        // host resolver, which leaves such segments to the chain in front of them, takes over.  (The marks of this launch
        // are still on the device: nothing was cleared.)
        if (next_count == 0 || (uint64_t)next_count * burst_shrink > launch_count) break;
        lz77_burst_schedule(P_, B_, U);
        launch_list = U.list;
        launch_count = next_count;
      }
      which ^= 1;  // (flags[which ^ 1] holds the newest flags; the common code below flips)
      lz77_gather_touched(P_, B_, U, cont_index_dev, cont_exits_dev, cont_entries_dev);
      dev_d2h((void*)counts, U.counters, 8);
      n_touched = counts[1];
      dev_d2h_async(cont_index, cont_index_dev, (size_t)n_touched * 4);
      dev_d2h_async(cont_exits, cont_exits_dev, (size_t)n_touched * sizeof(SegExit));
      dev_d2h_async(cont_entries, cont_entries_dev, (size_t)n_touched * sizeof(SegEntry));
      dev_sync();
      stamp("burst-done");
    } else {
    dev_d2d(B_.flags[which ^ 1], B_.flags[which], (size_t)P_.total_bytes + 64);
    stamp("uploaded");
    if (full_round) {
      lz77_parse_round(P_, B_, which, rbuf, 0);
    } else {
      lz77_parse_list(P_, B_, which, rbuf, list_dev, dirty_dev, count);
    }
    stats_.segments_parsed += count;
    lz77_diff_flags(P_, B_, which, which ^ 1);
    }
    if (getenv("BROTLI_MI355X_DEBUG_FLAGS")) {
      std::vector<uint8_t> fa(P_.total_bytes), fb(P_.total_bytes);
      dev_d2h(fa.data(), B_.flags[which], P_.total_bytes);
      dev_d2h(fb.data(), B_.flags[which ^ 1], P_.total_bytes);
      uint32_t shown = 0;
      for (uint32_t q = 0; q < P_.total_bytes && shown < 40; ++q)
        if ((fa[q] ^ fb[q]) & 1) {
          fprintf(stderr, "  flag change at %u (seg %u, off %u): %u -> %u\n", q, q / segment_bytes_, q % segment_bytes_, fa[q], fb[q]);
          shown++;
        }
    }
    // everything the host needs from this launch in one round trip: exit records, the change list, and -- in list
    // rounds -- the entries and marks of the segments that chains continued into
    if (burst) {
      // the exits and entries of every segment parsed during the burst (the device rewrote entries it chained itself)
      for (uint32_t i = 0; i < n_touched; ++i) {
        const uint32_t k = cont_index[i];
        pending[k] = 0;
        exits_[k] = cont_exits[i];
        entries_[k] = cont_entries[i];
        TouchSegment(k);
      }
      counts[0] = 1;  // (read the marks of the last lz77_rows_update below)
      counts[1] = 0;
    } else {
    counts[0] = counts[1] = 0;
    if (full_round) {
      dev_d2h_async(exits_.data(), B_.exits, (size_t)nseg * sizeof(SegExit));
      touch_all_ = true;
    } else {
      // the exits of the listed segments, and exit + rewritten entry of the segments that chains continued into (the
      // first kContFirst of them ride along; more are rare and fetched afterwards)
      lz77_gather_results(B_, list_dev, count, dirty_dev, nseg, got_exits_dev, cont_count_dev, cont_index_dev, cont_exits_dev, cont_entries_dev);
      dev_d2h_async(got_exits, got_exits_dev, (size_t)count * sizeof(SegExit));
      dev_d2h_async((void*)(counts + 1), cont_count_dev, 4);
      dev_d2h_async(cont_index, cont_index_dev, (size_t)kContFirst * 4);
      dev_d2h_async(cont_exits, cont_exits_dev, (size_t)kContFirst * sizeof(SegExit));
      dev_d2h_async(cont_entries, cont_entries_dev, (size_t)kContFirst * sizeof(SegEntry));
    }
    dev_d2h_async((void*)counts, B_.changed_count, 4);
    if (!use_rows_) dev_d2h_async(changed_all.data(), B_.changed_keys, (size_t)kChangedCap * 4);
    if (use_rows_) {
      // the rows are brought up to date on the device (it decides by itself between the incremental and the full
      // rebuild) while the host chains the exits together: wait for the copies only
      dev_mark();
      dev_memset(dirty_dev, 0, nseg);
      stamp("launched");
      if (full_round) lz77_reset_rows_changed(P_, B_);  // (everything was parsed)
      lz77_rows_update(P_, B_, which, which ^ 1, geo, dirty_dev, has_big_keys_);
      stamp("rows-queued");
      dev_wait_mark();
      stamp("exits-here");
    } else {
      dev_sync();
    }
    }
    const uint32_t n_changed = counts[0], n_cont = counts[1];
    if (!full_round && !burst) {
      for (uint32_t i = 0; i < count; ++i) {
        exits_[list[i]] = got_exits[i];
        TouchSegment(list[i]);
      }
      // chains that kept going into unscheduled segments (br_parse_chain) rewrote the entries of those
      if (n_cont > kContFirst) {
        dev_d2h(cont_index, cont_index_dev, (size_t)n_cont * 4);
        dev_d2h(cont_exits, cont_exits_dev, (size_t)n_cont * sizeof(SegExit));
        dev_d2h(cont_entries, cont_entries_dev, (size_t)n_cont * sizeof(SegEntry));
      }
      if (debug) fprintf(stderr, "  continued into %u segments\n", n_cont);
      for (uint32_t i = 0; i < n_cont; ++i) {
        const uint32_t k = cont_index[i];
        pending[k] = 0;
        exits_[k] = cont_exits[i];
        entries_[k] = cont_entries[i];
        TouchSegment(k);
        cache_wave[k] = 1;
        stats_.segments_parsed++;
      }
    }
    tm.stop(&stats_.ms_parse);
    which ^= 1;  // flags[which] now holds the newest flags
    if (selftest_rows_every_update && use_rows_) SelfTestRows(which);
    // the rank structures are brought up to date on the device while the host chains the exits together
    if (use_rows_) {
      if (n_changed != 0) (n_changed <= B_.changed_cap ? stats_.incremental_ranks : stats_.full_ranks)++;
    } else if (n_changed != 0) {
      // few changes: re-rank only the keys concerned, in place; otherwise rebuild everything into the other buffer
      // and diff the two
      std::vector<uint32_t> changed;
      bool incremental = false;
      if (n_changed <= kChangedCap) {
        changed.assign(changed_all.begin(), changed_all.begin() + n_changed);
        std::sort(changed.begin(), changed.end());
        changed.erase(std::unique(changed.begin(), changed.end()), changed.end());
        uint64_t affected = 0;
        uint32_t widest = 0;
        for (uint32_t key : changed) {
          const uint32_t width = key_last_[key] - key_first_[key];
          affected += width;
          widest = std::max(widest, width);
        }
        (void)widest;
        static const uint32_t inc_pct = getenv("BROTLI_MI355X_INC_PCT") ? (uint32_t)atoi(getenv("BROTLI_MI355X_INC_PCT")) : 100u;
        incremental = affected <= (uint64_t)P_.total_bytes * inc_pct / 100;
      }
      dev_memset(dirty_dev, 0, nseg);
      if (B_.recheck_count) dev_memset(B_.recheck_count, 0, 4);
      if (incremental) {
        std::vector<RerankChunk> chunks;
        for (uint32_t key : changed) {
          const uint32_t lo = key_first_[key], hi = key_last_[key];
          const uint32_t first_sum = (uint32_t)chunks.size();
          for (uint32_t b = lo; b < hi; b += kRerankChunk)
            chunks.push_back({lo, b, std::min(hi, b + kRerankChunk), first_sum, (uint32_t)chunks.size()});
        }
        RerankChunk* chunks_dev = (RerankChunk*)dev_alloc(chunks.size() * sizeof(RerankChunk) + 64);
        uint32_t* sums_dev = (uint32_t*)dev_alloc(chunks.size() * 4 + 64);
        dev_h2d(chunks_dev, chunks.data(), chunks.size() * sizeof(RerankChunk));
        lz77_rerank_keys(P_, B_, which, rbuf, chunks_dev, (uint32_t)chunks.size(), sums_dev, geo, dirty_dev);
        dev_free(chunks_dev);
        dev_free(sums_dev);
        stats_.incremental_ranks++;
      } else {
        lz77_rank_flags(P_, B_, which, rbuf ^ 1);
        lz77_validate(P_, B_, which, rbuf, rbuf ^ 1, geo, dirty_dev);
        rbuf ^= 1;
        stats_.full_ranks++;
      }
      // (the two above only listed the searched positions whose candidate list changed: their searches are repeated)
      lz77_recheck_searches(P_, B_, rbuf, geo, dirty_dev);
    }
    tm.stop(&stats_.ms_rank);
    if (const char* path = getenv("BROTLI_MI355X_DEBUG_FRONT")) {  // what the launch made of the segments at the frontier
      if (FILE* f = fopen(path, "a")) {
        uint32_t front = 0;  // first segment that was owed a re-parse when this launch was scheduled: all before it are final
        while (!was_dirty.empty() && front < nseg && !was_dirty[front]) ++front;
        fprintf(f, "R %u %u", round, front);
        for (uint32_t k = front; k < nseg && k < front + 24; ++k)
          fprintf(f, " %u:%u:%u:%u", k, exits_[k].pos, exits_[k].n_cmds, round == 0 ? 1u : (uint32_t)sched[k]);
        fprintf(f, "\n");
        fclose(f);
      }
    }
    const auto host_t0 = std::chrono::steady_clock::now();
    Resolve(false);
    const auto host_t1 = std::chrono::steady_clock::now();
    host_resolve_ms_ += std::chrono::duration<double, std::milli>(host_t1 - host_t0).count();
    stamp("resolved");
    const uint32_t rechecked = RecheckCacheOnly(which, &also_upload);
    stamp("rechecked");
    stats_.cache_rechecks += rechecked;
    for (uint32_t k = 0; k < nseg; ++k) dirty[k] = dirty_entry_[k] | pending[k];
    uint32_t n_dirty_entry = 0, n_dirty_valid = 0;
    for (uint32_t k = 0; k < nseg; ++k) n_dirty_entry += dirty[k];
    // When most chains were started at the wrong position the input does not re-synchronise (sparse hashing in
    // incompressible data, long runs: the phase of the previous chain never washes out).  It is re-cut into one chain
    // per input block -- nothing is guessed inside a block then -- and the iteration starts over.  (Doing this for the
    // offending blocks only was tried and is worse on mixed content: a single-chain block that is merely re-validated
    // costs a 64 KiB serial parse per round.)
    if (round == 0 && allow_restart && segment_bytes_ < block_bytes_ && nseg >= 64 && (uint64_t)(dbg_counts_[0] - std::min(dbg_counts_[0], predicted_runs_)) * 2 > nseg) {
      coarse_blocks_.assign(block_segment_bytes_.size(), 1);
      restart = true;
      break;
    }
    cand_dirty.clear();
    if (n_changed != 0) {
      cand_dirty.resize(nseg);
      dev_d2h(cand_dirty.data(), dirty_dev, nseg);
      for (uint32_t k = 0; k < nseg; ++k) {
        n_dirty_valid += cand_dirty[k];
        dirty[k] |= cand_dirty[k];
      }
      tm.stop(&stats_.ms_rank);
      stamp("dirty-here");
    }
    // the static dictionary got switched off at segment dict_death_seg_ and many chains behind it ran in the
    // wrong regime: they are re-parsed anyway; refresh their entry guesses with a dry run in the new regime
    const bool regime_flip = dict_death_seg_ != 0xffffffffu && dict_death_seg_ != last_death_seg && dict_flips_ > 64;
    // Which dirty segments get their own chain?  Normally all of them, each started from the entry chained out of its
    // predecessor's last exit (a guess while that predecessor is itself being redone -- fine when parses re-synchronise
    // quickly).  Where they do not (long literal stretches with sparse hashing keep the phase of the previous chain
    // for ever), the guess fails round after round and the fix would creep forward one segment per round.  So a
    // segment whose entry was wrong twice in a row, and whose predecessor is being redone as well, is left to that
    // predecessor's chain, which continues into it with its real exit state (br_parse_chain).
    for (uint32_t k = 0; k < nseg; ++k) entry_streak[k] = dirty_entry_[k] ? (uint8_t)std::min<uint32_t>(entry_streak[k] + 1u, 255u) : (uint8_t)0;
    if (const char* first = getenv("BROTLI_MI355X_DEBUG_MAP")) {  // value = first segment of the window shown (default 16)
      const uint32_t a = std::max<uint32_t>((uint32_t)atoi(first), 0u) ? (uint32_t)atoi(first) : 16u;
      std::string m;
      for (uint32_t k = a; k < a + 32 && k < nseg; ++k) m += dirty_entry_[k] ? (pending[k] ? 'P' : 'E') : (dirty[k] ? 'c' : '.');
      fprintf(stderr, "  map[%u..%u) %s\n", a, a + 32, m.c_str());
      for (uint32_t k = a; k < a + 12 && k < nseg; ++k)
        fprintf(stderr, "    seg %u [%u,%u) used pos %u apply %u head %u/%u cache %d %d %d %d | chained pos %u apply %u head %u/%u cache %d %d %d %d | exit pos %u cmds %u searches %u\n",
                k, segments_[k].start, segments_[k].end, entries_[k].pos, entries_[k].apply, entries_[k].head_kind, entries_[k].head_base,
                entries_[k].cache[0], entries_[k].cache[1], entries_[k].cache[2], entries_[k].cache[3], next_entries_[k].pos,
                next_entries_[k].apply, next_entries_[k].head_kind, next_entries_[k].head_base, next_entries_[k].cache[0],
                next_entries_[k].cache[1], next_entries_[k].cache[2], next_entries_[k].cache[3], exits_[k].pos, exits_[k].n_cmds,
                exits_[k].n_searches);
    }
    if (const char* path = getenv("BROTLI_MI355X_DEBUG_DIRTY")) {  // one line per round: why each segment is dirty
      if (FILE* f = fopen(path, "a")) {
        std::string m(nseg, '.');
        for (uint32_t k = 0; k < nseg; ++k) {
          const bool c = !cand_dirty.empty() && cand_dirty[k];
          m[k] = dirty_entry_[k] ? (c ? 'B' : (pending[k] ? 'P' : 'E')) : (c ? 'c' : (dirty[k] ? 'p' : '.'));
        }
        // (second half of the line: bit 1 = the distance cache at the entry differed, bit 0 = anything else)
        std::string why(nseg, '0');
        for (uint32_t k = 0; k < nseg; ++k) why[k] = (char)('0' + (entry_reason_[k] & 3));
        fprintf(f, "%s %s\n", m.c_str(), why.c_str());
        fclose(f);
      }
    }
    was_dirty = dirty;
    const bool aggressive = false;
    count = 0;
    // Cache front: text that repeats an earlier passage keeps that passage's distance alive in the cache through the
    // last-distance codes, for hundreds of commands and across any number of segments.  When the cache guessed at the head
    // of such a stretch was wrong (another occurrence of the passage), every segment hands the stale distance to the next:
    // segment k is re-parsed with the right cache in round r, k + 1 learns of it in round r + 1, and so on -- one segment
    // per round, while the segments ahead are re-parsed round after round with the stale cache because their candidates
    // keep changing (synth.mixed, 16 MiB at quality 5: 50 rounds).  Where the front is seen moving -- k differs in the cache
    // alone now, k - 1 did one round ago -- the segments behind k that are only owed a re-parse are left to k's chain, which
    // walks on with the real cache for as long as it arrives with another state than they were parsed with (br_parse_chain).
    bool front_run = false;
    cache_wave_next.assign(nseg, 0);
    for (uint32_t k = 0; k < nseg; ++k) {
      sched[k] = 0;
      if (segments_[k].flags & kSegFirstInBlock) front_run = false;
      if (!dirty[k]) {
        front_run = false;
        continue;
      }
      const bool must_redo = pending[k] || (!cand_dirty.empty() && cand_dirty[k]);  // its candidates changed
      if (front_run && !dirty_entry_[k]) {
        sched[k] = 2;
        pending[k] = 1;
        dirty[k] = 0;
        continue;
      }
      front_run = false;
      // (an entry predicted through a literal spree is as good as it gets: such a segment always gets its own chain)
      // (nor is a segment without copies whose entry differs in the distance cache alone left to its predecessor: a
      // chain that arrives there and sees the old parse come out again stops walking -- "passes the cache along",
      // br_parse_chain -- on the understanding that the cache composed here is verified for all such segments side by
      // side.  Deferring them as well made a changed cache creep through incompressible data two segments per round at
      // quality 9, where no lz77_check_cache answers for them: 74 rounds for 4 MiB of random bytes.)
      const bool passes_cache_along = entry_reason_[k] == 2 && exits_[k].n_cmds == 0 && exits_[k].ext_len == 0 &&
                                      getenv("BROTLI_MI355X_DEFER_CACHE_ONLY") == nullptr;
      const bool defer = (aggressive || entry_streak[k] >= 2) && !(segments_[k].flags & kSegFirstInBlock) && k > 0 && was_dirty[k - 1] &&
                         !predicted_entry_[k] && !passes_cache_along;
      if (defer) {
        sched[k] = must_redo ? 2 : 0;
        pending[k] = must_redo;  // stays owed until some chain really gets here
        dirty[k] = 0;
        continue;
      }
      // (kSchedOwnRows: the entry is the one it was parsed with last and only its candidate rows changed -- its chain may
      // restart from a checkpoint, lz77_chain.h Reparse)
      sched[k] = (dirty_entry_[k] || pending[k]) ? kSchedOwn : kSchedOwnRows;
      pending[k] = 0;
      list[count++] = k;
      entries_[k] = next_entries_[k];
      TouchSegment(k);
      const bool cache_only = dirty_entry_[k] && entry_reason_[k] == 2;
      cache_wave_next[k] = cache_only;
      front_run = follow_cache_fronts && cache_only && k > 0 && !(segments_[k].flags & kSegFirstInBlock) && cache_wave[k - 1];
    }
    cache_wave.swap(cache_wave_next);
    if (regime_flip) {
      last_death_seg = dict_death_seg_;
      static const bool no_regime_warmup = getenv("BROTLI_MI355X_NO_REGIME_WARMUP") != nullptr;
      if (warmup_bytes_ > 0 && dict_death_seg_ + 2 < nseg && !no_regime_warmup) {
        Warmup(dict_death_seg_ + 1, true, which, rbuf, &dirty);
        // (the dry run rewrote the entries of the scheduled segments behind that point: none of them is parsed "with the
        // entry it had last time" any more -- a chain that restarted from a checkpoint on that assumption kept the head of a
        // parse made from another entry; found by the emulation build's shadow parse on 256 MiB of text)
        for (uint32_t k = dict_death_seg_ + 1; k < nseg; ++k)
          if (sched[k] == kSchedOwnRows) sched[k] = kSchedOwn;
      }
    }
    if (getenv("BROTLI_MI355X_DEBUG")) fprintf(stderr, "mismatch pos %u apply %u cache %u apply-only %u | block-first: cache %u ext %u | head %u; dict death seg %u (forecast %u) flips %u\n", dbg_counts_[0], dbg_counts_[1], dbg_counts_[2], dbg_counts_[3], dbg_first_[0], dbg_first_[1], dbg_first_[2], dict_death_seg_, predicted_death_, dict_flips_);
    if (getenv("BROTLI_MI355X_DEBUG")) fprintf(stderr, "  resolver: %u of %u blocks taken over from the last pass\n", resolve_blocks_skipped_, (unsigned)block_segment_bytes_.size());
    if (getenv("BROTLI_MI355X_DEBUG")) fprintf(stderr, "round %u: flag changes %llu, dirty segments %u of %u (entry %u, candidates %u), predicted literal runs %u, cache rechecks passed %u\n", round, (unsigned long long)n_changed, count, nseg, n_dirty_entry, n_dirty_valid, predicted_runs_, rechecked);
    host_schedule_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t1).count();
    stamp("scheduled");
    if (count == 0) {
      if (const char* path = getenv("BROTLI_MI355X_DEBUG_FRONT")) {
        if (FILE* f = fopen(path, "a")) {
          fprintf(f, "FINAL");
          for (uint32_t k = 0; k < nseg; ++k) fprintf(f, " %u:%u", exits_[k].pos, exits_[k].n_cmds);
          fprintf(f, "\n");
          fclose(f);
        }
      }
      done = true;
      break;
    }
    full_round = false;
  }
  resolve_incremental_ = false;
  dev_free(dirty_dev);
  dev_free(list_dev);
  dev_free(geo_tables);
  dev_free(up_entries_dev);
  dev_free(got_exits_dev);
  dev_free(cont_count_dev);
  dev_free(cont_index_dev);
  dev_free(cont_exits_dev);
  dev_free(cont_entries_dev);
  dev_free(U.stale);
  dev_free(U.sched);
  dev_free(U.entry_dirty);
  dev_free(U.touched);
  dev_free(U.new_entries);
  dev_free(U.list);
  dev_free(U.counters);
  if (restart) {
    // what the pass so far says about the state at every block start
    saved_block_guess_.clear();
    for (uint32_t k = 0; k < nseg; ++k)
      if (segments_[k].flags & kSegFirstInBlock) saved_block_guess_[segments_[k].blk_start] = next_entries_[k];
    // Hardly any guess held: this input does not re-synchronise (incompressible stretches, data whose parse hangs on
    // the distance cache).  Parse it one chain per input block instead -- nothing is guessed inside a block then.
    Resegment(segment_bytes_);
    stats_.coarse_restarts++;
    RunRounds(false);
    return;
  }
  if (!done) throw std::runtime_error("brotli_mi355x: backward-reference search did not reach a fixed point");
  final_flags_ = which;
  if (selftest) {
    // the incrementally maintained rank structures / rows must equal a rebuild from the final flags
    if (use_rows_) SelfTestRows(which); else SelfTestRank(which, rbuf);
  }
  tm.stop(&stats_.ms_resolve);
  if (getenv("BROTLI_MI355X_DEBUG_EXITS"))
    for (uint32_t k = 0; k < nseg && k < 40; ++k)
      fprintf(stderr, "  seg %u [%u,%u) entry pos %u | exit pos %u insert %u cmds %u lits %u searches %u\n", k, segments_[k].start, segments_[k].end,
              entries_[k].pos, exits_[k].pos, exits_[k].insert_len, exits_[k].n_cmds, exits_[k].n_lits, exits_[k].n_searches);
  for (uint32_t k = 0; k < nseg; ++k) stats_.searches += exits_[k].n_searches;
}


void Lz77Stage::DumpFlags(uint8_t* out, size_t size) const {
  dev_sync();
  dev_d2h(out, B_.flags[final_flags_], std::min<size_t>(size, P_.total_bytes));
}

// Bring-up checks of the sort / rank kernels against a host recomputation (BROTLI_MI355X_SELFTEST=1).
void Lz77Stage::SelfTestSort() {
  const uint32_t n = P_.total_bytes;
  std::vector<uint16_t> keys(n), skeys(n);
  std::vector<uint32_t> by_key(n);
  dev_sync();
  dev_d2h(keys.data(), B_.keys, (size_t)n * 2);
  dev_d2h(skeys.data(), B_.sorted_keys, (size_t)n * 2);
  dev_d2h(by_key.data(), B_.by_key, (size_t)n * 4);
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t p = by_key[i];
    if (p >= n || seen[p]) throw std::runtime_error("selftest: by_key is not a permutation at slot " + std::to_string(i));
    seen[p] = 1;
    if (skeys[i] != keys[p]) throw std::runtime_error("selftest: sorted_keys mismatch at slot " + std::to_string(i));
    if (i > 0) {
      if (skeys[i - 1] > skeys[i]) throw std::runtime_error("selftest: keys not ascending at slot " + std::to_string(i));
      if (skeys[i - 1] == skeys[i] && by_key[i - 1] >= p)
        throw std::runtime_error("selftest: sort not stable at slot " + std::to_string(i));
    }
  }
}

void Lz77Stage::SelfTestRank(int which, int rbuf) {
  const uint32_t n = P_.total_bytes;
  std::vector<uint16_t> skeys(n);
  std::vector<uint32_t> by_key(n), sorted(n), info((size_t)n * 2);
  std::vector<uint8_t> flags(n);
  dev_sync();
  dev_d2h(skeys.data(), B_.sorted_keys, (size_t)n * 2);
  dev_d2h(by_key.data(), B_.by_key, (size_t)n * 4);
  dev_d2h(sorted.data(), B_.sorted[rbuf], (size_t)n * 4);
  dev_d2h(info.data(), B_.info[rbuf], (size_t)n * 8);
  dev_d2h(flags.data(), B_.flags[which], n);
  uint32_t first = 0, local = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t p = by_key[i];
    if (i == 0 || skeys[i - 1] != skeys[i]) {
      first = i;
      local = 0;
      if (key_first_[skeys[i]] != i) throw std::runtime_error("selftest: key_first mismatch for key " + std::to_string(skeys[i]));
    }
    if (i + 1 == n || skeys[i + 1] != skeys[i]) {
      if (key_last_[skeys[i]] != i + 1) throw std::runtime_error("selftest: key_last mismatch for key " + std::to_string(skeys[i]));
    }
    // second word: what the reference's 16-bit per-bucket counter lets a search see (ring_count, lz77_kernels.hip); with
    // counters carried in from the stream in front or a hasher reset inside the text only the slot is checked here
    const bool plain_counters = B_.count_base == nullptr && P_.reset_pos == 0;
    const uint32_t visible = std::min(local & 0xffffu, local);
    if (info[2 * (size_t)p] != first + local || (plain_counters && info[2 * (size_t)p + 1] != visible))
      throw std::runtime_error("selftest: info mismatch at slot " + std::to_string(i));
    if (flags[p] & 1) {
      if (sorted[first + local] != p) throw std::runtime_error("selftest: sorted mismatch at slot " + std::to_string(i));
      local++;
    }
  }
}

// Candidate rows against a host recomputation from the flags (BROTLI_MI355X_SELFTEST=1): after the initial build and
// after the last round, i.e. after every incremental update in between.
void Lz77Stage::SelfTestRows(int which) {
  const uint32_t n = P_.total_bytes;
  std::vector<uint16_t> skeys(n);
  std::vector<uint32_t> by_key(n), rows((size_t)n * kRowEntries);
  std::vector<uint8_t> flags(n), fbits(n), text((size_t)n + 64, 0);
  dev_sync();
  dev_d2h(skeys.data(), B_.sorted_keys, (size_t)n * 2);
  dev_d2h(by_key.data(), B_.by_key, (size_t)n * 4);
  dev_d2h(rows.data(), B_.rows, (size_t)n * kRowEntries * 4);
  dev_d2h(flags.data(), B_.flags[which], n);
  dev_d2h(fbits.data(), B_.fbits, n);
  dev_d2h(text.data(), B_.text, n);
  auto tag = [&](uint32_t p) {
    uint32_t v;
    memcpy(&v, text.data() + p, 4);
    return (uint32_t)((v * 0x9E3779B1u) >> 16);
  };
  const uint32_t depth = 1u << P_.block_bits;
  uint32_t first = 0, stored_before = 0, local_before = 0, since_reset = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t p = by_key[i];
    if (i == 0 || skeys[i - 1] != skeys[i]) {
      first = i;
      stored_before = B_.count_base ? carry_->key_counts[skeys[i]] : 0u;
      local_before = 0;
      since_reset = 0;
    }
    const uint32_t counter = (P_.reset_pos != 0 && p >= P_.reset_pos) ? since_reset : stored_before;
    if ((fbits[i] & 1) != (flags[p] & 1)) throw std::runtime_error("selftest: fbits mismatch at slot " + std::to_string(i));
    if (fbits[i] & 2) throw std::runtime_error("selftest: leftover change mark at slot " + std::to_string(i));
    {
      const bool big = P_.reset_pos != 0 || B_.count_base != nullptr || key_last_[skeys[i]] - key_first_[skeys[i]] >= 65536u;
      const bool want_wrap = big && counter != 0 && (counter & 0xffffu) == 0;
      if (want_wrap != ((fbits[i] & 4) != 0)) throw std::runtime_error("selftest: wrap mark mismatch at slot " + std::to_string(i));
    }
    const uint32_t max_backward = std::min(p, P_.max_backward_limit);
    const uint32_t d = std::min(depth, counter & 0xffffu);
    const uint32_t oldest = p >= P_.reset_pos ? P_.reset_vis : 0u;
    std::vector<uint32_t> want;
    uint32_t seen = 0;
    for (uint32_t j = i; j > first && seen < d;) {
      --j;
      if (!(flags[by_key[j]] & 1)) continue;
      if (p - by_key[j] > max_backward || by_key[j] < oldest) break;
      ++seen;
      if (tag(by_key[j]) == tag(p)) want.push_back(by_key[j]);
    }
    for (size_t c = 0; c <= want.size() && c < kRowEntries; ++c) {
      const uint32_t have = rows[(size_t)p * kRowEntries + c];
      const uint32_t expect = c < want.size() ? want[c] : kRowEnd;
      if (have != expect)
        throw std::runtime_error("selftest: row of position " + std::to_string(p) + " entry " + std::to_string(c) + ": " + std::to_string(have) +
                                 " instead of " + std::to_string(expect));
    }
    stored_before += flags[p] & 1;
    local_before += flags[p] & 1;
    if (P_.reset_pos != 0 && p >= P_.reset_vis) since_reset += flags[p] & 1;
  }
  (void)local_before;
}

void Lz77Stage::KeyCountsBetween(uint32_t from, uint32_t upto, std::vector<uint32_t>* out) {
  uint32_t* dev = (uint32_t*)dev_alloc(2 * 65536 * 4);
  lz77_key_counts(P_, B_, final_flags_, upto, dev, false);
  lz77_key_counts(P_, B_, final_flags_, from, dev + 65536, false);
  std::vector<uint32_t> both(2 * 65536);
  dev_d2h(both.data(), dev, both.size() * 4);
  dev_free(dev);
  out->resize(65536);
  for (uint32_t k = 0; k < 65536; ++k) (*out)[k] = both[k] - both[65536 + k];
}

void Lz77Stage::KeyCountsBefore(uint32_t upto, std::vector<uint32_t>* out) {
  uint32_t* dev = (uint32_t*)dev_alloc(65536 * 4);
  lz77_key_counts(P_, B_, final_flags_, upto, dev);
  out->resize(65536);
  dev_d2h(out->data(), dev, 65536 * 4);
  dev_free(dev);
}

void Lz77Stage::Gather() {
  const uint32_t nseg = (uint32_t)segments_.size();
  for (uint32_t k = 0; k < nseg; ++k)
    if (exits_[k].bad_commands != 0)
      throw std::runtime_error(
          "brotli_mi355x: the reference encoder fails on this input: a match cut to one byte at the end of the custom dictionary "
          "gives a copy it cannot encode (fix_unbroken_len, backward_references/mod.rs:42-54; GetCopyLengthCode, command.rs:91-93)");
  std::vector<uint32_t> offsets(nseg), counts(nseg);
  // Two batches: a command can receive both a carried-in insert length (kind 2) and an extension of its copy (kind 0);
  // the patches of one batch touch distinct commands, so each batch is applied with one thread per patch.
  std::vector<CmdPatch> fix, fix_ext;
  size_t t = 0, ti = 0;
  uint64_t total = 0;
  for (uint32_t k = 0; k < nseg; ++k) {
    offsets[k] = (uint32_t)total;
    counts[k] = exits_[k].n_cmds;
    total += exits_[k].n_cmds;
    while (ti < trailing_.size() && trailing_[ti].after_segment == k) {
      CmdPatch p{};
      p.index = (uint32_t)total;
      p.kind = 1;
      p.value = trailing_[ti].insert_len;
      fix.push_back(p);
      total++;
      ti++;
    }
  }
  (void)t;
  for (const Carry& c : carries_) {
    CmdPatch p{};
    p.index = offsets[c.segment];
    p.kind = 2;
    p.value = c.literals;
    fix.push_back(p);
  }
  for (const Patch& pt : patches_) {
    CmdPatch p{};
    p.index = offsets[pt.segment] + pt.index;
    p.kind = 0;
    p.value = pt.ext;
    fix_ext.push_back(p);
  }
  total_cmds_ = total;
  stats_.total_commands = total;
  // meta-block command offsets
  for (MetaBlockPlan& mb : metablocks_) {
    const uint32_t first_seg = mb.cmd_offset;
    mb.cmd_offset = first_seg < nseg ? offsets[first_seg] : (uint32_t)total;
  }
  if (gathered_capacity_ < total + 16) {
    dev_free(gathered_cmds_);
    gathered_capacity_ = total + total / 8 + 1024;
    gathered_cmds_ = (Command*)dev_alloc(gathered_capacity_ * sizeof(Command));
  }
  dev_h2d(gather_offsets_dev_, offsets.data(), nseg * 4);
  dev_h2d(gather_counts_dev_, counts.data(), nseg * 4);
  lz77_gather_commands(P_, B_, nseg, gather_offsets_dev_, gather_counts_dev_, gathered_cmds_);
  for (const std::vector<CmdPatch>* batch : {&fix, &fix_ext}) {
    if (batch->empty()) continue;
    CmdPatch* fix_dev = (CmdPatch*)dev_alloc(batch->size() * sizeof(CmdPatch));
    dev_h2d(fix_dev, batch->data(), batch->size() * sizeof(CmdPatch));
    lz77_patch_commands(gathered_cmds_, fix_dev, (uint32_t)batch->size());
    dev_sync();
    dev_free(fix_dev);
  }
  dev_sync();
}

}  // namespace brotli_mi355x
