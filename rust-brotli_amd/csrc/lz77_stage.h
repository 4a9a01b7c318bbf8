// lz77_stage.h -- host orchestration of the backward-reference search on the device.
//
// Replaces, for one complete stream, the per-block calls the reference makes from encode_data():
//   InitOrStitchToPreviousBlock (encode.rs:2417), extend_last_command (:2435),
//   BrotliCreateBackwardReferences (:2438) and the meta-block flush rule (:2454-2483),
// and produces exactly the command list the reference would produce, by speculative parallel
// parsing + a host "resolver" that chains segment exits to entries until a fixed point is reached.
#ifndef BROTLI_MI355X_LZ77_STAGE_H_
#define BROTLI_MI355X_LZ77_STAGE_H_

#include <stdint.h>
#include <map>
#include <memory>
#include <utility>
#include <vector>

#include "device_api.h"
#include "encoder_params.h"
#include "quick_api.h"

namespace brotli_mi355x {

struct MetaBlockPlan {
  uint32_t start;       // last_flush_pos (position in text)
  uint32_t end;         // input_pos at the flush
  uint32_t cmd_offset;  // into the gathered command array
  uint32_t n_cmds;      // including the trailing insert-only command, if any
  uint32_t n_literals;
  bool uncompressed;    // should_compress() == false (encode.rs:1325-1354) or forced by the size fallback
  bool is_last;
  int32_t dist_cache_after[4];
  int32_t saved_dist_cache[4];  // dist cache at the start of the meta-block (for IR-free storing)
  // static-dictionary throttle state behind the meta-block (where a later batch of the stream resumes)
  uint32_t dict_lookups_after, dict_matches_after;
  bool dict_dead_after;
};

struct Lz77Stats {
  uint32_t rounds = 0;
  uint64_t segments_parsed = 0;
  uint64_t searches = 0;
  uint64_t total_commands = 0;
  uint32_t incremental_ranks = 0, full_ranks = 0, coarse_restarts = 0;
  uint32_t burst_launches = 0;  // list launches scheduled on the device (device_api.h, bursts)
  // wall-clock milliseconds per phase (device synchronised), filled when profiling is enabled
  uint64_t cache_rechecks = 0;
  double ms_keys = 0, ms_sort = 0, ms_init = 0, ms_warmup = 0, ms_rank = 0, ms_parse = 0, ms_resolve = 0, ms_gather = 0, ms_total = 0;
};

// Qualities 10 / 11: the H10 trees as a piece of a stream leaves them for the next one (device memory, owned here)
struct ZopfliCarry {
  uint32_t* buckets = nullptr;  // [1 << 17]
  uint32_t* forest = nullptr;   // [2 << lgwin]
  uint32_t lgwin = 0;
  uint64_t text_base = 0;       // stream position of text position 0 of the piece that left them
  ZopfliCarry() = default;
  ZopfliCarry(const ZopfliCarry&) = delete;
  ZopfliCarry& operator=(const ZopfliCarry&) = delete;
  ~ZopfliCarry();
};

// Qualities 2 .. 4: the BasicHasher table (slots + dictionary-throttle counters, quick_api.h) as a piece of a stream leaves it
struct QuickCarry {
  uint32_t* table = nullptr;  // [quick_table_words()]
  uint64_t text_base = 0;     // stream position of text position 0 of the piece that left it
  QuickCarry() = default;
  QuickCarry(const QuickCarry&) = delete;
  QuickCarry& operator=(const QuickCarry&) = delete;
  ~QuickCarry();
};

// What an encoder keeps between two encode_data calls of one stream (BROTLI_OPERATION_FLUSH): the reference's hasher
// contents, distance cache and dictionary-throttle counters.  The hasher is represented by which positions of the
// stream so far are stored in it.
struct StreamCarry {
  bool valid = false;
  int32_t dist_cache[4] = {4, 11, 15, 16};
  uint32_t dict_lookups = 0, dict_matches = 0;
  bool dict_dead = false;
  std::vector<uint8_t> stored;  // per byte of the kept part of the stream: bit 0 = the position is in the hash table
  HasherParams hasher;          // fixed by the first encode_data (encode.rs:1125-1161)
  size_t size_hint = 0;
  // ---- bounded-memory streaming: only a window of the stream so far is kept as the prefix of the next piece
  uint64_t stream_base = 0;     // stream position of the first kept byte; a multiple of the ring-buffer size, so that
                                // ring-buffer indices (position & ring_mask) of the kept bytes are what they were
  std::vector<uint32_t> key_counts;  // per hash key: positions stored in the hash table in front of stream_base (the
                                     // reference's u16 ring counter num[key] counts from the start of the stream)
  uint32_t tail_bits = 0, tail_nbits = 0;  // last, incomplete byte of the output so far (a piece that ends on a
                                           // meta-block boundary without a flush is not byte aligned)
  // ---- what the first piece of a stream with a custom dictionary / a catable stream fixes for all that follow
  uint32_t dict_break = 0;         // ring_buffer_break: the reference keeps cutting matches at this RING index for the whole
                                   // stream, also after the dictionary has been overwritten (mod.rs:42-54 on masked indices)
  bool use_dictionary = true;      // static dictionary still in use (a custom dictionary / catable turn it off for good)
  uint32_t prev_floor = 0;         // text position below which the context bytes of the next meta-block read as 0: the end of a
                                   // custom dictionary until the first meta-block has been written (encode.rs:2526-2534 runs only then)
  uint32_t catable_raw_bytes = 0;  // is_first_mb: 0 nothing, 1 one, 2 both raw first bytes of a catable stream are out (encode.rs:2283-2333)
  std::shared_ptr<ZopfliCarry> zopfli;  // qualities 10 / 11: the trees at the resume point (null: nothing searched yet)
  std::shared_ptr<QuickCarry> quick;    // qualities 2 .. 4: the hash table at the resume point (null: nothing searched yet)
  bool magic_owed = false;  // the magic-number block (BROTLI_PARAM_MAGIC_NUMBER) has not been written yet: so far only metadata
                            // blocks asked for before any input went out, and those do not pass through encode_data
};

class Lz77Stage {
 public:
  Lz77Stage() = default;
  ~Lz77Stage();
  Lz77Stage(const Lz77Stage&) = delete;
  Lz77Stage& operator=(const Lz77Stage&) = delete;

  // params must be finalized (FinalizeParams + ChooseHasher).  text_dev holds prefix_bytes of
  // custom dictionary followed by input_bytes of input, plus >= 64 readable zero bytes.
  // raw_head_bytes: bytes at the start of the input that the caller stores uncompressed before the
  // first block is searched (catable streams: 2, encode.rs:2283-2333).
  void Setup(const EncoderParams& params, uint8_t* text_dev, uint32_t prefix_bytes, uint32_t input_bytes,
             uint32_t raw_head_bytes, uint32_t segment_bytes);
  // forces meta-block `index` (and only it) to be stored uncompressed in the next Run (size fallback,
  // encode.rs:2141-2163)
  void ForceUncompressed(uint32_t index) { forced_uncompressed_.push_back(index); }
  // continuation of a stream after a flush: the prefix is the stream so far, `carry` its state; stream_is_last = false
  // keeps the ISLAST bit off the final meta-block (more input may follow)
  void SetStreamState(const StreamCarry* carry, bool stream_is_last, bool partial = false, bool last_block_processed_early = false) {
    carry_ = carry;
    stream_is_last_ = stream_is_last;
    partial_ = partial;
    early_last_ = last_block_processed_early && stream_is_last;
  }
  // last_block_processed_early: the stream ends exactly at the end of an input block that the reference had already
  // run through encode_data (is_last = false) when the FINISH operation came without further input.  Whether the
  // meta-block got closed there was then up to the flush rule; if it was, the stream ends with an EMPTY last meta-block
  // (needs_empty_last()), if not, the open meta-block becomes the last one (encode.rs:2454-2483, 2940-2975).
  bool needs_empty_last() const { return needs_empty_last_; }
  // partial = true: more input of the same stream follows without a flush in between.  Only the meta-blocks that the
  // flush rule (encode.rs:2454-2477) closes by itself are planned; resume_pos() is where the first open one starts --
  // the next piece starts there again.
  uint32_t resume_pos() const { return resume_pos_; }
  // text position (a block start inside the input, or 0) at which the reference's hasher is reset by its 32-bit position
  // wrap (Lz77Params::reset_pos); call after Setup()
  void SetHasherReset(uint32_t pos) {
    P_.reset_pos = pos;
    P_.reset_vis = pos >= 3 ? pos - 3 : 0;
  }
  // stored positions per key in [from, upto) without the carried-in counts
  void KeyCountsBetween(uint32_t from, uint32_t upto, std::vector<uint32_t>* out);
  // stored positions per hash key in front of text position `upto` (added to the carried-in counts): the key_counts of
  // the next piece when it keeps text[upto ..) as its prefix
  void KeyCountsBefore(uint32_t upto, std::vector<uint32_t>* out);
  // state after Run() for the next continuation (dist cache comes from metablocks().back())
  // qualities 10 / 11: hands the H10 trees over to the carry of the next piece -- as they are at the end of the text, or
  // (partial piece) as they were in front of the block that starts the meta-block still open
  void ExportZopfli(StreamCarry* co, bool partial);
  bool is_zopfli() const { return use_zopfli_; }
  // qualities 2 .. 4: the same for the BasicHasher table
  void ExportQuick(StreamCarry* co, bool partial);
  bool is_quick() const { return use_quick_; }
  void FinalDictState(uint32_t* lookups, uint32_t* matches, bool* dead) const {
    *lookups = final_dict_lookups_;
    *matches = final_dict_matches_;
    *dead = final_dict_dead_;
  }
  uint32_t total_bytes() const { return P_.total_bytes; }
  void set_warmup_bytes(uint32_t n) { warmup_bytes_ = n; }
  void Run();
  void DumpFlags(uint8_t* out, size_t size) const;  // test hook: final per-position flags

  const std::vector<MetaBlockPlan>& metablocks() const { return metablocks_; }
  Command* commands_dev() const { return gathered_cmds_; }
  uint32_t num_commands() const { return (uint32_t)total_cmds_; }
  const Lz77Stats& stats() const { return stats_; }
  const Lz77Params& device_params() const { return P_; }
  const uint8_t* text_dev() const { return B_.text; }

 private:
  void BuildSegments();
  void Resegment(uint32_t segment_bytes);
  void RunRounds(bool allow_restart);
  void RunLive();
  void RunZopfli();
  void RunQuick();
  void RunQuickSpec();
  void InitEntries();
  void InitFlags();
  bool Resolve(bool final_pass);
  void Gather();
  struct WarmupJob {  // dry runs in flight between WarmupBegin and WarmupEnd
    std::vector<uint32_t> ks;
    PinnedArray<Segment> wsegs;
    PinnedArray<SegEntry> wentries;
    PinnedArray<SegExit> wexits;
    Segment* wsegs_dev = nullptr;
    SegEntry* wentries_dev = nullptr;
    SegExit* wexits_dev = nullptr;
    uint32_t count = 0;
    bool dict_dead = false, whole_input = true;
    int mark = 0;
  };
  void Warmup(uint32_t first_seg, bool dict_dead, int which, int rbuf, const std::vector<uint8_t>* only_after_dirty);
  void WarmupBegin(WarmupJob* job, uint32_t first_seg, uint32_t end_seg, bool dict_dead, int which, int rbuf,
                   const std::vector<uint8_t>* only_after_dirty, int mark);
  void WarmupEnd(WarmupJob* job);
  void SelfTestSort();
  void SelfTestRank(int which, int rbuf);
  void SelfTestRows(int which);
  bool FetchShouldCompress();
  bool ResolvePass(bool final_pass, bool incremental);
  void Release();

  EncoderParams params_;
  Lz77Params P_{};
  Lz77Buffers B_{};
  LiveBuffers L_{};    // live chains (lz77_live.h)
  bool use_live_ = false;
  bool use_zopfli_ = false;  // qualities 10 / 11 (zopfli_device.h)
  ZopfliJob Z_{};
  uint32_t* zsnap_buckets_ = nullptr;  // partial pieces: the trees in front of the block that starts the open meta-block
  uint32_t* zsnap_forest_ = nullptr;
  bool use_quick_ = false;   // qualities 2 .. 4 (quick_device.h)
  QuickJob Q_{};
  uint32_t* qsnap_table_ = nullptr;    // partial pieces: the table in front of the block that starts the open meta-block
  bool use_qspec_ = false;   // ... on the speculative path (quick_spec.h): whole streams in one piece
  QuickSpec S_{};
  EncoderParams qspec_params_{};  // (what Setup was called with, for the fall-back to the serial path)
  uint8_t* qspec_text_ = nullptr;
  uint32_t qspec_prefix_ = 0, qspec_input_ = 0, qspec_raw_head_ = 0;
  bool qspec_coarse_ = false;    // the input was re-cut into one chain per block (it did not fall into step inside blocks)
  bool qspec_books_in_ = false;  // a later piece: the books of the throttle as the carried table holds them (qspec_lookups_ / _matches_)
  uint32_t qspec_lookups_ = 0, qspec_matches_ = 0;
  bool live_verify_ = false;
  std::vector<LiveBlockState> live_state_;  // Resolve(): the meta-block books at the entry of every block (live chains)
  uint32_t input_bytes_ = 0;
  uint32_t raw_head_bytes_ = 0;
  uint32_t segment_bytes_ = 4096;
  uint32_t warmup_bytes_ = 384;
  uint32_t block_bytes_ = 65536;
  std::vector<Segment> segments_;
  std::vector<uint32_t> block_first_segment_, block_segment_bytes_;  // per input block (+ one-past-the-end entry)
  std::vector<uint8_t> coarse_blocks_;  // blocks that are parsed by a single chain
  uint32_t total_cmd_slots_ = 0;
  std::vector<double> warm_lookups_, warm_matches_;  // per segment, forecast from the warm-up dry run
  uint32_t predicted_death_ = 0xffffffffu;
  int final_flags_ = 0;
  const StreamCarry* carry_ = nullptr;
  bool stream_is_last_ = true;
  bool partial_ = false;
  bool early_last_ = false, needs_empty_last_ = false;
  uint32_t resume_pos_ = 0;
  uint32_t final_dict_lookups_ = 0, final_dict_matches_ = 0;
  bool final_dict_dead_ = false;
  std::map<uint32_t, SegEntry> saved_block_guess_;  // by block start
  size_t cmds_bytes_ = 0;
  PinnedArray<uint32_t> key_first_, key_last_;  // host copy of the slot range of every key
  PinnedArray<Segment> segments_upload_;        // page-locked staging of segments_ on its way to the device
  PinnedArray<SegEntry> entries_;   // entries used by the most recent parse (page-locked: uploaded whole in round 0)
  std::vector<SegEntry> next_entries_;
  PinnedArray<SegExit> exits_;  // (written by the device every round)
  struct RoundBuffers {  // what list rounds move between host and device (RunRounds)
    PinnedArray<uint32_t> up_index, cont_index, counts;
    PinnedArray<SegEntry> up_entries, cont_entries;
    PinnedArray<SegExit> got_exits, cont_exits;
  } round_buffers_;
  std::vector<MetaBlockPlan> metablocks_;
  std::vector<uint32_t> forced_uncompressed_;
  struct Patch {
    uint32_t segment;  // segment holding the command to patch
    uint32_t index;    // index inside that segment's slab
    uint32_t ext;      // copy length to add (extend_last_command)
  };
  std::vector<Patch> patches_;
  struct TrailingInsert {
    uint32_t after_segment;  // inserted after the commands of this segment
    uint32_t insert_len;
  };
  std::vector<TrailingInsert> trailing_;
  struct Carry {
    uint32_t segment;   // first command of this segment ...
    uint32_t literals;  // ... gets this many extra literals in front
  };
  std::vector<Carry> carries_;
  Command* gathered_cmds_ = nullptr;
  size_t gathered_capacity_ = 0;
  size_t total_cmds_ = 0;
  uint32_t* histo_dev_ = nullptr;
  uint32_t* gather_offsets_dev_ = nullptr;
  uint32_t* gather_counts_dev_ = nullptr;
  Lz77Stats stats_;
  uint32_t dbg_counts_[4] = {0, 0, 0, 0};
  uint32_t dbg_first_[4] = {0, 0, 0, 0};
  uint32_t* dbg_mismatch_ = nullptr;  // set to dbg_counts_ under BROTLI_MI355X_DEBUG
  std::map<std::pair<uint32_t, uint32_t>, bool> should_compress_cache_;
  std::vector<uint8_t> wanted_guesses_;  // the provisional answers the pass went on with
  bool should_compress_guess_ = true;
  std::vector<std::pair<uint32_t, uint32_t>> wanted_histograms_;  // {start, bytes} of the meta-blocks ResolvePass() has no should_compress answer for
  uint32_t first_dirty_ = 0;
  std::vector<uint8_t> predicted_entry_;  // the entry chained for segment k comes out of a predicted literal run
  std::vector<uint8_t> entry_reason_;  // why dirty_entry_[k] is set, see Resolve()
  bool substitute_inherited_pushes_ = false;  // Resolve(): pushed distances that came out of the entry cache follow it
  uint32_t RecheckCacheOnly(int which, std::vector<uint32_t>* accepted = nullptr);
  double host_resolve_ms_ = 0, host_schedule_ms_ = 0;  // BROTLI_MI355X_PROFILE
  uint32_t predicted_runs_ = 0;  // segments whose exit the last Resolve() predicted (literal spree arithmetic)
  // ---- Resolve() block by block: a block whose segments nobody has touched since the last pass (entries_ / exits_) and that
  // is entered in the state it was entered in then comes out as it did then -- what it appended to the output lists is copied,
  // what it marked stays marked, and the pass goes on in the state it was left in.  Late rounds of a large input parse a few
  // dozen of a quarter of a million segments; the pass over all of them was 3 ms a round at 1 GiB.
  struct ResolveFlow {  // everything a block's processing depends on besides its own segments
    int32_t cache[4], saved_cache[4];
    uint32_t last_insert_len, last_flush_pos, mb_first_seg, resume_pos;
    uint64_t num_commands, num_literals, mb_cmds;
    uint32_t dict_state, dict_L, dict_M, dict_left_alive_at, dict_flips;
    int64_t dict_slack;
    uint32_t last_valid, last_seg, last_idx, last_dist_code, last_copy_len;
    uint32_t compress_guess, needs_empty_last;
  };
  struct ResolveBlockRecord {  // what the last pass did in one block
    ResolveFlow in;                                          // the state it was entered in
    uint32_t n_metablocks, n_patches, n_trailing, n_carries;  // the lengths of the output lists at that moment
    uint32_t marks, first_dirty, predicted_runs;              // dirty marks set inside the block (first_dirty: nseg = none)
    uint32_t dbg_counts[4], dbg_first[4];
    uint32_t merged_ext;                                      // added to the last patch of the blocks in front (extend_last_command)
  };
  std::vector<ResolveBlockRecord> block_records_;  // [blocks + 1]: the last entry holds the state and list lengths at the end
  std::vector<MetaBlockPlan> metablocks_prev_;
  std::vector<Patch> patches_prev_;
  std::vector<TrailingInsert> trailing_prev_;
  std::vector<Carry> carries_prev_;
  std::vector<uint8_t> block_touched_;  // [blocks]
  bool block_records_valid_ = false, touch_all_ = true, resolve_incremental_ = false;
  uint32_t resolve_blocks_skipped_ = 0;
  void TouchSegment(uint32_t k) {
    if (!block_touched_.empty()) block_touched_[segments_[k].block_index] = 1;
  }
  std::vector<uint8_t> dirty_entry_;
  uint32_t dict_death_seg_ = 0xffffffffu;
  uint32_t dict_flips_ = 0;
  bool owns_buffers_ = false;
  uint32_t* count_base_dev_ = nullptr;  // carried-in ring counters per key (StreamCarry::key_counts)
  bool use_rows_ = false;      // quality 5: candidate rows instead of rank structures (device_api.h)
  bool has_big_keys_ = false;  // some hash key owns >= 65 536 positions (the u16 ring counter of the reference wraps)
};

}  // namespace brotli_mi355x
#endif
