// device_api.h -- the narrow seam between the C++ host driver and the gfx950 kernels.
//
// The host driver (encoder_host.cpp) only talks to the device through these functions.  The product
// library implements them with HIP kernels (lz77_kernels.hip, metablock_kernels.hip).  Tests link the
// same host driver against a serial CPU emulation of this seam (tests/emu/) so that the speculative
// parse / resolver logic can be exercised without a GPU; that emulation is never part of the product.
#ifndef BROTLI_MI355X_DEVICE_API_H_
#define BROTLI_MI355X_DEVICE_API_H_

#include <stddef.h>
#include <stdint.h>

#include "lz77_types.h"

// Every thread of the host program works on its own HIP stream (hipStreamPerThread): independent encoder calls from
// different threads overlap on the GPU, nothing ever runs on the synchronising null stream.
// BR_STREAM is what every launch, copy, fill and event of the library names as its stream.  Small zero fills (dev_alloc of small
// blocks, dev_memset(p, 0, n)) are not issued one by one -- a 152 KB call queued about a hundred of them, 5 us of dispatch each --
// but noted and written by ONE launch in front of the next operation on the stream: naming the stream is what flushes them, so they
// keep their place in the order of the stream.
#define BR_STREAM (::brotli_mi355x::dev_stream_flushed())

struct ihipStream_t;
namespace brotli_mi355x {

ihipStream_t* dev_stream_flushed();  // the calling thread's stream, behind the zero fills noted so far (device_runtime.hip)

// ---- memory ----
void* dev_alloc(size_t bytes);  // zero-initialised device allocation; throws std::runtime_error
void* dev_alloc_uninit(size_t bytes);  // contents undefined: only for arrays that are written before they are read
void dev_free(void* p);
void dev_memset(void* p, int value, size_t bytes);
void dev_h2d(void* dst, const void* src, size_t bytes);
void dev_d2h(void* dst, const void* src, size_t bytes);        // copy + wait
// bulk transfers between the caller's (ordinary, pageable) memory and the device: chunks are bounced through two
// page-locked buffers so that the host copy of one overlaps the bus transfer of the other (the runtime's own path for
// pageable memory moves 64 MiB in ~28 ms); memory that is already page-locked is copied directly.  dev_d2h_bulk
// returns when the data has arrived.
void dev_h2d_bulk(void* dst, const void* src, size_t bytes);
void dev_d2h_bulk(void* dst, const void* src, size_t bytes);
void dev_d2h_async(void* dst, const void* src, size_t bytes);  // several copies, then one dev_sync()
void dev_d2d(void* dst, const void* src, size_t bytes);
void dev_sync();
void dev_mark();       // records a point in the calling thread's stream ...
void dev_wait_mark();  // ... and waits until everything queued before the last mark has finished (later work keeps running)
// the same with a number (0..3): several points of the stream waited for one after the other
void dev_mark_n(int i);
void dev_wait_mark_n(int i);
// page-locked host memory for the buffers that go back and forth every round (copies to and from pageable memory are
// staged by the runtime and block the caller)
void* dev_host_alloc(size_t bytes);
void dev_host_free(void* p);
void dev_pool_counters(double* out4, bool reset);  // hipMalloc calls / ms, hipFree calls / ms of the calling thread
// before work is handed to helper threads that allocate from pools of their own: if less than `bytes` are free and the calling thread
// keeps more than twice `own_share` pooled (left-overs of earlier, larger calls), its idle blocks go back to the driver
void dev_make_room_for(size_t bytes, size_t own_share);
void dev_make_room(unsigned min_free_share);  // trims every thread's pool if less than that percentage of the device memory is free
size_t dev_trim_pool();  // returns the pooled, currently unused device memory of the calling thread to the driver; bytes freed
int dev_current_device();        // the calling thread's device (helper threads adopt their caller's)
void dev_use_device(int device);
int dev_device_count();          // visible HIP devices
const char* dev_name();  // "hip:gfx950 ..." or "host-emulation"

// Static read-only tables resident on the device (dictionary, dictionary hash, log tables ...).
struct DeviceTables {
  const uint16_t* dict_hash;
  const uint8_t* dict_data;
  const uint32_t* dict_offsets_by_length;
  const uint8_t* dict_size_bits_by_length;
  const float* logs_16;
  const float* logs_8;
  const uint8_t* utf8_context_lookup;   // 512
  const uint8_t* signed_context_lookup; // 256
  const uint16_t* dict_lut_buckets;     // kStaticDictionaryBuckets [32768] (BrotliFindAllStaticDictionaryMatches, qualities 10 / 11)
  const uint32_t* dict_lut_words;       // kStaticDictionaryWords [31705]
};
const DeviceTables& dev_tables();

// ---- LZ77 stage ----
static constexpr uint32_t kChangedCap = 1u << 16;
struct RankInitialHint {
  uint32_t first_block_start, block_bytes;
  uint32_t prefix_is_dictionary;  // the prefix flags follow the custom-dictionary rule (all stored but the last htl - 1)
};
struct SegGeometry;

struct Lz77Buffers {
  uint8_t* text;        // total_bytes + 64
  uint16_t* keys;       // total_bytes
  uint32_t* by_key;     // positions sorted by (key, position)            [total_bytes]
  uint16_t* sorted_keys;// keys in that order                              [total_bytes]
  uint8_t* fbits;       // stored bit of by_key[i], scratch of the full rank pass   [total_bytes + 64]
  uint32_t* info[2];    // per position {rank, count of stored same-key positions before it} [2 * total_bytes], double buffered
  uint32_t* sorted[2];  // stored positions in (key,pos) order             [total_bytes], double buffered
  uint32_t* key_base;   // per key: stored positions in front of its first slot (scratch of the full rank pass) [65536 + 1]
  uint32_t* key_first;  // per key: first / one-past-last slot in (key,pos) order  [65536 + 1] each
  uint32_t* key_last;
  uint32_t* changed_keys;  // keys whose stored flags changed in the last parse launch(es) [kChangedCap]
  uint32_t* changed_count; // [1]; word 8: sampled count of run starts (lz77_compute_keys)
  // candidate rows (ring depth <= 16, i.e. quality 5; null otherwise -- then info / sorted are used instead)
  uint16_t* stag;       // br_tag16 of by_key[i]                           [total_bytes]
  uint32_t* rows;       // per position kRowEntries candidates             [16 * total_bytes]
  // checkpoints (lz77_chain.h, Checkpoint): one 128-byte record per 512 bytes of text, and per segment the lowest / highest
  // searched position whose candidate row changed since the segment was parsed last (written by lz77_rows_update, reset by
  // lz77_chain_check for the segments a launch parsed)
  void* checkpoints = nullptr;            // Checkpoint[total_bytes / 512 + 2]
  uint32_t* rows_changed_lo = nullptr;    // [segments]
  uint32_t* rows_changed_hi = nullptr;    // [segments]
  uint32_t splice_lists = 1;              // list launches run the chains that restart from / stop at checkpoints (set per launch by the host)
  uint32_t* dict_items = nullptr;  // optional: the two static-dictionary hash items of every position (k_compute_keys)  [total_bytes + 64]
  uint32_t* changed_slot;  // slot of every changed position                [changed_cap]
  uint32_t* row_ctl;    // device-side control words of lz77_rows_update    [4]
  uint32_t* reset_counts;      // Lz77Params::reset_pos != 0: per key, stored positions in front of reset_vis  [65536]
  const uint32_t* count_base;  // optional, per key: positions stored in front of the text (a later piece of a stream)  [65536]
  uint32_t* run_end;    // optional run table (lz77_run_table), null when the input has no long runs  [total_bytes]
  unsigned long long* smask;  // stored bits of the slots, one word per 64 slots      [total_bytes / 64 + 2]
  uint32_t* gprev;      // per 64 slots: 1 + last stored slot in front of them           [total_bytes / 64 + 2]
  uint8_t* big_tile;    // per 1024 slots: holds slots of a key with >= 65 536 slots  [total_bytes / 1024 + 64]
  // "potential" mask, one bit per slot: some slot of the same key in front of it, within max_backward, carries the same tag.
  // A slot without the bit has an EMPTY candidate row whatever the stored flags are (a row only ever holds tag-equal
  // predecessors), so the validation passes skip it.  Parse-independent; computed on the device the first time a full
  // validation pass is due (pot_state[0] != 0: the mask is there) -- inputs that never need one (text) never pay for it.
  unsigned long long* pot = nullptr;  // [total_bytes / 64 + 2]
  uint32_t* pot_state = nullptr;      // [16], zero at the start of a call: [0] mask there, [1] listed slots, [2] list overflowed
  uint32_t* pot_list = nullptr;       // the slots with the bit, any order  [pot_list_cap]
  uint32_t pot_list_cap = 0;
  // Where flags flipped in the last launch, coarsely: one bit per (hash key, 1 << cell_shift bytes of text), set by
  // lz77_diff_flags* for every changed position; the last word is the "too many to tell" mark.  A candidate row holds
  // stored positions of its key at most max_backward bytes back (1 << cell_shift >= that), so a listed slot whose two
  // cells are clear kept its row (keys whose ring counter can wrap excepted: lz77_rows_update ignores the cells there).
  uint32_t* flip_cells = nullptr;  // [65536 * cells_per_key / 32 + 2]
  uint32_t cell_shift = 0, cells_per_key = 0;
  uint32_t changed_cap; // entries in changed_keys / changed_slot
  // rank-structure chains (qualities 6-9): log of every search (ChainTables::search_log, kSearchLogWords words per
  // position) and the list of searched positions whose candidate list changed in this round (lz77_recheck_searches)
  uint16_t* sorted_tag[2] = {nullptr, nullptr};  // br_tag16 of every entry of sorted[] (needs stag)
  uint32_t* search_log = nullptr;
  uint32_t* recheck_list = nullptr;
  uint32_t* recheck_count = nullptr;
  uint32_t recheck_cap = 0;
  uint8_t* flags[2];    // stored flags, double buffered                   [total_bytes + 64]
  Command* cmds;        // num_segments * cmd_slab_stride
  Segment* segments;    // num_segments
  SegEntry* entries;    // num_segments
  SegExit* exits;       // num_segments
  void* sort_tmp;       // scratch for the radix sort / scans
  size_t sort_tmp_bytes;
};

size_t lz77_sort_tmp_bytes(uint32_t total_bytes);

// hash key of every position (mod.rs:990-992 H5, :1138-1140 H6); also samples how much of the input lies in runs of one
// byte (B.changed_count[8])
void lz77_compute_keys(const Lz77Params& P, const Lz77Buffers& B);
// run_end[p] for every position (see ChainTables::run_end); needs B.run_end allocated
void lz77_run_table(const Lz77Params& P, const Lz77Buffers& B);
// first guess of the stored flags (both buffers): prefix positions, stitch positions and block tails are
// static, everything else is assumed stored
// (prefix_flags_host: optional stored flags of the first prefix_flags_bytes positions, for the continuation of a
// stream after a flush; otherwise the prefix is a custom dictionary that was stored position by position)
void lz77_init_flags(const Lz77Params& P, const Lz77Buffers& B, uint32_t first_block_start, const uint8_t* prefix_flags_host = nullptr,
                     uint32_t prefix_flags_bytes = 0);
// stable sort of positions by key -> by_key / sorted_keys
void lz77_sort_by_key(const Lz77Params& P, const Lz77Buffers& B);
// sorted[rbuf] / info[rbuf] from flags[which] (all keys).  `initial` (optional): flags[which] is still exactly what
// lz77_init_flags wrote, which lets the kernel skip most of the random flag reads.
void lz77_rank_flags(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const RankInitialHint* initial = nullptr);
// Candidate rows (see kRowEntries in lz77_chain.h).  lz77_rows_init: fbits + every row from flags[which] (no validation).
// lz77_rows_update: after a parse launch and lz77_diff_flags(prev, next) -- brings fbits and the rows up to date with
// flags[next] and marks (dirty[k] = 1) the chains that searched a position whose row changed.  Everything is decided on
// the device (few changes: only the rows behind the changed slots; list overflow or a change in a key with >= 65 536
// slots: all rows), nothing is read back.  has_big_keys: some key owns >= 65 536 slots (exact ring counters needed).
void lz77_rows_init(const Lz77Params& P, const Lz77Buffers& B, int which, const RankInitialHint* initial, bool has_big_keys);
void lz77_rows_update(const Lz77Params& P, const Lz77Buffers& B, int prev, int next, const SegGeometry& geo, uint8_t* dirty_dev,
                      bool has_big_keys);
// one round of speculative parsing: segments [first, num_segments) read flags[which] (through
// rank/sorted) and write flags[which ^ 1], cmds and exits
void lz77_parse_round(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, uint32_t first_segment);
// the same over an explicit list of segment indices (device array); flags are read from and written to flags[which]
// sched_dev[k] = 1 for the segments in the list, 2 for segments left to the chain of their predecessor that must be
// redone in any case, 0 otherwise; a chain continues into following segments marked 2, or marked 0 whose entry it
// changes (br_parse_chain), rewrites B.entries for them and marks them 3
void lz77_parse_list(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const uint32_t* list_dev, uint8_t* sched_dev,
                     uint32_t count);
// list rounds move only what changed between host and device:
// B.entries[index_dev[i]] = entries_dev[i] for i < count
void lz77_scatter_entries(const Lz77Buffers& B, const uint32_t* index_dev, const SegEntry* entries_dev, uint32_t count);
// after lz77_parse_list: exits_out[i] = B.exits[list_dev[i]] for i < count, and the segments that chains continued into
// (sched_dev[k] == 3) appended in any order to cont_index / cont_exits / cont_entries, their number in *cont_count
void lz77_gather_results(const Lz77Buffers& B, const uint32_t* list_dev, uint32_t count, const uint8_t* sched_dev, uint32_t num_segments,
                         SegExit* exits_out, uint32_t* cont_count, uint32_t* cont_index, SegExit* cont_exits, SegEntry* cont_entries);
// lists the keys of the positions whose stored flag differs between flags[prev] and flags[next] in B.changed_keys /
// B.changed_count (the count may exceed kChangedCap; only the first kChangedCap entries are kept)
void lz77_diff_flags(const Lz77Params& P, const Lz77Buffers& B, int prev, int next);
// ---- bursts: several list launches in a row without the host resolver in between (candidate rows only).  After a launch
// the device itself chains the exits of the segments it parsed into the entries of their successors inside the block and
// schedules the next launch: segments whose candidates changed (lz77_rows_update), whose entry changed, or that were left
// to a chain that did not get there.  What it cannot judge -- block starts (meta-block books, extend_last_command), the
// static-dictionary throttle, literal sprees carried by arithmetic -- it leaves alone: the host resolver runs after the
// burst over everything, exactly as after a single launch, and alone decides when the parse is the fixed point.
struct BurstBuffers {
  uint8_t* sched = nullptr;        // [segments] as for lz77_parse_list
  uint8_t* cand_dirty = nullptr;   // [segments] marks of lz77_rows_update
  uint8_t* entry_dirty = nullptr;  // [segments] the state handed over by the predecessor differs from the entry last used
  uint8_t* touched = nullptr;      // [segments] parsed at least once since the host last cleared it
  uint8_t* stale = nullptr;        // [segments] parsed by the last launch: flags[which ^ 1] lags behind flags[which] there and nowhere else
  SegEntry* new_entries = nullptr; // [segments] valid where entry_dirty
  uint32_t* list = nullptr;        // [segments] the next launch
  uint32_t* counters = nullptr;    // [0] length of list, [1] number of touched segments (lz77_gather_touched)
};
// after lz77_parse_list(list, sched): marks touched[], entry_dirty[] / new_entries[] (see above); the rows_changed_lo / _hi
// marks of the segments the launch parsed are reset -- call it BEFORE the lz77_rows_update that follows the launch
void lz77_chain_check(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U);
// Between the launches of a burst the two flag arrays differ only inside the segments the last launch parsed (U.stale, set by
// lz77_chain_check): instead of copying all flags before a launch and comparing all of them after it --
// lz77_flags_catch_up: flags[dst] := flags[src] inside the stale segments, marks cleared (before the launch that writes flags[dst]);
// lz77_diff_flags_touched: lz77_diff_flags restricted to the stale segments (after lz77_chain_check)
void lz77_flags_catch_up(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, int src, int dst);
void lz77_diff_flags_touched(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, int prev, int next);
// rows_changed_lo / _hi of every segment := none (after a launch that parsed all of them)
void lz77_reset_rows_changed(const Lz77Params& P, const Lz77Buffers& B);
// the checkpoints of the listed segments are dropped (their parse now stands for another distance cache at the entry than
// the records were taken with, Lz77Stage::RecheckCacheOnly)
void lz77_drop_checkpoints(const Lz77Params& P, const Lz77Buffers& B, const uint32_t* list_dev, uint32_t count);
// counters[0] = number of segments lz77_burst_schedule would list (nothing is changed)
void lz77_burst_count(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U);
// list := segments with cand_dirty | entry_dirty | (sched == 2: left to a chain that stopped short); their sched := 1, all
// others 0; B.entries := new_entries where entry_dirty; the marks are cleared; counters[0] = length
void lz77_burst_schedule(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U);
// index / exits / entries of the touched segments, in any order; counters[1] = their number; touched[] is cleared
void lz77_gather_touched(const Lz77Params& P, const Lz77Buffers& B, const BurstBuffers& U, uint32_t* index_out, SegExit* exits_out, SegEntry* entries_out);
// marks (dirty[k] = 1) the segments that searched a position whose candidate list differs between the rank
// structures rbuf_old and rbuf_new
struct SegGeometry {
  uint32_t prefix_bytes, first_block_start, block_bytes, num_blocks, num_segments, block_size /* ring depth */;
  // per input block (segments may be cut differently in every block): index of its first segment [num_blocks + 1]
  // and the bytes per segment [num_blocks]
  const uint32_t* block_first_segment;
  const uint32_t* block_segment_bytes;
};
void lz77_validate(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf_old, int rbuf_new, const SegGeometry& geo,
                   uint8_t* dirty_dev);
// After lz77_validate / lz77_rerank_keys with B.recheck_list set: those two only LIST the searched positions whose
// candidate list changed in a way that could matter; this repeats the search of every listed position under the rank
// structures rbuf (br_recheck_search) and marks the segment dirty when it finds something else than the chain found.
void lz77_recheck_searches(const Lz77Params& P, const Lz77Buffers& B, int rbuf, const SegGeometry& geo, uint8_t* dirty_dev);
// incremental form of rank + validate: re-ranks the listed chunks (all slots of every changed key, cut into pieces of
// kRerankChunk slots, keys in ascending order) of sorted[rbuf] / info[rbuf] in place from flags[which] and marks the
// segments that searched a position of those keys whose candidate list changed.  sums_dev: num_chunks words of scratch.
static constexpr uint32_t kRerankChunk = 4096;
struct RerankChunk {
  uint32_t key_lo;     // first slot of the key
  uint32_t begin, end; // slots of this chunk
  uint32_t first_sum;  // index (in sums) of the key's first chunk
  uint32_t my_sum;     // index of this chunk
};
void lz77_rerank_keys(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const RerankChunk* chunks_dev, uint32_t num_chunks,
                      uint32_t* sums_dev, const SegGeometry& geo, uint8_t* dirty_dev);
// same kernel over an explicit list of (segment, entry) pairs (used for the warm-up dry run)
void lz77_parse_custom(const Lz77Params& P, const Lz77Buffers& B, int which, int rbuf, const Segment* segments_dev,
                       SegEntry* entries_dev, SegExit* exits_dev, uint32_t count);
// Would a literal-only parse of these segments (no match found at any searched position, flags[which] says which were
// searched) stay literal-only with another distance cache at its entry?  ok_dev[i] = 1 when none of the P.ndist
// candidate distances derived from it (4 at qualities 5-6, 10 at 7-8) yields even a two-byte match at any searched
// position (FindLongestMatch accepts cache candidates from length 2, mod.rs:1707-1741) -- conservative: 0 only means
// "parse it again".  H5 / H6 hashers only.
struct CacheCheck {
  uint32_t segment;
  int32_t cache[4];
};
void lz77_check_cache(const Lz77Params& P, const Lz77Buffers& B, int which, const CacheCheck* items_dev, uint32_t count, uint8_t* ok_dev);
// out[key] = (base ? base[key] : 0) + number of positions < upto of that key whose stored bit is set in flags[which]
void lz77_key_counts(const Lz77Params& P, const Lz77Buffers& B, int which, uint32_t upto, uint32_t* out_dev, bool with_base = true);
// accumulated device time (HIP events) of the parse kernel launches since the last call
// work[3] (optional): what those launches did -- positions walked, searches, commands written (all chains, re-parses included)
void lz77_parse_timing(double* total_ms, uint32_t* launches, uint64_t* segments, uint64_t* work = nullptr);
// every-13th-byte histograms of `count` spans {start, bytes} of the text at once (should_compress of all literal-only meta-blocks
// of a resolver pass): out[r * 256 + v].  ranges and out are HOST arrays; runs beside whatever the calling thread's stream holds
// and returns when the histograms are there.
void lz77_sample_histograms(const uint8_t* text_dev, const uint32_t* ranges, uint32_t count, uint32_t* out);
// every-13th-byte literal histogram for should_compress (encode.rs:1325-1354)
void lz77_sample_histogram(const uint8_t* text, uint32_t start, uint32_t bytes, uint32_t* histo256_dev);
// gathers the per-segment command slabs into one array: out[offsets[k] + i] = slab_k[i]
void lz77_gather_commands(const Lz77Params& P, const Lz77Buffers& B, uint32_t num_segments, const uint32_t* offsets_dev,
                          const uint32_t* counts_dev, Command* out);

// ---- qualities 10 / 11 (zopfli_device.h): the H10 binary-tree hasher and the Zopfli shortest-path parse, block by block ----
struct ZopfliJob {
  uint32_t quality = 10, lgwin = 22;
  uint32_t use_dictionary = 1;
  uint32_t dist_alphabet_size = 0;
  uint32_t block_bytes = 0;      // 1 << lgblock: bound of everything that is sized per block
  // device memory (Lz77Stage owns it)
  uint32_t* buckets = nullptr;   // [1 << 17]
  uint32_t* forest = nullptr;    // [2 << lgwin]
  void* nodes = nullptr;         // ZNode[block_bytes + 1]
  float* literal_costs = nullptr;  // [block_bytes + 2]
  float* cost_dist = nullptr;    // [dist_alphabet_size + 64]
  float* cost_cmd = nullptr;     // [704]
  unsigned long long* matches = nullptr;  // quality 10: [128]; quality 11: [128 * block_bytes]
  uint32_t* num_matches = nullptr;        // quality 11: [block_bytes]
  Command* tmp_cmds = nullptr;   // quality 11: [block_bytes / 2 + 8]
  uint32_t* histo = nullptr;     // [2048]
  // the trees of different hash keys side by side (br_zopfli_matches_of_group): nodes of the block's own positions, what the
  // hasher looked like in front of the block (a block whose matches do not hold is parsed again the sequential way), which
  // positions got a node, and the control words of the block (ZBlockCtl)
  uint32_t* forest_new = nullptr;   // [2 << lgwin]
  uint32_t* forest_bak = nullptr;   // [2 << lgwin]
  uint32_t* buckets_bak = nullptr;  // [1 << 17]
  uint8_t* rerooted = nullptr;      // [block_bytes]
  uint32_t* ctl = nullptr;          // [16]
};
// empties the hasher: buckets = invalid position, forest = 0 (InitializeH10, hash_to_binary_tree.rs:149-190)
void lz77_zopfli_init(const ZopfliJob& J);
// the trees of the piece in front (buckets_src / forest_src, as ZopfliJob's) become this job's: copied, and every position in them
// moved down by `delta` -- the text of this piece starts `delta` bytes further into the stream (a multiple of the window size:
// node slots stay what they were); positions in front of the new text lie beyond every window and become the invalid position
void lz77_zopfli_import(const ZopfliJob& J, const uint32_t* buckets_src, const uint32_t* forest_src, uint32_t delta);
// HasherPrependCustomDictionary (encode.rs:1163-1194): the positions [0, dict_bytes - 127) of the text go into the trees
void lz77_zopfli_prepend(const Lz77Params& P, const Lz77Buffers& B, const ZopfliJob& J, uint32_t dict_bytes);
// input block `block` (= segment `block`: one segment per block) with the entry B.entries[block]: commands into its slab, exit into
// B.exits[block].  The blocks of a stream go through here in order.  Needs the positions sorted by 16-bit hash key (lz77_compute_keys
// with bucket_bits 16 + lz77_sort_by_key: two H10 keys per group).  Returns true if the block had to be parsed the sequential way.
bool lz77_zopfli_block(const Lz77Params& P, const Lz77Buffers& B, const ZopfliJob& J, uint32_t block);

// ---- live chains (lz77_live.h): chains that parse a span of input blocks on a private copy of the reference's bucket rings ----
struct LiveBuffers {
  uint16_t* num = nullptr;       // [tables][1 << bucket_bits]
  uint32_t* buckets = nullptr;   // [tables][(1 << bucket_bits) << block_bits]
  uint32_t* slot_of = nullptr;   // slot of every position in (key, position) order (inverse of by_key)   [total_bytes]
  uint32_t* rank[2] = {nullptr, nullptr};   // per flags buffer: prefix count of the stored slots          [total_bytes + 1]
  uint32_t* entry[2] = {nullptr, nullptr};  // per flags buffer: ring entries of the stored slots, compacted [total_bytes]
  uint8_t* changed_key = nullptr;           // [65536]
  LiveBlockState* state = nullptr;          // [blocks] meta-block books at every block entry (ChainTables::live_state)
  uint32_t tables = 0;                      // one per span
  uint32_t span_blocks = 1;                 // blocks per span; table t serves the blocks [t * span_blocks, (t + 1) * span_blocks)
};
// slot_of from by_key (once per text)
void lz77_live_slots(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L);
// rank[which] / entry[which] from flags[which]
void lz77_live_index(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which);
// for every listed block first[i]: its span's table := the rings as a search at text position start[i] finds them under
// flags[which] (needs lz77_live_index(which))
void lz77_live_materialise(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which, const uint32_t* first_dev,
                           const uint32_t* start_dev, uint32_t count);
// one chain per listed block first[i]: parses from there to the end of its span (br_parse_live); flags go to flags[which ^ 1],
// every search is logged in B.search_log, the entries the chains derive for the later blocks of a span to B.entries
void lz77_live_parse(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int which, const uint32_t* first_dev, uint32_t count);
// Repeats logged searches against the rings that flags[next] imply (needs lz77_live_index(next)) and sets dirty[block] = 1 where
// one comes out differently.  prev < 0: every search of every block; otherwise the searches of the blocks with
// reparsed[block] != 0 and, elsewhere, those whose hash key had a flag change between flags[prev] and flags[next].
void lz77_live_verify(const Lz77Params& P, const Lz77Buffers& B, const LiveBuffers& L, int prev, int next, const SegGeometry& geo,
                      const uint8_t* reparsed_dev, uint8_t* dirty_dev);

// applies extend_last_command / trailing insert-only fix-ups to the gathered commands
void lz77_patch_commands(Command* cmds, const CmdPatch* patches_dev, uint32_t n);

// grow-only array in page-locked host memory (contents are not preserved by resize_discard).  Copies from and to it are
// truly asynchronous: wait (dev_sync / dev_wait_mark) before rewriting the source of an upload or letting the array go.
template <typename T>
struct PinnedArray {
  T* ptr = nullptr;
  size_t cap = 0, n = 0;
  PinnedArray() = default;
  PinnedArray(const PinnedArray&) = delete;
  PinnedArray& operator=(const PinnedArray&) = delete;
  ~PinnedArray() { dev_host_free(ptr); }
  void resize_discard(size_t count) {
    if (count > cap) {
      dev_host_free(ptr);
      cap = count + count / 4 + 16;
      ptr = (T*)dev_host_alloc(cap * sizeof(T));
    }
    n = count;
  }
  void assign(size_t count, const T& v) {
    resize_discard(count);
    for (size_t i = 0; i < count; ++i) ptr[i] = v;
  }
  T* data() { return ptr; }
  const T* data() const { return ptr; }
  size_t size() const { return n; }
  T& operator[](size_t i) { return ptr[i]; }
  const T& operator[](size_t i) const { return ptr[i]; }
};

}  // namespace brotli_mi355x
#endif
