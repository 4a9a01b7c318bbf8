// lz77_types.h -- plain-old-data shared by the host driver and the HIP kernels of the LZ77 stage.
//
// Domain vocabulary follows the reference (rust-brotli): "block" = one input block of 1<<lgblock
// bytes handed to BrotliCreateBackwardReferences (src/enc/encode.rs:2438), "command" = one
// insert&copy record (src/enc/command.rs:12-21), "dist cache" = the 4 last distances.
// New here: "segment" = a slice of a block that one wavefront parses speculatively ("chain").
#ifndef BROTLI_MI355X_LZ77_TYPES_H_
#define BROTLI_MI355X_LZ77_TYPES_H_

#include <stdint.h>

namespace brotli_mi355x {

// src/enc/command.rs:12-21 (16 bytes)
struct Command {
  uint32_t insert_len_;
  uint32_t copy_len_;   // low 25 bits copy length, high 7 bits (copy_code - copy_len)
  uint32_t dist_extra_;
  uint16_t cmd_prefix_;
  uint16_t dist_prefix_;  // low 10 bits distance code, high 6 bits number of extra bits
};

// Per-job constants of the backward-reference search (H5 / H5q5 / H6 AdvHasher family,
// src/enc/backward_references/mod.rs:919-1813, parameters chosen by encode.rs:834-893).
struct Lz77Params {
  uint32_t total_bytes;         // M: dictionary prefix + input, positions are offsets into this text
  uint32_t prefix_bytes;        // bytes in front of the input: custom dictionary, or the part of the stream already
                                // encoded before a BROTLI_OPERATION_FLUSH
  uint32_t dict_break;          // ring_buffer_break: matches may not straddle the end of a custom dictionary
                                // (mod.rs:42-54); 0 when the prefix is earlier stream data
  uint32_t ring_mask;           // (1 << (1 + max(lgwin, lgblock))) - 1   (encode.rs:587-601)
  uint32_t max_backward_limit;  // (1 << lgwin) - 16                      (mod.rs:2393)
  uint32_t hasher_kind;         // 5 = 32-bit hash of 4 bytes, 6 = 64-bit hash of hash_len bytes
  uint32_t bucket_bits;
  uint32_t block_bits;          // ring depth per key = 1 << block_bits
  uint32_t hash_len;            // H6 only
  uint32_t ndist;               // num_last_distances_to_check (4, 10 or 16)
  uint32_t htl;                 // HashTypeLength == StoreLookahead (4 or 8)
  uint32_t literal_byte_score;   // H9 scores with the full value and shifts the sum (mod.rs:685-708)
  uint32_t score_per_byte;      // literal_byte_score >> 2 (135)
  uint32_t use_dictionary;      // static dictionary allowed (params.use_dictionary)
  uint32_t spree_window;        // LiteralSpreeLengthForSparseSearch: 64 (q<9) or 512
  uint32_t dist_max_distance;   // params.dist.max_distance
  uint32_t quality;
  uint32_t num_segments;
  uint32_t cmd_slab_stride;     // commands reserved per segment
  uint32_t dist_postfix_bits;   // params.dist (encode.rs:2169-2190)
  uint32_t num_direct_distance_codes;
  // Hasher reset inside this text (0 = none): the reference empties its hash table when its 32-bit position wraps
  // (encode.rs:1623-1631, 1705-1710, 2472-2474).  Searches at positions >= reset_pos see only what was inserted from
  // reset_vis on (= reset_pos - 3: StitchToPreviousBlock re-inserts the last three positions of the block in front).
  uint32_t reset_pos, reset_vis;
  // First position from which the entries that the reference's StoreRangeOptBatch writes into an H5 bucket ring are MASKED
  // positions (mod.rs:1163-1232: the stream position has passed the ring-buffer size); such an entry ends the bucket walk
  // of every later search (kFlagMasked, lz77_chain.h).  kNeverMasked: not an H5 hasher, or the text ends inside the first
  // ring-buffer revolution of the stream.  Anything else is parsed by live chains (lz77_live.h).
  uint32_t masked_from;
  // the meta-block flush rule (encode.rs:2454-2477) for chains that walk from block to block by themselves (live chains)
  uint32_t block_bytes;          // 1 << lgblock
  uint32_t max_metablock_bytes;  // MaxMetablockSize; the limits on literals and commands are an eighth of it
};
static constexpr uint32_t kNeverMasked = 0xffffffffu;

// What a live chain knows about the open meta-block when it enters a block (Lz77Stage::Resolve keeps the same books):
// where it started, the commands and literals in it so far, and its last command -- extend_last_command (encode.rs:360-400)
// runs at the start of a block only if there is one, nothing is pending behind it and its distance is the last distance.
struct LiveBlockState {
  uint32_t mb_start, mb_cmds, mb_lits;
  uint32_t last_valid, last_dist_code, last_copy_len;
  int32_t saved_cache[4];  // the distance cache at its start (a meta-block stored uncompressed hands it on, encode.rs:1994)
  uint32_t pad[2];
};
// candidate rows (lz77_chain.h): entries per position, end-of-row marker
static constexpr uint32_t kRowEntries = 16;
static constexpr uint32_t kRowEnd = 0xffffffffu;

enum SegmentFlags : uint32_t {
  kSegFirstInBlock = 1u,
  kSegLastInBlock = 2u,
  kSegTailStitched = 4u,  // next block's StitchToPreviousBlock will store blk_end-3..blk_end-1
  kSegWarmup = 8u,        // dry run over the tail of a segment: only the exit state matters, nothing is written
};

// Static geometry of one segment.
struct Segment {
  uint32_t start;      // first position owned (loop-top positions >= start belong here)
  uint32_t end;        // one past
  uint32_t blk_start;  // start of the enclosing input block (before extend_last_command)
  uint32_t blk_end;    // pos_end of the enclosing block
  uint32_t flags;
  uint32_t cmd_base;   // index of this segment's command slab
  uint32_t block_index;
  uint32_t cmd_cap;    // commands its slab can hold
};

// What covers the positions between a segment's start and the position where its chain takes over.
enum HeadKind : uint32_t {
  kHeadNone = 0,
  kHeadCopy = 1,      // copy that started at head_base: base + 1 is 3 if it was probed (head_p1) else "not stored",
                      // the rest is stored up to store_end (StoreRange, mod.rs:2519-2527)
  kHeadUnstored = 2,  // extend_last_command / end-of-block flush: nothing is stored
  kHeadVec4 = 3,      // Store4Vec4 from head_base: every 4th position (mod.rs:2538-2541)
  kHeadEven4 = 4,     // StoreEvenVec4 from head_base: every 2nd position (mod.rs:2542-2545)
};

// Parse state handed to a chain when it starts (speculated, then confirmed by the host resolver).
// One record of the search log (ChainTables::search_log, rank-structure chains): [0..3] the distance cache the search ran
// with, [4] len, [5] distance, [6] score of what the cache + ring stages found, [7] bit 0 found, bit 1 stored.
static constexpr uint32_t kSearchLogWords = 8;

struct SegEntry {
  uint32_t pos;         // loop-top position where the true parse enters this segment
  uint32_t apply;       // apply_random_heuristics (mod.rs:2407, 2492)
  int32_t cache[4];     // dist_cache[0..3]
  uint32_t insert_len;  // informational: pending literals carried in (not used by the chain)
  uint32_t ext_allowed; // first-in-block only: extend_last_command may run (encode.rs:2435-2437)
  uint32_t dict_lookups;  // static-dictionary throttle counters at entry (mod.rs:1957-1960)
  uint32_t dict_matches;
  uint32_t ext_max_distance;  // max_distance for extend_last_command
  uint32_t dict_exact;  // the throttle counters are the true ones (not a guess): br_parse_chain may rely on them
  // How to flag the positions [start, pos) of this segment, which the LAST step of the previous chain covers (a copy,
  // an extension, a sparse-store jump ...).  Every position's flag is written by the chain of the segment it lies in,
  // so that no two chains ever write the same byte.  See HeadKind.
  uint32_t head_kind, head_base, head_p1;
  uint32_t pad;
};

struct SegExit {
  uint32_t pos;
  uint32_t apply;
  int32_t cache[4];
  uint32_t insert_len;
  uint32_t n_cmds;
  uint32_t n_lits;          // sum of the LOCAL insert lengths of the emitted commands (carry-in excluded)
  uint32_t ext_len;         // bytes consumed by extend_last_command (first-in-block)
  uint32_t dict_lookups;    // counters at exit
  uint32_t dict_matches;
  uint32_t last_dist_code;  // restored distance code of the last emitted command (0xffffffff: none)
  uint32_t bad_commands;    // copies shorter than 2 bytes (a match cut to 1 byte at the end of a custom dictionary): the
                            // reference cannot encode them (GetCopyLengthCode, command.rs:91-93) and fails
  uint32_t n_searches;
  uint32_t last_copy_len;   // copy_len of last emitted command (low 25 bits)
  uint32_t dict_mode;       // 0 no consult, 1 alive at every consult, 2 dead at every consult, 3 mixed,
                            // 4 ran without the dictionary and without the books (exact "off" at the entry)
  int32_t dict_maxdef;      // mode 1: max over consults of (local lookups - 128 * local matches)
  uint32_t n_pushes;        // number of dist-cache pushes in this segment, saturated at 4
  uint32_t tail_kind, tail_base, tail_p1;  // the step that carried the parse past the segment end (-> next entry's head_*)
  uint32_t n_pushes_all;    // the same count, not saturated (splicing two parses at a checkpoint adds and subtracts it)
  uint32_t dict_entry_lookups, dict_entry_matches;  // the throttle counters this parse was entered with (dict_lookups / _matches count on from them)
};

// ---- checkpoints: parsing a segment again without parsing all of it --------------------------------------------------------
// A re-parse launch lasts as long as one chain needs for its segment (plus what it walks into), however few chains there
// are -- and most segments are parsed again for a reason that touches a fraction of them: a candidate row changed at one
// searched position, or the entry state changed and the parse falls back into step with the old one after a hundred bytes.
// So every (candidate-row) parse leaves a record of its complete loop-top state at the first position it reaches at or
// behind every kCheckpointStride-byte boundary inside its segment.  A later parse of the same segment
//   * with the same entry, parsed again because rows changed from position p on, RESTARTS from the last record in front
//     of p (everything up to there -- commands, flags, state -- is what it would produce again), and
//   * whatever it started from, STOPS at a boundary where it finds itself in exactly the recorded state (position, pending
//     literals, spree countdown, distance cache, and a static-dictionary regime under which the old parse of the rest holds),
//     provided no row changed behind that point: the rest of the old parse -- its commands, its flags, its exit -- stands,
//     and the two halves are spliced (br_parse_segment).
// Records behind a splice point are dropped (their running counts belong to the old head); the next full parse of the
// segment writes them afresh.
static constexpr uint32_t kCheckpointStride = 512;
static constexpr uint32_t kCheckpointValid = 0x43503031u;  // "CP01"
struct Checkpoint {  // 128 bytes
  uint32_t pos, insert_len, apply;
  int32_t dc[4];
  uint32_t n_cmds, n_lits, n_searches, n_pushes, n_bad, last_dist_code, last_copy_len, ext_len;
  uint32_t tail_kind, tail_base, tail_p1;
  uint32_t d_lookups, d_matches, d_mode;
  int32_t d_maxdef;
  uint32_t d_vlookups, d_vwould;
  int32_t d_vmaxdef;
  uint32_t entry_lookups, entry_matches, no_dict;
  uint32_t valid;
  uint32_t pad[3];
};
static_assert(sizeof(Checkpoint) == 128, "one checkpoint = two 64-byte lines");

// marks in the sched[] array of a list launch
static constexpr uint8_t kSchedOwn = 1;        // has a chain of its own in this launch (its entry changed, or it is owed a full parse)
static constexpr uint8_t kSchedLeft = 2;       // left to the chain in front of it; must be redone whatever state that chain arrives with
static constexpr uint8_t kSchedWalked = 3;     // a chain walked into it (set by the chain)
static constexpr uint8_t kSchedOwnRows = 5;    // has a chain of its own; its entry is the one it was parsed with last, rows changed

// Post-parse fix-ups of the gathered command array.
struct CmdPatch {
  uint32_t index;  // command index in the gathered array
  uint32_t kind;   // 0: extend_last_command (copy_len += value, encode.rs:360-400); 1: insert-only command
                   //    carrying `value` literals (Command::init_insert, command.rs:38-44); 2: insert_len += value
  uint32_t value;
  uint32_t pad;
};

}  // namespace brotli_mi355x
#endif
