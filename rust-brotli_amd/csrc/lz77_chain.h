// lz77_chain.h -- one speculative parse chain of the backward-reference search.
//
// A chain is the greedy/lazy parse of rust-brotli's CreateBackwardReferences
// (src/enc/backward_references/mod.rs:2376-2552) restricted to one segment of one input block,
// executed by ONE 64-lane wavefront: the control flow is wave-uniform, the per-position search
// (AdvHasher::FindLongestMatch, mod.rs:1684-1812) spreads its <= ndist + block_size candidates
// over the lanes (one candidate per lane: load, compare, extend), then folds them in the
// reference's exact order.
//
// The reference's bucket ring (num[]/buckets[], mod.rs:932-941) is replaced by a parse-independent
// surrogate: positions sorted by (key, position), a per-position "stored" flag and the prefix sum of
// the flags in sorted order (rank).  The candidates the ring would hold when position p is searched
// are the min(block_size, count mod 65536) stored positions that precede p in its key's list.
// The flags of positions handled by OTHER chains come from the previous round (Jacobi iteration);
// the host resolver repeats rounds until entries and flags are a fixed point, which is the
// sequential parse.
//
// The same source is compiled for the host (BROTLI_HOST_EMU) to test the algorithm without a GPU:
// there a "wave" is one lane and the candidate loops run serially.  The emulation build is test
// infrastructure; the product library only contains the gfx950 code.
#ifndef BROTLI_MI355X_LZ77_CHAIN_H_
#define BROTLI_MI355X_LZ77_CHAIN_H_

#include "lz77_types.h"

#if defined(BROTLI_HOST_EMU)
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define BR_DEV inline
#define BR_LANE 0
#define BR_NLANES 1
#define BR_SYNC() ((void)0)
#define BR_WAVE_SYNC() ((void)0)
#define BR_ATOMIC_INC(ptr) ((*(ptr))++)
#define BR_UNIFORM(x) (x)
#define BR_READLANE(x, lane) (x)
#define BR_SCALAR 1
#define BR_UNROLL
#else
#include <hip/hip_runtime.h>
#define BR_DEV __device__ __forceinline__
#define BR_LANE ((int)threadIdx.x)
#define BR_NLANES 64
#define BR_SYNC() __syncthreads()
// Every chain kernel runs ONE wavefront per workgroup (__launch_bounds__(64)): the LDS operations of a wavefront are executed in
// program order, so between the chain's own LDS writes and reads nothing has to be waited for -- only the compiler must not move
// them across each other.  (__syncthreads() drains vmcnt as well: the flag stores of the parse loop.)
#define BR_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define BR_ATOMIC_INC(ptr) atomicAdd((ptr), 1u)
// The control flow of a chain is wave-uniform by construction, but values that come back from vector loads or
// cross-lane operations live in VGPRs and would make the compiler predicate every branch with exec masks.
// BR_UNIFORM moves such a value into an SGPR (all lanes hold the same value); BR_READLANE picks one lane's value.
#define BR_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define BR_READLANE(x, lane) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (int)(lane)))
#define BR_SCALAR 0
#endif

#include "lz77_live.h"
#include "entropy_device.h"

namespace brotli_mi355x {

static constexpr int kMaxCandidatesH9 = 16 + 256;  // ndist <= 16, ring depth <= 256 (H9, quality 9)
static constexpr int kMaxCandidatesAdv = 16 + 128;   // ring depth <= 128 (quality <= 8)
static constexpr int kMaxCandidatesDeep = 16 + 512;  // the 512-deep H5 / H6 rings of quality 11 + Q9_5 (ChainScratchT<.., kDeep>)
// per-position flag byte: bit 0 = the position is in the hash table, bit 1 = FindLongestMatch ran on it
static constexpr uint8_t kFlagStored = 1, kFlagSearched = 2;
// the position went into its bucket ring as a MASKED position (StoreRangeOptBatch past the first ring-buffer revolution,
// Lz77Params::masked_from): a stored entry like any other for the ring counters, but the bucket walk of every later search
// ends when it reaches it.  Only ever set together with kFlagStored.
static constexpr uint8_t kFlagMasked = 4;
// (written by live chains only, lz77_live.h: where masked entries exist the parse runs on private copies of the rings)
// how the rank structures (sorted[], qualities 6..8) hold such an entry: position | kMaskedEntry.  `cur - q <= max_backward`
// fails for it in br_probe_pair, which is where the reference's walk breaks too.
static constexpr uint32_t kMaskedEntry = 0x80000000u;
static constexpr uint32_t kMinScore = 30 * 8 * 8 + 100;  // mod.rs:2408-2410

struct ChainTables {
  const uint8_t* text;         // dictionary prefix + input (+ >= 16 bytes of zero padding)
  const uint32_t* info;        // per position: {rank among the stored positions in (key,pos) order,
                               //                number of stored positions of the same key before it}
  const uint32_t* sorted;      // stored positions in (key,pos) order
  // optional, parallel to sorted: 16-bit hash of the first four bytes at every entry (br_tag16).  A ring entry takes part in
  // the search only if its first four bytes equal those at the searched position (FindMatchLengthWithLimitMin4,
  // static_dict.rs:134-147): entries with another tag are not even fetched.  Deep rings (256 entries at quality 9) hold
  // mostly such entries -- 15 hash bits for four bytes -- and their text gathers were what the parse was waiting for.
  const uint16_t* sorted_tag = nullptr;
  const uint32_t* rows;        // kRows chains: per position kRowEntries candidate positions, newest first, 0xffffffff-terminated
  const uint32_t* dict_items = nullptr;  // optional, kRows chains: per position the two static-dictionary hash items (Lz77Buffers::dict_items)
  uint8_t* flags_next;         // stored flags produced by this round
  Command* cmds;
  const uint16_t* dict_hash;   // kStaticDictionaryHash (src/enc/dictionary_hash.rs)
  const uint8_t* dict_data;    // RFC 7932 dictionary
  const uint32_t* dict_offsets_by_length;
  const uint8_t* dict_size_bits_by_length;
  uint32_t dist_postfix_bits;
  uint32_t num_direct_distance_codes;
  // measurement (bench.py roofline): per launch-set totals of what the chains actually did -- [0] positions walked,
  // [1] searches, [2] commands written.  Null in the host emulation.
  unsigned long long* work;
  // optional (inputs with long runs of one byte, e.g. zero fill): run_end[p] = first position behind p whose byte differs
  // from text[p].  Lets the match-length code jump over a run instead of comparing it 32 bytes at a time for every one
  // of the ~20 candidates of a position (all of which match to the end of the block inside a run).
  const uint32_t* run_end;
  // optional (rank-structure chains, qualities 6-9): kSearchLogWords words per position, written by every search of a
  // real (not dry-run) parse -- the distance cache it ran with and what the cache + ring stages found.  Lets the
  // validation after a flag change repeat single searches (lz77_recheck_searches) instead of re-parsing every segment
  // whose candidate lists were touched.
  uint32_t* search_log = nullptr;
  // live chains (lz77_live.h): table k belongs to the chain of segment k
  const uint16_t* keys = nullptr;     // hash key of every position
  uint16_t* live_num = nullptr;       // [tables][1 << bucket_bits]
  uint32_t* live_buckets = nullptr;   // [tables][(1 << bucket_bits) << block_bits]
  LiveBlockState* live_state = nullptr;  // [blocks]: the meta-block books at the entry of every block (read for the first one)
  EntropyTables logs = {nullptr, nullptr};  // should_compress in the live chain (encode.rs:1325-1354)
  // checkpoints (candidate-row chains, see Checkpoint): one record per kCheckpointStride bytes of text; and per segment the
  // lowest / highest searched position whose candidate row changed since the segment was parsed last (lz77_rows_update)
  Checkpoint* checkpoints = nullptr;
  const uint32_t* rows_changed_lo = nullptr;
  const uint32_t* rows_changed_hi = nullptr;
  uint32_t splice_off = 0;  // bit 0: never restart from a checkpoint, bit 1: never stop at one
};

static constexpr uint32_t kInfoWindow = 64;
static constexpr uint32_t kMaxContinuation = 4;

// ---- candidate rows (the hash table of the reference, indexed by POSITION) -------------------------------------------
// For ring depth 16 (quality 5) the candidates of every position are materialised once per flag state: rows[p] holds the
// positions FindLongestMatch would find in bucket key(p) when it searches p -- the (up to) 16 most recent stored
// same-key positions in front of p, nearest first, cut at the first one farther away than max_backward (the bucket walk
// breaks there, mod.rs:1769-1776) -- minus those whose first four bytes certainly differ from the four bytes at p: a ring
// entry takes part in the search only through FindMatchLengthWithLimitMin4 (static_dict.rs:134-147), which returns 0
// for them, so they can be dropped without any effect.  "Certainly differ" = a 16-bit hash of the four bytes (br_tag16)
// differs.  The rows of a segment are contiguous in memory: the chain streams them through an LDS window instead of
// chasing rank -> bucket row through two dependent random loads, and a flag change touches the rows of the next <= 16
// stored positions of its key only (lz77_update_rows).
#if !defined(BR_ROW_WINDOW)
#define BR_ROW_WINDOW 32
#endif
static constexpr uint32_t kRowWindow = BR_ROW_WINDOW;  // positions held in LDS
BR_DEV uint32_t br_tag16(uint32_t first_four_bytes) { return (first_four_bytes * 0x9E3779B1u) >> 16; }

// kDeep: the candidate scratch of the 512-deep rings, an instantiation of its own so that the kernels of quality 6-8 keep
// their registers and LDS (with the deep scratch they would run at half the occupancy)
template <bool kH9, bool kRows = false, bool kDeep = false>
struct ChainScratchT {  // one per wavefront (LDS on the device)
  static constexpr int kMaxCandidates = kRows ? 16 + (int)kRowEntries : (kH9 ? kMaxCandidatesH9 : (kDeep ? kMaxCandidatesDeep : kMaxCandidatesAdv));
  // !kRows: rank records (info) of positions [win_base, win_base + kInfoWindow), two words each
  //  kRows: candidate rows of positions [win_base, win_base + kRowWindow)
  alignas(16) uint32_t win[kRows ? kRowWindow * kRowEntries : kInfoWindow * 2];
  uint16_t dictwin[kRows ? kRowWindow * 2 : 2];  // kRows: the two static-dictionary hash items of every window position
  uint32_t dictsrc[kRows ? kRowWindow * 2 : 1];  // kRows: where the word of each item starts in the dictionary (offset into dict_data)
  int32_t dc[16];                // distance cache incl. the derived entries (mod.rs:632-651); lanes index it by candidate
  uint32_t cand_prev[2][kMaxCandidates + 2];  // [probe slot][candidate]; the two dictionary probes come last
  uint32_t cand_len[2][kMaxCandidates + 2];
  // !kRows, device: the candidates whose text has to be fetched (slot numbers, see br_probe_pair)
  uint16_t fetch_list[kRows ? 2 : 2 * (kMaxCandidates + 2)];
  uint32_t keep[12];  // br_parse_chain: what the previous parse of the segment being redone left behind
  // kBrotliDictionaryOffsetsByLength, copied in by br_parse_chain: the dictionary lanes of a probe look up where their word
  // starts, and a load from the table in global memory was a whole memory round trip in front of the candidate fetch
  uint32_t dict_off[kRows ? 32 : 1];
};

struct SearchResult {
  uint32_t len, len_x_code, distance, score;
  bool found;
  bool stored;  // false only for H9 when a cache match reaches the ring-buffer end: the bucket stage, and with it the
                // insertion of the position, is skipped (mod.rs:789-791, 868-870)
};

BR_DEV uint32_t br_load32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
BR_DEV uint64_t br_load64(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
BR_DEV uint32_t br_log2_floor_nonzero(uint32_t v) { return 31u ^ (uint32_t)__builtin_clz(v); }

// FindMatchLengthWithLimit, src/enc/static_dict.rs:125-132 (8-byte XOR + ctz extension)
BR_DEV uint32_t br_match_len(const uint8_t* a, const uint8_t* b, uint32_t limit) {
  uint32_t i = 0;
  while (i + 8 <= limit) {
    uint64_t x = br_load64(a + i) ^ br_load64(b + i);
    if (x != 0) return i + (uint32_t)(__builtin_ctzll(x) >> 3);
    i += 8;
  }
  while (i < limit) {
    if (a[i] != b[i]) return i;
    ++i;
  }
  return limit;
}

// Scores of a distance-cache candidate (short code c) and of a ring candidate.
// AdvHasher (H5/H6): BackwardReferenceScoreUsingLastDistance minus BackwardReferencePenaltyUsingLastDistance /
// BackwardReferenceScore, mod.rs:1871-1889, 1151-1154.  H9: mod.rs:685-708 with kDistanceShortCodeCost.
template <bool kH9>
BR_DEV uint32_t br_score_cache(const Lz77Params& P, uint32_t len, uint32_t c) {
  if (kH9) {
    // kDistanceShortCodeCost[c] - (120 * 8 * 8 - 127), one byte per entry
    const uint64_t tab_lo = ((uint64_t)187) | ((uint64_t)32 << 8) | ((uint64_t)10 << 16) | ((uint64_t)0 << 24) | ((uint64_t)34 << 32) |
                            ((uint64_t)34 << 40) | ((uint64_t)31 << 48) | ((uint64_t)31 << 56);
    const uint64_t tab_hi = ((uint64_t)28) | ((uint64_t)28 << 8) | ((uint64_t)22 << 16) | ((uint64_t)22 << 24) | ((uint64_t)12 << 32) |
                            ((uint64_t)12 << 40) | ((uint64_t)2 << 48) | ((uint64_t)2 << 56);
    const uint32_t cost = (120u * 8u * 8u - 127u) + (uint32_t)(((c < 8 ? tab_lo : tab_hi) >> (8 * (c & 7))) & 0xff);
    return (P.literal_byte_score * len + cost) >> 2;
  }
  uint32_t score = P.score_per_byte * len + 30 * 8 * 8 + 15;
  if (c != 0) score -= 39u + ((0x1ca10u >> (c & 0xe)) & 0xe);
  return score;
}
template <bool kH9>
BR_DEV uint32_t br_score_ring(const Lz77Params& P, uint32_t len, uint32_t backward) {
  if (kH9) return (120u * 8u * 8u + P.literal_byte_score * len - 120u * br_log2_floor_nonzero(backward)) >> 2;
  return 30 * 8 * 8 + P.score_per_byte * len - 30 * br_log2_floor_nonzero(backward);
}

// Does replacing the candidate list `a` of position p by `b` (both newest first: a_last[0], a_last[-1], ...) possibly
// change what FindLongestMatch finds there?  A ring entry takes part in the search only through
// FindMatchLengthWithLimitMin4 (static_dict.rs:134-147), which ignores it unless its first four bytes equal those at p;
// entries that fail this test can come and go without any effect (they do not even end the bucket walk, which stops
// on distance alone).  So only the entries in the symmetric difference of the two lists need a look.
BR_DEV bool br_row_change_matters(const uint8_t* text, uint32_t p, const uint32_t* a_last, uint32_t na, const uint32_t* b_last,
                                  uint32_t nb) {
  const uint32_t head = br_load32(text + p);
  uint32_t i = 0, j = 0;
  while (i < na || j < nb) {
    const uint32_t qa = i < na ? *(a_last - i) : 0u, qb = j < nb ? *(b_last - j) : 0u;
    if (i < na && j < nb && qa == qb) {
      ++i;
      ++j;
      continue;
    }
    uint32_t q;
    if (j >= nb || (i < na && qa > qb)) {
      q = qa;
      ++i;
    } else {
      q = qb;
      ++j;
    }
    if (q & kMaskedEntry) return true;  // a masked ring entry ends the bucket walk whatever its text is
    if (br_load32(text + q) == head) return true;
  }
  return false;
}

// ---- command.rs -------------------------------------------------------------------------------
// ComputeDistanceCode, command.rs:48-68
BR_DEV uint32_t br_compute_distance_code(uint32_t distance, uint32_t max_distance, const int32_t* dc) {
  if (distance <= max_distance) {
    const int64_t d = (int64_t)distance;
    const uint64_t offset0 = (uint64_t)(d + 3 - (int64_t)dc[0]);
    const uint64_t offset1 = (uint64_t)(d + 3 - (int64_t)dc[1]);
    if (d == (int64_t)dc[0]) return 0;
    if (d == (int64_t)dc[1]) return 1;
    if (offset0 < 7) return (0x09750468u >> (4 * (uint32_t)offset0)) & 0xf;
    if (offset1 < 7) return (0x0fdb1aceu >> (4 * (uint32_t)offset1)) & 0xf;
    if (d == (int64_t)dc[2]) return 2;
    if (d == (int64_t)dc[3]) return 3;
  }
  return distance + 15;
}
// GetInsertLengthCode / GetCopyLengthCode, command.rs:71-108
BR_DEV uint32_t br_insert_length_code(uint32_t insertlen) {
  if (insertlen < 6) return insertlen;
  if (insertlen < 130) {
    uint32_t nbits = br_log2_floor_nonzero(insertlen - 2) - 1u;
    return (nbits << 1) + ((insertlen - 2) >> nbits) + 2;
  }
  if (insertlen < 2114) return br_log2_floor_nonzero(insertlen - 66) + 10;
  if (insertlen < 6210) return 21;
  if (insertlen < 22594) return 22;
  return 23;
}
BR_DEV uint32_t br_copy_length_code(uint32_t copylen) {
  if (copylen < 10) return copylen - 2;
  if (copylen < 134) {
    uint32_t nbits = br_log2_floor_nonzero(copylen - 6) - 1u;
    return (nbits << 1) + ((copylen - 6) >> nbits) + 4;
  }
  if (copylen < 2118) return br_log2_floor_nonzero(copylen - 70) + 12;
  return 23;
}
// combine_length_codes, command.rs:110-125
BR_DEV uint16_t br_combine_length_codes(uint32_t inscode, uint32_t copycode, bool use_last_distance) {
  uint32_t bits64 = (copycode & 0x7u) | ((inscode & 0x7u) << 3);
  if (use_last_distance && inscode < 8 && copycode < 16) return (uint16_t)(copycode < 8 ? bits64 : (bits64 | 64));
  int sub_offset = 2 * (int)((copycode >> 3) + 3 * (inscode >> 3));
  int offset = (sub_offset << 5) + 0x40 + ((0x520d40 >> sub_offset) & 0xc0);
  return (uint16_t)((uint32_t)offset | bits64);
}
// Command::init, command.rs:273-297 with PrefixEncodeCopyDistance :134-173
BR_DEV Command br_make_command(uint32_t ndirect, uint32_t npostfix, uint32_t insertlen, uint32_t copylen, uint32_t copylen_code,
                               uint32_t distance_code) {
  Command c;
  c.insert_len_ = insertlen;
  int32_t delta = (int32_t)copylen_code - (int32_t)copylen;
  c.copy_len_ = copylen | ((uint32_t)(uint8_t)(int8_t)delta << 25);
  if (distance_code < 16 + ndirect) {
    c.dist_prefix_ = (uint16_t)distance_code;
    c.dist_extra_ = 0;
  } else {
    uint64_t dist = (1ull << (npostfix + 2)) + ((uint64_t)distance_code - 16 - ndirect);
    uint32_t bucket = (63u ^ (uint32_t)__builtin_clzll(dist)) - 1;
    uint64_t postfix_mask = (1u << npostfix) - 1;
    uint64_t postfix = dist & postfix_mask;
    uint64_t prefix = (dist >> bucket) & 1;
    uint64_t offset = (2 + prefix) << bucket;
    uint64_t nbits = bucket - npostfix;
    c.dist_prefix_ = (uint16_t)((nbits << 10) | (16 + ndirect + ((2 * (nbits - 1) + prefix) << npostfix) + postfix));
    c.dist_extra_ = (uint32_t)((dist - offset) >> npostfix);
  }
  c.cmd_prefix_ = br_combine_length_codes(br_insert_length_code(insertlen), br_copy_length_code(copylen_code),
                                          (c.dist_prefix_ & 0x3ff) == 0);
  return c;
}

// The chains only record what a command is made of; the prefix codes are computed when the per-segment slabs are
// gathered (one thread per command instead of one lane of a whole wavefront).
BR_DEV Command br_raw_command(uint32_t insertlen, uint32_t copylen, uint32_t copylen_code, uint32_t distance_code) {
  Command c;
  c.insert_len_ = insertlen;
  const int32_t delta = (int32_t)copylen_code - (int32_t)copylen;
  c.copy_len_ = copylen | ((uint32_t)(uint8_t)(int8_t)delta << 25);
  c.dist_extra_ = distance_code;
  c.cmd_prefix_ = 0;
  c.dist_prefix_ = 0;
  return c;
}
BR_DEV Command br_finish_command(const Command& raw, uint32_t ndirect, uint32_t npostfix) {
  const uint32_t copylen = raw.copy_len_ & 0x01ffffffu;
  const uint32_t m = raw.copy_len_ >> 25;
  const int32_t delta = (int32_t)(int8_t)(uint8_t)(m | ((m & 0x40) << 1));
  return br_make_command(ndirect, npostfix, raw.insert_len_, copylen, (uint32_t)((int32_t)copylen + delta), raw.dist_extra_);
}

// fix-ups applied after the parse reached its fixed point
BR_DEV void br_apply_patch(Command* cmds, const CmdPatch& p) {
  Command c = cmds[p.index];
  if (p.kind == 0) {
    // extend_last_command, encode.rs:385-397: note the 7-bit delta is NOT sign extended there
    c.copy_len_ += p.value;
    const uint32_t copy_code = (c.copy_len_ & 0x01ffffffu) + (c.copy_len_ >> 25);
    c.cmd_prefix_ = br_combine_length_codes(br_insert_length_code(c.insert_len_), br_copy_length_code(copy_code),
                                            (c.dist_prefix_ & 0x3ff) == 0);
  } else if (p.kind == 2) {
    // literals pending at the segment entry belong to its first command
    c.insert_len_ += p.value;
    const uint32_t m = c.copy_len_ >> 25;
    const int32_t delta = (int32_t)(int8_t)(uint8_t)(m | ((m & 0x40) << 1));
    const uint32_t copy_code = (uint32_t)((int32_t)(c.copy_len_ & 0x01ffffffu) + delta);
    c.cmd_prefix_ = br_combine_length_codes(br_insert_length_code(c.insert_len_), br_copy_length_code(copy_code),
                                            (c.dist_prefix_ & 0x3ff) == 0);
  } else {
    // Command::init_insert, command.rs:38-44
    c.insert_len_ = p.value;
    c.copy_len_ = 4u << 25;
    c.dist_extra_ = 0;
    c.dist_prefix_ = (uint16_t)((1u << 10) | 16u);
    c.cmd_prefix_ = br_combine_length_codes(br_insert_length_code(p.value), br_copy_length_code(4), false);
  }
  cmds[p.index] = c;
}

// ---- ring buffer emulation -----------------------------------------------------------------------
// The reference reads the input through a ring buffer of ring_mask+1 bytes that is filled one block
// at a time (encode.rs:709-831).  Everything it reads below pos_end equals the flat text; the byte AT
// pos_end (reachable by the quick-reject test when best_len == max_length) is not written yet: it is
// 0 during the first lap (zero-initialised storage + 7 cleared bytes, encode.rs:823-830) and the
// byte from one lap earlier afterwards.
BR_DEV uint8_t br_unwritten_byte(const Lz77Params& P, const ChainTables& t, uint32_t pos) {
  return pos <= P.ring_mask ? (uint8_t)0 : t.text[pos - (P.ring_mask + 1u)];
}

struct DictState {
  uint32_t lookups, matches;  // absolute counters (entry hint + local)
  uint32_t lookups0, matches0;
  uint32_t mode;              // see SegExit::dict_mode
  int32_t maxdef;
  // bookkeeping of a chain that finds the dictionary switched off: what it WOULD have looked up, and whether any of
  // those lookups would have produced a match.  While none does, its parse is also the parse of a live dictionary,
  // which lets the host accept it in either regime (and keep exact counters).
  uint32_t vlookups, vwould;
  int32_t vmaxdef;
};

// ---- speculative probe of two consecutive positions ---------------------------------------------------
// Phase 1 of AdvHasher::FindLongestMatch (mod.rs:1684-1812) for positions p0 and p0 + 1 at once: every
// candidate of both positions (dist-cache entries, bucket ring entries, the two static-dictionary probes)
// gets its own lane, which locates it and measures the common prefix.  All global loads of the two searches
// are issued together, so the pair costs one dependent-load chain (info -> sorted -> text) instead of two.
// The greedy parse consumes search(p0) and then, in almost every case, search(p0 + 1) (as the lazy probe
// after a match, or as the next position after a miss), with an unchanged distance cache.
struct ProbeMeta {
  uint32_t pos;       // p0; 0xffffffff = nothing probed
  uint32_t version;   // dist-cache version the probe was computed with
  uint32_t g[2], nbucket[2];
  uint32_t win_base;  // first position of the rank-record window held in ChainScratch::win
  // kRows on the device: the candidate of THIS lane (lanes 0..31 serve p0, lanes 32..63 p0 + 1; see br_probe_pair_rows)
  uint32_t r_prev, r_len;
  uint32_t no_dict;   // the static dictionary is known to be switched off for good: no probes, no bookkeeping
  uint32_t log_on = 0;  // write ChainTables::search_log
  // live chains: key and ring counter of the two probed positions as the probe found them (the position is filed when its
  // search is folded, br_search), and a window of 64 hash keys, one per lane
  uint32_t live_key[2], live_n[2];
  uint32_t kwin_base, kwin;
#if defined(BR_CHAIN_PROFILE)
  unsigned long long t_probe, t_fold, n_probe, n_fold, t_setup, t_refill, n_refill;
#endif
};
#if defined(BR_CHAIN_PROFILE)
#define BR_TICK() ((unsigned long long)__builtin_amdgcn_s_memtime())
extern __device__ unsigned long long g_chain_prof[16];
#endif

#if BR_SCALAR
BR_DEV uint32_t br_match_len_wide(const uint8_t* a, const uint8_t* b, uint32_t limit, const uint32_t* = nullptr, uint32_t = 0, uint32_t = 0) {
  return br_match_len(a, b, limit);
}
#else
// Common prefix of a and b, at most `limit`: the first 32 bytes of both sides are fetched in one go (four independent
// 16-byte loads, one memory round trip; both buffers are padded so reading past `limit` is harmless), only longer
// matches fall back to the 8-byte loop.
// run_end / a_pos / b_pos: optional run table and the text positions of a and b (see ChainTables::run_end)
BR_DEV uint32_t br_match_len_wide(const uint8_t* a, const uint8_t* b, uint32_t limit, const uint32_t* run_end = nullptr, uint32_t a_pos = 0,
                                  uint32_t b_pos = 0) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2), aligned(1)));
  const u64x2 a0 = *(const u64x2*)a, a1 = *(const u64x2*)(a + 16);
  const u64x2 b0 = *(const u64x2*)b, b1 = *(const u64x2*)(b + 16);
  const unsigned long long x0 = a0.x ^ b0.x, x1 = a0.y ^ b0.y, x2 = a1.x ^ b1.x, x3 = a1.y ^ b1.y;
  uint32_t n;
  if (x0 != 0) n = (uint32_t)(__builtin_ctzll(x0) >> 3);
  else if (x1 != 0) n = 8 + (uint32_t)(__builtin_ctzll(x1) >> 3);
  else if (x2 != 0) n = 16 + (uint32_t)(__builtin_ctzll(x2) >> 3);
  else if (x3 != 0) n = 24 + (uint32_t)(__builtin_ctzll(x3) >> 3);
  else n = 32;
  if (n >= 32 && limit > 32) {
    // long matches (runs, repeats): 32 bytes per trip while they last, then the 8-byte loop for the remainder
    uint32_t i = 32;
    if (run_end != nullptr) {
      // both sides start with 32 equal bytes; if those are one and the same byte, both sit in runs of it: the shorter
      // run ends the match, and only when they end together does the comparison go on behind them
      const unsigned long long pat = (a0.x & 0xffull) * 0x0101010101010101ull;
      if (a0.x == pat && a0.y == pat && a1.x == pat && a1.y == pat) {
        const uint32_t ra = run_end[a_pos] - a_pos, rb = run_end[b_pos] - b_pos;
        if (ra != rb) {
          const uint32_t r = ra < rb ? ra : rb;
          return r < limit ? r : limit;
        }
        if (ra >= limit) return limit;
        i = ra;
      }
    }
    while (i + 32 <= limit) {
      const u64x2 c0 = *(const u64x2*)(a + i), c1 = *(const u64x2*)(a + i + 16);
      const u64x2 d0 = *(const u64x2*)(b + i), d1 = *(const u64x2*)(b + i + 16);
      if (((c0.x ^ d0.x) | (c0.y ^ d0.y) | (c1.x ^ d1.x) | (c1.y ^ d1.y)) != 0) break;
      i += 32;
    }
    n = i + br_match_len(a + i, b + i, limit - i);
  }
  return n < limit ? n : limit;
}
#endif

#if !BR_SCALAR
// kRows, device: fixed lane layout, results stay in registers.  Half w = lane >> 5 probes position p0 + w; inside a half
// lane c = lane & 31 holds: c < ndist the distance-cache candidate c, ndist <= c < ndist + 16 ring entry c - ndist of the
// row, c = ndist + 16 / + 17 the two static-dictionary probes.  One memory round trip per probe: the rows and the
// dictionary hash items of the next kRowWindow positions sit in LDS (refilled with one coalesced load per window).
static constexpr uint32_t kRowDictLane = 16;  // offset of the dictionary lanes behind the cache lanes
// Common prefix of the 16 bytes at a and at b, 0..16, without a branch: the position of the lowest differing bit over the four
// dword differences (v_ffbl gives 0xffffffff for "equal", which survives the OR with the dword's bit offset and loses every minimum).
// The compiler turned the cascade "first differing qword" of br_match_len_wide into four nested exec-mask regions per probe
// (round 6: the parse kernel is bound by instruction issue, SALU first), and most candidates differ within a few bytes.
typedef uint32_t br_u32x4 __attribute__((ext_vector_type(4), aligned(1)));
// bit number of the lowest set bit, 0xffffffff for 0: v_ffbl_b32 as the hardware defines it (__ffs / __builtin_ctz cost a compare and a
// select more per word to make the zero case defined)
BR_DEV uint32_t br_ffbl(uint32_t x) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
BR_DEV uint32_t br_common16v(const br_u32x4 a0, const br_u32x4 b0) {
  const uint32_t t0 = br_ffbl(a0.x ^ b0.x), t1 = br_ffbl(a0.y ^ b0.y) | 32u, t2 = br_ffbl(a0.z ^ b0.z) | 64u, t3 = br_ffbl(a0.w ^ b0.w) | 96u;
  const uint32_t bits = min(min(min(t0, t1), t2), t3);  // (v_min3_u32, v_min_u32)
  return min(bits >> 3, 16u);
}
BR_DEV uint32_t br_common16(const uint8_t* a, const uint8_t* b) { return br_common16v(*(const br_u32x4*)a, *(const br_u32x4*)b); }

template <bool kH9>
BR_DEV void br_probe_pair_rows(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, true>& s, ProbeMeta& m, uint32_t p0,
                               const int32_t* cache, uint32_t cache_version, uint32_t pos_end) {
  const uint32_t ndist = P.ndist;
  const bool use_dict = P.use_dictionary && !m.no_dict;
#if defined(BR_CHAIN_PROFILE)
  const unsigned long long tp0 = BR_TICK();
#endif
  if (p0 < m.win_base || p0 + 1 >= m.win_base + kRowWindow) {
    BR_WAVE_SYNC();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // dictionary hash items of the window positions (SearchInStaticDictionary, mod.rs:1942-1988): lane = 2 * position + probe
    // (first4 holds the items themselves where the per-position array exists: one memory round trip instead of two)
    const bool items_ready = t.dict_items != nullptr;
    uint32_t first4[kRowWindow * 2 / 64];
#pragma unroll
    for (uint32_t j = 0; j < kRowWindow * 2 / 64; ++j) {
      const uint32_t q = min(p0 + j * 32 + ((uint32_t)BR_LANE >> 1), P.total_bytes);  // (text and items are padded by 64 bytes)
      first4[j] = !use_dict ? 0u : (items_ready ? t.dict_items[q] : br_load32(t.text + q));
    }
    u32x4 v[kRowWindow * (kRowEntries / 4) / 64];
#pragma unroll
    for (uint32_t j = 0; j < kRowWindow * (kRowEntries / 4) / 64; ++j) {
      const uint32_t i = j * 64 + BR_LANE;
      const uint32_t q = p0 + i / (kRowEntries / 4);
      const u32x4 none = {kRowEnd, kRowEnd, kRowEnd, kRowEnd};
      v[j] = none;
      if (q < P.total_bytes) v[j] = __builtin_nontemporal_load((const u32x4*)t.rows + (size_t)q * (kRowEntries / 4) + (i % (kRowEntries / 4)));
    }
    if (use_dict) {
#pragma unroll
      for (uint32_t j = 0; j < kRowWindow * 2 / 64; ++j) {
        const uint32_t item = items_ready ? ((first4[j] >> (16u * (BR_LANE & 1))) & 0xffffu)
                                          : (uint32_t)t.dict_hash[(((first4[j] * 0x1e35a7bdu) >> (32 - 14)) << 1) + (BR_LANE & 1)];
        s.dictwin[j * 64 + BR_LANE] = (uint16_t)item;
        // where the word starts: kBrotliDictionaryOffsetsByLength[len] + len * index (one dependent LDS read per window
        // instead of one in front of every probe's candidate fetch)
        const uint32_t wlen = item & 0x1f;
        s.dictsrc[j * 64 + BR_LANE] = s.dict_off[wlen] + wlen * (item >> 5);
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < kRowWindow * (kRowEntries / 4) / 64; ++j) ((u32x4*)s.win)[j * 64 + BR_LANE] = v[j];
    m.win_base = p0;
    BR_WAVE_SYNC();
#if defined(BR_CHAIN_PROFILE)
    m.t_refill += BR_TICK() - tp0;
    m.n_refill++;
#endif
  }
  m.pos = p0;
  m.version = cache_version;
  m.nbucket[0] = m.nbucket[1] = kRowEntries;  // (the generic fold walks the whole row; kRowEnd entries end the walk)
  // (the lane number through an empty asm: what hangs on it -- the role masks of the lanes -- is then computed where it is used, one
  // compare each, instead of being hoisted out of the parse loop into scalar register pairs that spill)
  uint32_t lane_id = (uint32_t)BR_LANE;
  asm volatile("" : "+v"(lane_id));
  const uint32_t w = lane_id >> 5, c = lane_id & 31u;
  const uint32_t cur = p0 + w;
  const uint32_t max_length = pos_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  const uint32_t rel = cur - m.win_base;
  const uint32_t ci = c - ndist;  // ring entry number (wraps around for the cache lanes)
  const bool is_cache = c < ndist;
  const bool is_ring = ci < kRowEntries;
  const bool is_dict = use_dict && (ci - kRowDictLane) < 2u;
  // Straight-line code: every lane reads all three kinds of source (LDS, in bounds whatever the lane) and selects; the lanes
  // behind the dictionary lanes and those without a candidate compare the text at `cur` with itself and drop the result.
  const int32_t cb = cache[c & 15u];
  const uint32_t ring_e = s.win[rel * kRowEntries + (ci & (kRowEntries - 1u))];
  const uint32_t di = rel * 2u + (ci & 1u);
  const uint32_t item = s.dictwin[di];
  const uint32_t doff = s.dictsrc[di];
  const uint32_t cache_prev = (cb > 0 && (uint32_t)cb <= max_backward) ? cur - (uint32_t)cb : 0xffffffffu;
  const uint32_t wlen = item & 0x1fu;
  const bool dict_word = is_dict && item != 0 && wlen <= max_length;
  const uint32_t prev = is_cache ? cache_prev : (is_ring ? ring_e : (is_dict ? item : 0xffffffffu));  // kRowEnd == "no candidate"
  const bool valid = dict_word || (!is_dict && prev != 0xffffffffu);
  const uint32_t limit = dict_word ? wlen : max_length;
  const uint8_t* base = dict_word ? t.dict_data : t.text;
  const uint8_t* src = base + (dict_word ? doff : (valid ? prev : cur));
  const uint8_t* cur_data = t.text + cur;
#if defined(BR_CHAIN_PROFILE)
  m.t_setup += BR_TICK() - tp0;
#endif
  uint32_t n = br_common16(src, cur_data);
  if (valid && n >= 16u && limit > 16u) {
    // the minority of candidates that agree in 16 bytes: the next 16, and behind those (rare: runs, long repeats) the 32-byte
    // strides and the run table of br_match_len_wide
    const uint32_t n2 = br_common16(src + 16, cur_data + 16);
    n = 16u + n2;
    if (n2 >= 16u && limit > 32u) n = br_match_len_wide(src, cur_data, limit, dict_word ? nullptr : t.run_end, prev, cur);
  }
  m.r_len = valid ? (n < limit ? n : limit) : 0u;
  m.r_prev = prev;
}
#endif


#if !BR_SCALAR
// Live chains, ring depth <= 16: the lane layout of br_probe_pair_rows, the ring entries read from the chain's own table
// (lz77_live.h).  Two memory round trips per probe: ring counters + bucket rows of both positions, then the candidate
// text.  The hash keys come from a window of 64 keys held one per lane.  Position p0 is filed when its search is folded
// (br_search), i.e. before p0 + 1 is searched: when both have the same key, the second half of the wave sees p0 as its
// newest entry.
template <bool kH9>
BR_DEV void br_probe_pair_live16(const Lz77Params& P, const ChainTables& t, const LiveRing& lr, ProbeMeta& m, uint32_t p0, const int32_t* cache,
                                 uint32_t cache_version, uint32_t pos_end) {
  const uint32_t ndist = P.ndist;
  const bool use_dict = P.use_dictionary && !m.no_dict;
  const uint32_t lane = (uint32_t)BR_LANE;
  if (p0 < m.kwin_base || p0 + 1 >= m.kwin_base + 64u) {
    m.kwin_base = p0;
    m.kwin = (uint32_t)lr.keys[min(p0 + lane, P.total_bytes)];  // (the key array is padded)
  }
  const uint32_t k0 = BR_READLANE(m.kwin, p0 - m.kwin_base);
  // (a position without a key -- fewer than 4 bytes in front of the end of the text -- is never searched)
  const bool second = p0 + 1 + P.htl <= P.total_bytes;
  const uint32_t k1 = second ? BR_READLANE(m.kwin, p0 + 1 - m.kwin_base) : k0;
  const bool same = second && k1 == k0;
  const uint32_t w = lane >> 5, c = lane & 31u;
  const uint32_t key = w ? k1 : k0;
  const uint32_t depth = 1u << lr.bits;
  // ring counter and the 16 slots of the ring are requested together (one round trip); which slot is the newest entry
  // comes out of the counter afterwards
  const uint32_t raw_n = (uint32_t)BR_LIVE_LD16(lr.num + key);
  const uint32_t raw_e = (c >= ndist && c < ndist + kRowEntries) ? (uint32_t)BR_LIVE_LD32(lr.buckets + (((size_t)key << lr.bits) | (c - ndist))) : 0u;
  const uint32_t n0 = BR_READLANE(raw_n, 0), n1 = (BR_READLANE(raw_n, 32) + (same ? 1u : 0u)) & 0xffffu;
  const uint32_t n = w ? n1 : n0;
  const uint32_t visible = !(w && !second) ? (n < depth ? n : depth) : 0u;
  const uint32_t cur = p0 + w;
  const uint32_t max_length = pos_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  const uint8_t* cur_data = t.text + cur;
  const bool is_cache = c < ndist;
  const uint32_t i = c - ndist;  // ring entry number, newest first
  const bool is_ring = !is_cache && i < visible;
  uint32_t e = kLiveBreak;
  {
    const uint32_t from = (lane & 32u) + ndist + ((n - 1u - i) & (depth - 1u));  // the lane that loaded slot (n - 1 - i) mod depth
    const uint32_t moved = (uint32_t)__shfl((int)raw_e, (int)(from & 63u), 64);
    if (is_ring) e = (w && same && i == 0) ? p0 : moved;
  }
  // the bucket walk ends with the first entry that is masked or out of reach (mod.rs:1763-1775)
  const bool brk = is_ring && (e >= kLiveBreak || cur - e > max_backward);
  const unsigned long long brk_mask = __ballot(brk);
  const uint32_t brk_half = (uint32_t)(brk_mask >> (32u * w));
  const uint32_t first_brk = brk_half ? (uint32_t)__ffs((int)brk_half) - 1u : 32u;  // lane number within the half
  m.live_key[0] = k0;
  m.live_key[1] = k1;
  m.live_n[0] = n0;
  m.live_n[1] = n1;
  m.pos = p0;
  m.version = cache_version;
  m.nbucket[0] = m.nbucket[1] = kRowEntries;
  const bool is_dict = use_dict && c >= ndist + kRowDictLane && c < ndist + kRowDictLane + 2;
  uint32_t prev = 0xffffffffu, limit = max_length;
  const uint8_t* src = nullptr;
  if (is_cache) {
    const int64_t b = (int64_t)cache[c];
    if (b > 0 && b <= (int64_t)max_backward) prev = cur - (uint32_t)b;
  } else if (is_ring) {
    if (c < first_brk) prev = e;
  } else if (is_dict) {
    const uint32_t item = t.dict_hash[(((br_load32(cur_data) * 0x1e35a7bdu) >> (32 - 14)) << 1) + (c - ndist - kRowDictLane)];
    prev = item;
    if (item != 0) {
      const uint32_t wlen = item & 0x1f;
      if (wlen <= max_length) {
        src = t.dict_data + t.dict_offsets_by_length[wlen] + wlen * (item >> 5);
        limit = wlen;
      }
    }
  }
  if (!is_dict && prev != 0xffffffffu) src = t.text + prev;
  m.r_len = src ? br_match_len_wide(src, cur_data, limit, is_dict ? nullptr : t.run_end, prev, cur) : 0u;
  m.r_prev = prev;
}
#endif

template <bool kH9, bool kRows, bool kLive = false, bool kDeep = false>
BR_DEV void br_probe_pair(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, kRows, kDeep>& s, ProbeMeta& m, uint32_t p0,
                          const int32_t* cache, uint32_t cache_version, uint32_t pos_end, const LiveRing* live = nullptr) {
#if !BR_SCALAR
  if constexpr (kRows && kLive) {
    br_probe_pair_live16<kH9>(P, t, *live, m, p0, cache, cache_version, pos_end);
    return;
  } else if constexpr (kRows) {
    br_probe_pair_rows<kH9>(P, t, s, m, p0, cache, cache_version, pos_end);
    return;
  }
#endif
  const uint32_t ndist = P.ndist;
  const uint32_t block_size = 1u << P.block_bits;
  const uint32_t ndict = (P.use_dictionary && !m.no_dict) ? 2u : 0u;
  uint32_t n[2];
#if defined(BR_CHAIN_PROFILE)
  const unsigned long long tp0 = BR_TICK();
#endif
  // live chains: ring entry i (newest first) of probe slot w, see br_probe_pair_live16
  bool live_same = false;
  auto live_entry = [&](uint32_t w, uint32_t i) -> uint32_t {
    if (w && live_same && i == 0) return p0;
    return BR_LIVE_LD32(live->buckets + (((size_t)m.live_key[w] << live->bits) | ((m.live_n[w] - 1u - i) & (block_size - 1u))));
  };
  if constexpr (kLive) {
    const bool second = p0 + 1 + P.htl <= P.total_bytes;
    const uint32_t k0 = BR_UNIFORM(live->keys[p0]);
    const uint32_t k1 = second ? BR_UNIFORM(live->keys[p0 + 1]) : k0;
    live_same = second && k1 == k0;
    const uint32_t n0 = BR_UNIFORM(BR_LIVE_LD16(live->num + k0));
    const uint32_t n1 = second ? ((live_same ? n0 + 1u : BR_UNIFORM(BR_LIVE_LD16(live->num + k1))) & 0xffffu) : 0u;
    m.live_key[0] = k0;
    m.live_key[1] = k1;
    m.live_n[0] = n0;
    m.live_n[1] = n1;
    m.g[0] = m.g[1] = 0;
    m.nbucket[0] = n0 < block_size ? n0 : block_size;
    m.nbucket[1] = n1 < block_size ? n1 : block_size;
  } else if (kRows) {
#if BR_SCALAR
    for (int w = 0; w < 2; ++w) {
      uint32_t nb = 0;
      if (p0 + w < P.total_bytes)
        while (nb < kRowEntries && t.rows[(size_t)(p0 + w) * kRowEntries + nb] != kRowEnd) ++nb;
      m.g[w] = 0;
      m.nbucket[w] = nb;
    }
#else
    // candidate rows of the positions from p0 on: contiguous, 64 bytes per position
    if (p0 < m.win_base || p0 + 1 >= m.win_base + kRowWindow) {
      BR_SYNC();
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      for (uint32_t i = BR_LANE; i < kRowWindow * (kRowEntries / 4); i += BR_NLANES) {
        const uint32_t q = p0 + i / (kRowEntries / 4);
        u32x4 v = {kRowEnd, kRowEnd, kRowEnd, kRowEnd};
        if (q < P.total_bytes) v = __builtin_nontemporal_load((const u32x4*)t.rows + (size_t)q * (kRowEntries / 4) + (i % (kRowEntries / 4)));
        ((u32x4*)s.win)[i] = v;
      }
      m.win_base = p0;
      BR_SYNC();
#if defined(BR_CHAIN_PROFILE)
      m.t_refill += BR_TICK() - tp0;
      m.n_refill++;
#endif
    }
    {
      const uint32_t l = (uint32_t)BR_LANE;
      const uint32_t e = l < 2 * kRowEntries ? s.win[(p0 - m.win_base + l / kRowEntries) * kRowEntries + (l % kRowEntries)] : kRowEnd;
      const unsigned long long valid = __ballot(e != kRowEnd);
      // (entries are packed from the front: the number of valid ones is where the row ends)
      m.g[0] = m.g[1] = 0;
      m.nbucket[0] = (uint32_t)__popcll(valid & 0xffffull);
      m.nbucket[1] = (uint32_t)__popcll((valid >> kRowEntries) & 0xffffull);
    }
#endif
  } else {
    // rank records of the positions around p0: one coalesced load serves the next ~60 positions
    if (p0 < m.win_base || p0 + 1 >= m.win_base + kInfoWindow) {
      BR_SYNC();
      for (uint32_t i = BR_LANE; i < kInfoWindow; i += BR_NLANES) {
        const uint32_t q = p0 + i;
        const bool ok = q < P.total_bytes;
        s.win[2 * i] = ok ? t.info[2 * (size_t)q] : 0u;
        s.win[2 * i + 1] = ok ? t.info[2 * (size_t)q + 1] : 0u;
      }
      m.win_base = p0;
      BR_SYNC();
    }
    for (int w = 0; w < 2; ++w) {
      const uint32_t g = s.win[2 * (p0 + w - m.win_base)];
      const uint32_t num_copy = s.win[2 * (p0 + w - m.win_base) + 1] & 0xffffu;  // num[key] is u16 and wraps (mod.rs:1752-1760)
      m.g[w] = g;
      m.nbucket[w] = num_copy < block_size ? num_copy : block_size;
    }
  }
  for (int w = 0; w < 2; ++w) n[w] = ndist + m.nbucket[w] + ndict;
#if defined(BR_CHAIN_PROFILE)
  m.t_setup += BR_TICK() - tp0;
#endif
  m.pos = p0;
  m.version = cache_version;
  const uint32_t total = n[0] + n[1];
#if !BR_SCALAR
  if constexpr (!kRows) {
    // Deep rings (up to 2 x 274 candidates at quality 9) in three steps instead of one dependent load chain per 64
    // candidates: (1) all ring entries and their tags are requested at once, (2) every candidate gets its `prev`, and
    // those whose text must be looked at -- valid cache distances, ring entries in reach whose tag equals that of the
    // searched position, dictionary items -- are compacted into a list, (3) the list is worked off 64 entries at a time
    // (typically two trips).  Same results as the loop below, slot for slot.
    constexpr uint32_t kMaxTrips = (2u * (uint32_t)(ChainScratchT<kH9, kRows, kDeep>::kMaxCandidates + 2) + 63u) / 64u;
    const uint32_t lane = (uint32_t)BR_LANE;
    const uint32_t tag_of[2] = {br_tag16(br_load32(t.text + p0)), br_tag16(br_load32(t.text + p0 + 1))};
    uint32_t ring_q[kMaxTrips], ring_tag[kMaxTrips];
#pragma unroll
    for (uint32_t k = 0; k < kMaxTrips; ++k) {
      const uint32_t slot = k * 64u + lane;
      ring_q[k] = 0;
      ring_tag[k] = 0;
      if (slot < total) {
        const uint32_t w = slot < n[0] ? 0u : 1u;
        const uint32_t c = slot - (w ? n[0] : 0u);
        if (c >= ndist && c < ndist + m.nbucket[w]) {
          if constexpr (kLive) {
            ring_q[k] = live_entry(w, c - ndist);
            ring_tag[k] = tag_of[w];
          } else {
            const uint32_t at = m.g[w] - 1 - (c - ndist);
            ring_q[k] = t.sorted[at];
            ring_tag[k] = t.sorted_tag != nullptr ? (uint32_t)t.sorted_tag[at] : tag_of[w];
          }
        }
      }
    }
    uint32_t listed = 0;
#pragma unroll
    for (uint32_t k = 0; k < kMaxTrips; ++k) {
      const uint32_t slot = k * 64u + lane;
      bool fetch = false;
      if (slot < total) {
        const uint32_t w = slot < n[0] ? 0u : 1u;
        const uint32_t c = slot - (w ? n[0] : 0u);
        const uint32_t cur = p0 + w;
        const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
        uint32_t prev = 0xffffffffu;
        if (c < ndist) {
          const int64_t b = (int64_t)cache[c];
          if (b > 0 && b <= (int64_t)max_backward) prev = cur - (uint32_t)b;
          fetch = prev != 0xffffffffu;
        } else if (c < ndist + m.nbucket[w]) {
          if (cur - ring_q[k] <= max_backward && !(kLive && ring_q[k] >= kLiveBreak)) prev = ring_q[k];  // else: marks the point where the bucket walk breaks
          fetch = prev != 0xffffffffu && ring_tag[k] == tag_of[w];
        } else {
          // static dictionary probe (SearchInStaticDictionary, mod.rs:1942-1988): the item is fetched in step 3
          prev = 0;
          fetch = true;
        }
        s.cand_prev[w][c] = prev;
        s.cand_len[w][c] = 0;
      }
      const unsigned long long mask = __ballot(fetch);
      if (fetch) s.fetch_list[listed + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)slot;
      listed += (uint32_t)__popcll(mask);
    }
    BR_SYNC();
    for (uint32_t first = 0; first < listed; first += 64) {
      const uint32_t i = first + lane;
      if (i < listed) {
        const uint32_t slot = s.fetch_list[i];
        const uint32_t w = slot < n[0] ? 0u : 1u;
        const uint32_t c = slot - (w ? n[0] : 0u);
        const uint32_t cur = p0 + w;
        const uint32_t max_length = pos_end - cur;
        const uint8_t* cur_data = t.text + cur;
        const bool is_dict = c >= ndist + m.nbucket[w];
        uint32_t prev = s.cand_prev[w][c], limit = max_length;
        const uint8_t* src = nullptr;
        if (is_dict) {
          const uint32_t key = (((br_load32(cur_data) * 0x1e35a7bdu) >> (32 - 14)) << 1) + (c - ndist - m.nbucket[w]);
          const uint32_t item = t.dict_hash[key];
          prev = item;
          s.cand_prev[w][c] = item;
          if (item != 0) {
            const uint32_t wlen = item & 0x1f;
            if (wlen <= max_length) {
              src = t.dict_data + t.dict_offsets_by_length[wlen] + wlen * (item >> 5);
              limit = wlen;
            }
          }
        } else {
          src = t.text + prev;
        }
        if (src) s.cand_len[w][c] = br_match_len_wide(src, cur_data, limit, is_dict ? nullptr : t.run_end, prev, cur);
      }
    }
    BR_SYNC();
    return;
  }
#endif
  for (uint32_t slot = BR_LANE; slot < total; slot += BR_NLANES) {
    const uint32_t w = slot < n[0] ? 0u : 1u;
    const uint32_t c = slot - (w ? n[0] : 0u);
    const uint32_t cur = p0 + w;
    const uint32_t max_length = pos_end - cur;
    const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
    const uint8_t* cur_data = t.text + cur;
    const bool is_cache = c < ndist;
    const bool is_bucket = !is_cache && c < ndist + m.nbucket[w];
    const bool is_dict = !is_cache && !is_bucket;
    // step 1 (one memory round trip for every kind of candidate): where does the candidate live?
    uint32_t q = 0, item = 0;
    bool other_tag = false;
    if (is_bucket) {
      if (kLive) {
        q = live_entry(w, c - ndist);
      } else if (kRows) {
#if BR_SCALAR
        q = t.rows[(size_t)cur * kRowEntries + (c - ndist)];
#else
        q = s.win[(cur - m.win_base) * kRowEntries + (c - ndist)];
#endif
      } else {
        q = t.sorted[m.g[w] - 1 - (c - ndist)];
        if (t.sorted_tag != nullptr) other_tag = t.sorted_tag[m.g[w] - 1 - (c - ndist)] != br_tag16(br_load32(cur_data));
      }
    }
    if (is_dict) {
      // static dictionary probe i (SearchInStaticDictionary, mod.rs:1942-1988)
      const uint32_t key = (((br_load32(cur_data) * 0x1e35a7bdu) >> (32 - 14)) << 1) + (c - ndist - m.nbucket[w]);
      item = t.dict_hash[key];
    }
    uint32_t prev = 0xffffffffu, limit = max_length;
    const uint8_t* src = nullptr;
    if (is_cache) {
      const int64_t b = (int64_t)cache[c];
      if (b > 0 && b <= (int64_t)max_backward) prev = cur - (uint32_t)b;
    } else if (is_bucket) {
      if (cur - q <= max_backward && !(kLive && q >= kLiveBreak)) prev = q;  // else: marks the point where the bucket walk breaks
    } else {
      prev = item;
      if (item != 0) {
        const uint32_t wlen = item & 0x1f;
        if (wlen <= max_length) {
          src = t.dict_data + t.dict_offsets_by_length[wlen] + wlen * (item >> 5);
          limit = wlen;
        }
      }
    }
    // (a ring entry with another tag starts with other bytes: a match of length 0 as far as the search is concerned)
    if (!is_dict && prev != 0xffffffffu && !other_tag) src = t.text + prev;
    // step 2 (second round trip): measure the common prefix
    const uint32_t len = src ? br_match_len_wide(src, cur_data, limit, is_dict ? nullptr : t.run_end, prev, cur) : 0u;
    s.cand_prev[w][c] = prev;
    s.cand_len[w][c] = len;
  }
  BR_SYNC();
}

// kBrotliDictionarySizeBitsByLength (RFC 7932 appendix A; lengths 4..24), four bits per length: in registers instead of a
// table in memory
BR_DEV uint32_t br_dict_size_bits(uint32_t len) {
  const uint64_t lo = 0x899aaaaabbaa0000ull;   // lengths 0..15: 0,0,0,0,10,10,11,11,10,10,10,10,10,9,9,8
  const uint64_t hi = 0x556677877ull;         // lengths 16..24: 7,7,8,7,7,6,6,5,5
  return (uint32_t)(((len < 16 ? lo : hi) >> (4u * (len & 15u))) & 0xfu);
}

// The static dictionary stage of FindLongestMatch: SearchInStaticDictionary + TestStaticDictionaryItem, mod.rs:1891-1988
// (shallow = false), on the two probed hash items; probed(i, &item, &matchlen) hands them over (matchlen: the common
// prefix of the dictionary word and the text, 0 if the word is longer than max_length).
template <typename Probed>
BR_DEV void br_dictionary_stage(const Lz77Params& P, const ChainTables& t, DictState& ds, bool no_dict, uint32_t max_length, uint32_t max_backward,
                                SearchResult& out, Probed probed) {
  if (out.found || !P.use_dictionary) return;
  const bool dead = ds.matches < (ds.lookups >> 7);
  const uint32_t seen = dead ? 2u : 1u;
  ds.mode = (ds.mode == 0 || ds.mode == seen) ? seen : 3u;
  if (no_dict) {
    // switched off for good under the exact counters of this round: no probes, no virtual bookkeeping.  Should a later
    // pass of the resolver find the dictionary still alive here (something changed upstream), this parse says
    // nothing about what a live dictionary would have done: mode 4 = "ran blind", valid only while the dictionary is off.
    ds.mode = 4;
    return;
  }
  if (dead && ds.vwould) return;
  if (dead) {
    if ((int32_t)ds.vlookups > ds.vmaxdef) ds.vmaxdef = (int32_t)ds.vlookups;
  } else {
    const int32_t def = (int32_t)(ds.lookups - ds.lookups0) - 128 * (int32_t)(ds.matches - ds.matches0);
    if (def > ds.maxdef) ds.maxdef = def;
  }
  uint32_t threshold = out.score;
  for (uint32_t i = 0; i < 2; ++i) {
    uint32_t item, matchlen;
    probed(i, &item, &matchlen);
    if (dead) ds.vlookups++; else ds.lookups++;
    if (item == 0) continue;
    const uint32_t len = item & 0x1f;
    const uint32_t dist = item >> 5;
    if (len > max_length) continue;
    if (matchlen + 10 <= len || matchlen == 0) continue;
    const uint32_t cut = len - matchlen;
    const uint32_t transform_id = (cut << 2) + (uint32_t)((0x071b520ada2d3200ull >> (cut * 6)) & 0x3f);
    const uint32_t backward = max_backward + dist + 1 + (transform_id << br_dict_size_bits(len));
    if (backward > P.dist_max_distance) continue;
    const uint32_t score = 30 * 8 * 8 + P.score_per_byte * matchlen - 30 * br_log2_floor_nonzero(backward);
    if (score < threshold) continue;
    threshold = score;
    if (dead) {
      ds.vwould = 1;
      continue;
    }
    out.len = matchlen;
    out.len_x_code = len ^ matchlen;
    out.distance = backward;
    out.score = score;
    ds.matches++;
    out.found = true;
  }
}

// Phase 2: fold the candidates of probe slot w in the reference's order, then the static dictionary stage.
template <bool kH9, bool kRows, bool kDeep = false>
BR_DEV SearchResult br_fold_probe(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, kRows, kDeep>& s, const ProbeMeta& m, uint32_t w,
                                  DictState& ds, uint32_t blk_end, SearchResult* before_dictionary = nullptr, bool candidates_only = false) {
  const uint32_t cur = m.pos + w;
  const uint32_t max_length = blk_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  const uint32_t ndist = P.ndist;
  const uint32_t ncand = ndist + m.nbucket[w];
  SearchResult out;
  out.len = 0;
  out.len_x_code = 0;
  out.distance = 0;
  out.score = kMinScore;
  out.found = false;
  out.stored = true;
#if BR_SCALAR
  uint32_t best_len = 0;
  uint32_t best_score = kMinScore;
  const uint32_t brk = P.dict_break;
  for (uint32_t c = 0; c <= ncand; ++c) {
    if (kH9 && c == ndist && (cur & P.ring_mask) + best_len > P.ring_mask) {
      out.stored = false;  // H9: no bucket stage, no insertion
      break;
    }
    if (c == ncand) break;
    const uint32_t prev = s.cand_prev[w][c];
    if (prev == 0xffffffffu) {
      if (c < ndist) continue;
      break;  // backward > max_backward: older ring entries are farther still
    }
    const uint32_t unbroken = s.cand_len[w][c];
    // quick reject (mod.rs:1713-1718 / 1765-1773)
    if ((cur & P.ring_mask) + best_len > P.ring_mask || (prev & P.ring_mask) + best_len > P.ring_mask) continue;
    bool same;
    if (best_len < unbroken) {
      same = true;
    } else if (best_len == unbroken && unbroken < max_length) {
      same = false;
    } else {
      const uint8_t cb = (cur + best_len < blk_end) ? t.text[cur + best_len] : br_unwritten_byte(P, t, cur + best_len);
      same = cb == t.text[prev + best_len];
    }
    if (!same) continue;
    // fix_unbroken_len, mod.rs:42-54
    uint32_t len = unbroken;
    // (on the ring-buffer index of the candidate, like the reference: the rule comes back with every revolution of the ring)
    if (brk != 0 && (prev & P.ring_mask) < brk && (prev & P.ring_mask) + unbroken > brk) len = brk - (prev & P.ring_mask);
    if (c < ndist) {
      if (unbroken >= 3 || (unbroken == 2 && c < 2)) {
        const uint32_t score = br_score_cache<kH9>(P, len, c);
        if (best_score < score) {
          best_score = score;
          best_len = len;
          out.len = len;
          out.distance = cur - prev;
          out.score = score;
          out.found = true;
        }
      }
    } else if (unbroken >= 4) {  // FindMatchLengthWithLimitMin4 != 0
      const uint32_t backward = cur - prev;
      const uint32_t score = br_score_ring<kH9>(P, len, backward);
      if (best_score < score) {
        best_score = score;
        best_len = len;
        out.len = len;
        out.distance = backward;
        out.score = score;
        out.found = true;
      }
    }
  }
#else
  // Wave-parallel form of the same fold.  Every lane scores its own candidate; the reference's sequential rule
  // ("a later candidate replaces the best one only if it passes the quick-reject test at best_len and scores
  // strictly higher") is replayed by repeatedly taking the FIRST lane that beats the current best.  A candidate
  // whose match is not longer than best_len can never score higher than the current best (cache hits carry at
  // most a 53 point penalty against 135 points per byte, ring entries are ordered by increasing distance), so
  // "passes the quick reject" reduces to unbroken > best_len, except when the best match already reaches the
  // block end (best_len == max_length), where the byte behind it decides.
  uint32_t best_len = 0;
  uint32_t best_score = kMinScore;
  const uint32_t brk = P.dict_break;
  bool walk_broken = false;
  bool folded = false;
  if constexpr (kRows) {
    // The candidates of this position are in the registers of its half of the wave (br_probe_pair_rows); entries behind
    // the end of the row hold kRowEnd and never take part.  Straight-line scoring (selects, no exec-mask regions).
    uint32_t lane_id = (uint32_t)BR_LANE;
    asm volatile("" : "+v"(lane_id));  // (see br_probe_pair_rows)
    const uint32_t c = lane_id & 31u;
    const bool in_range = (lane_id >> 5) == w && c < ndist + kRowEntries;
    const bool is_cache = c < ndist;
    const uint32_t prev = in_range ? m.r_prev : 0xffffffffu;
    const uint32_t unbroken = in_range ? m.r_len : 0u;
    const bool has = prev != 0xffffffffu;
    const uint32_t prev_ring = prev & P.ring_mask;
    const bool special_lane = has && ((prev_ring + unbroken > P.ring_mask) || unbroken == max_length || (prev_ring < brk && prev_ring + unbroken > brk));
    const bool cur_near_wrap = (cur & P.ring_mask) + max_length > P.ring_mask;
    if (!cur_near_wrap && __ballot(special_lane) == 0) {
      folded = true;
      const uint32_t backward = cur - prev;
      uint32_t score;
      bool type_ok;
      if constexpr (kH9) {
        score = is_cache ? br_score_cache<kH9>(P, unbroken, c & 15u) : br_score_ring<kH9>(P, unbroken, has ? backward : 1u);
        type_ok = is_cache ? (unbroken >= 3 || (unbroken == 2 && c < 2)) : unbroken >= 4;
      } else {
        // BackwardReferenceScoreUsingLastDistance - BackwardReferencePenaltyUsingLastDistance / BackwardReferenceScore
        // (mod.rs:1871-1889, 1151-1154) as one expression: score_per_byte * len + a per-candidate constant
        const uint32_t penalty = c == 0 ? 0u : 39u + ((0x1ca10u >> (c & 0xeu)) & 0xeu);
        const uint32_t lg = br_log2_floor_nonzero(has ? backward : 1u);
        const uint32_t bias = is_cache ? (30u * 8u * 8u + 15u) - penalty : 30u * 8u * 8u - 30u * lg;
        score = P.score_per_byte * unbroken + bias;
        const uint32_t min_len = is_cache ? (c < 2u ? 2u : 3u) : 4u;
        type_ok = unbroken >= min_len;
      }
      unsigned long long live = __ballot(has && type_ok);
      uint32_t best_lane = 64;
      while (live != 0) {
        const unsigned long long mm = live & __ballot(unbroken > best_len && score > best_score);
        if (mm == 0) break;
        const uint32_t f = (uint32_t)__ffsll((long long)mm) - 1u;
        best_len = BR_READLANE(unbroken, f);
        best_score = BR_READLANE(score, f);
        best_lane = f;
        live = f >= 63 ? 0ull : (mm >> (f + 1)) << (f + 1);
      }
      if (best_lane != 64) {
        out.len = best_len;
        out.distance = BR_READLANE(backward, best_lane);
        out.score = best_score;
        out.found = true;
      }
    } else {
      // rare (a candidate next to a ring-buffer wrap, the block end or the end of a custom dictionary): hand the
      // candidates to the general fold below through LDS
      BR_WAVE_SYNC();
      if (in_range) {
        s.cand_prev[w][c] = m.r_prev;
        s.cand_len[w][c] = m.r_len;
      }
      BR_WAVE_SYNC();
    }
  } else if (ncand <= 64) {
    // Fast path (all candidates in one pass of the wave, nothing near a ring-buffer wrap, the block end or the
    // custom-dictionary boundary): straight-line scoring, then a loop whose body is two compares, a mask AND,
    // a find-first-set and two lane reads.
    const uint32_t c = (uint32_t)BR_LANE;
    const bool in_range = c < ncand;
    const bool is_cache = c < ndist;
    const uint32_t prev = in_range ? s.cand_prev[w][c] : 0xffffffffu;
    const uint32_t unbroken = in_range ? s.cand_len[w][c] : 0u;
    const bool has = prev != 0xffffffffu;
    const unsigned long long stop = __ballot(in_range && !is_cache && !has);
    const unsigned long long below_stop = stop ? ((stop & (0ull - stop)) - 1ull) : ~0ull;  // lanes in front of the first stop
    const bool special_lane = has && (((prev & P.ring_mask) + unbroken > P.ring_mask) || unbroken == max_length ||
                                      ((prev & P.ring_mask) < brk && (prev & P.ring_mask) + unbroken > brk));
    const bool cur_near_wrap = (cur & P.ring_mask) + max_length > P.ring_mask;
    if (!cur_near_wrap && __ballot(special_lane) == 0) {
      folded = true;
      const uint32_t backward = cur - prev;
      const uint32_t score = is_cache ? br_score_cache<kH9>(P, unbroken, c & 15u) : br_score_ring<kH9>(P, unbroken, has ? backward : 1u);
      const bool type_ok = is_cache ? (unbroken >= 3 || (unbroken == 2 && c < 2)) : unbroken >= 4;
      unsigned long long live = __ballot(has && type_ok) & below_stop;
      uint32_t best_lane = 64;
      while (live != 0) {
        const unsigned long long m = live & __ballot(unbroken > best_len && score > best_score);
        if (m == 0) break;
        const uint32_t f = (uint32_t)__ffsll((long long)m) - 1u;
        best_len = BR_READLANE(unbroken, f);
        best_score = BR_READLANE(score, f);
        best_lane = f;
        live = f >= 63 ? 0ull : (m >> (f + 1)) << (f + 1);
      }
      if (best_lane != 64) {
        out.len = best_len;
        out.distance = BR_READLANE(backward, best_lane);
        out.score = best_score;
        out.found = true;
      }
    }
  }
  for (uint32_t base = 0; !folded && base < ncand && !walk_broken; base += 64) {
    const uint32_t c = base + (uint32_t)BR_LANE;
    const bool in_range = c < ncand;
    const bool is_cache = c < ndist;
    const uint32_t prev = in_range ? s.cand_prev[w][c] : 0xffffffffu;
    const uint32_t unbroken = in_range ? s.cand_len[w][c] : 0u;
    // the bucket walk stops at the first ring entry that is too far away
    const unsigned long long stop = __ballot(in_range && !is_cache && prev == 0xffffffffu);
    bool alive = in_range && prev != 0xffffffffu;
    if (stop != 0) {
      const int first_stop = __ffsll((long long)stop) - 1;
      if (BR_LANE > first_stop) alive = false;
      walk_broken = true;
    }
    uint32_t len = unbroken;
    if (brk != 0 && alive && (prev & P.ring_mask) < brk && (prev & P.ring_mask) + unbroken > brk) len = brk - (prev & P.ring_mask);  // fix_unbroken_len
    const uint32_t backward = cur - prev;
    bool type_ok;
    uint32_t score;
    if (is_cache) {
      type_ok = unbroken >= 3 || (unbroken == 2 && c < 2);
      score = br_score_cache<kH9>(P, len, c & 15u);
    } else {
      type_ok = unbroken >= 4;
      score = br_score_ring<kH9>(P, len, alive ? backward : 1u);
    }
    alive = alive && type_ok;
    // only needed when a match runs to the end of the block: compare the byte behind it (ring buffer semantics)
    bool tail_eq = false;
    if (alive && unbroken == max_length) tail_eq = br_unwritten_byte(P, t, cur + max_length) == t.text[prev + max_length];
    int start_lane = 0;
    for (;;) {
      if ((cur & P.ring_mask) + best_len > P.ring_mask) {
        // H9: if this happened before any ring entry was looked at, the position is not inserted either
        if (kH9 && base == 0 && (start_lane == 0 || (uint32_t)(start_lane - 1) < ndist)) out.stored = false;
        break;
      }
      const bool pass = alive && BR_LANE >= start_lane && !((prev & P.ring_mask) + best_len > P.ring_mask) &&
                        (unbroken > best_len || (unbroken == best_len && best_len == max_length && tail_eq)) && score > best_score;
      const unsigned long long m = __ballot(pass);
      if (m == 0) break;
      const int f = __ffsll((long long)m) - 1;
      best_len = BR_READLANE(len, f);
      best_score = BR_READLANE(score, f);
      out.len = best_len;
      out.distance = BR_READLANE(backward, f);
      out.score = best_score;
      out.found = true;
      start_lane = f + 1;
    }
  }
#endif
  if (before_dictionary) *before_dictionary = out;
  if (candidates_only) return out;
  br_dictionary_stage(P, t, ds, m.no_dict != 0, max_length, max_backward, out, [&](uint32_t i, uint32_t* item_out, uint32_t* matchlen_out) {
#if !BR_SCALAR
    const uint32_t dict_lane = 32u * w + ndist + kRowDictLane + i;
    *item_out = kRows ? BR_READLANE(m.r_prev, dict_lane) : BR_UNIFORM(s.cand_prev[w][ncand + i]);
    *matchlen_out = kRows ? BR_READLANE(m.r_len, dict_lane) : BR_UNIFORM(s.cand_len[w][ncand + i]);
#else
    *item_out = BR_UNIFORM(s.cand_prev[w][ncand + i]);
    *matchlen_out = BR_UNIFORM(s.cand_len[w][ncand + i]);
#endif
  });
  return out;
}

#if BR_SCALAR
// Common prefix of the text at a_pos and b_pos (a_pos < b_pos), at most `limit`: 8 bytes at a time; with the run table a
// pair that starts with 32 equal bytes of one value is settled by the lengths of the two runs (see br_match_len_wide).
BR_DEV uint32_t br_match_len_scalar(const ChainTables& t, uint32_t a_pos, uint32_t b_pos, uint32_t limit) {
  const uint8_t* a = t.text + a_pos;
  const uint8_t* b = t.text + b_pos;
  uint32_t i = 0;
  if (t.run_end != nullptr && limit > 32) {
    const uint64_t a0 = br_load64(a), pat = (a0 & 0xffull) * 0x0101010101010101ull;
    if (a0 == pat && br_load64(b) == pat && br_load64(a + 8) == pat && br_load64(b + 8) == pat && br_load64(a + 16) == pat &&
        br_load64(b + 16) == pat && br_load64(a + 24) == pat && br_load64(b + 24) == pat) {
      const uint32_t ra = t.run_end[a_pos] - a_pos, rb = t.run_end[b_pos] - b_pos;
      if (ra != rb) {
        const uint32_t r = ra < rb ? ra : rb;
        return r < limit ? r : limit;
      }
      if (ra >= limit) return limit;
      i = ra;
    }
  }
  // 64 bytes of either side per memory round trip while the match lasts (a lane has no other lane to wait behind)
  while (i + 64 <= limit) {
    uint64_t x[8];
    BR_UNROLL
    for (int k = 0; k < 8; ++k) x[k] = br_load64(a + i + 8 * k) ^ br_load64(b + i + 8 * k);
    uint32_t first = 64;
    BR_UNROLL
    for (int k = 7; k >= 0; --k)
      if (x[k] != 0) first = 8u * (uint32_t)k + (uint32_t)(__builtin_ctzll(x[k]) >> 3);
    if (first != 64) return i + first;
    i += 64;
  }
  return i + br_match_len(a + i, b + i, limit - i);
}

// the first 16 bytes at a text position, in registers
struct Head16 {
  uint64_t lo, hi;
};
BR_DEV Head16 br_load_head(const uint8_t* p) {
  Head16 h;
  h.lo = br_load64(p);
  h.hi = br_load64(p + 8);
  return h;
}
BR_DEV uint32_t br_head_common(const Head16& a, const Head16& b) {  // common prefix, 0..16
  const uint64_t x = a.lo ^ b.lo;
  if (x != 0) return (uint32_t)(__builtin_ctzll(x) >> 3);
  const uint64_t y = a.hi ^ b.hi;
  if (y != 0) return 8u + (uint32_t)(__builtin_ctzll(y) >> 3);
  return 16u;
}
BR_DEV uint8_t br_head_byte(const Head16& a, uint32_t i) { return (uint8_t)(i < 8 ? a.lo >> (8 * i) : a.hi >> (8 * (i - 8))); }

// AdvHasher::FindLongestMatch (mod.rs:1684-1812) for ONE position, candidate by candidate in the reference's order, the
// ring entries taken from the candidate row of the position (kRows chains).  The scalar form of br_probe_pair_rows +
// br_fold_probe, which the host emulation runs.  (It was also compiled for the device with one chain per LANE, 64 chains to a
// wavefront, for round 0: correct, and 3.5 x slower than one chain per wavefront -- 32 768 chains are half a wavefront per
// SIMD, and a lone wavefront issues the ~2 000 predicated instructions of an unrolled search at one per ~5 cycles.
// All loads of a search are issued together for that experiment: the row, then the first 16 bytes of every candidate.)
BR_DEV SearchResult br_search_rows_scalar(const Lz77Params& P, const ChainTables& t, DictState& ds, bool no_dict, uint32_t cur, const int32_t* cache,
                                          uint32_t blk_end) {
  constexpr uint32_t kCache = 4, kCand = kCache + kRowEntries;
  const uint32_t max_length = blk_end - cur;
  const uint32_t max_backward = cur < P.max_backward_limit ? cur : P.max_backward_limit;
  const uint32_t brk = P.dict_break;
  const uint32_t ndist = P.ndist < kCache ? P.ndist : kCache;  // (rows exist for ring depth 16, i.e. quality 5: four cache candidates)
  SearchResult out;
  out.len = 0;
  out.len_x_code = 0;
  out.distance = 0;
  out.score = kMinScore;
  out.found = false;
  out.stored = true;
  uint32_t best_len = 0, best_score = kMinScore;
  const uint32_t cur_ring = cur & P.ring_mask;
  // ---- where the candidates are
  uint32_t prev[kCand];
BR_UNROLL
  for (uint32_t c = 0; c < kCache; ++c) {
    const int64_t b = c < ndist ? (int64_t)cache[c] : 0;
    prev[c] = (b > 0 && b <= (int64_t)max_backward) ? cur - (uint32_t)b : 0xffffffffu;
  }
  {
    const uint32_t* row = t.rows + (size_t)cur * kRowEntries;
BR_UNROLL
    for (uint32_t i = 0; i < kRowEntries; ++i) prev[kCache + i] = row[i];
  }
  // ---- their first 16 bytes (one memory round trip for all of them)
  const Head16 here = br_load_head(t.text + cur);
  Head16 head[kCand];
BR_UNROLL
  for (uint32_t c = 0; c < kCand; ++c) head[c] = br_load_head(t.text + (prev[c] != 0xffffffffu ? prev[c] : cur));
  // ---- the fold, in the reference's order
  bool row_open = true;
BR_UNROLL
  for (uint32_t c = 0; c < kCand; ++c) {
    const bool is_cache = c < kCache;
    const uint32_t q = prev[c];
    if (q == 0xffffffffu) {
      if (!is_cache) row_open = false;  // (kRowEnd: the row ends here)
      continue;
    }
    if (!is_cache && !row_open) continue;
    // quick reject at best_len (mod.rs:1713-1718 / 1765-1773); the byte AT the block end is whatever the ring buffer holds there
    if (cur_ring + best_len > P.ring_mask || (q & P.ring_mask) + best_len > P.ring_mask) continue;
    bool same;
    if (best_len < 16 && cur + best_len < blk_end) {
      same = br_head_byte(here, best_len) == br_head_byte(head[c], best_len);
    } else {
      const uint8_t cb = (cur + best_len < blk_end) ? t.text[cur + best_len] : br_unwritten_byte(P, t, cur + best_len);
      same = cb == t.text[q + best_len];
    }
    if (!same) continue;
    if (!is_cache && (uint32_t)head[c].lo != (uint32_t)here.lo) continue;  // FindMatchLengthWithLimitMin4 == 0, static_dict.rs:134-147
    uint32_t unbroken = br_head_common(head[c], here);
    if (unbroken >= 16 && max_length > 16) unbroken = br_match_len_scalar(t, q, cur, max_length);
    if (unbroken > max_length) unbroken = max_length;
    if (is_cache ? !(unbroken >= 3 || (unbroken == 2 && c < 2)) : unbroken < 4) continue;
    uint32_t len = unbroken;
    if (brk != 0 && (q & P.ring_mask) < brk && (q & P.ring_mask) + unbroken > brk) len = brk - (q & P.ring_mask);  // fix_unbroken_len, mod.rs:42-54
    const uint32_t backward = cur - q;
    const uint32_t score = is_cache ? br_score_cache<false>(P, len, c) : br_score_ring<false>(P, len, backward);
    if (best_score < score) {
      best_score = score;
      best_len = len;
      out.len = len;
      out.distance = backward;
      out.score = score;
      out.found = true;
    }
  }
  const uint32_t first4 = (uint32_t)here.lo;
  br_dictionary_stage(P, t, ds, no_dict, max_length, max_backward, out, [&](uint32_t i, uint32_t* item_out, uint32_t* matchlen_out) {
    const uint32_t item = t.dict_hash[(((first4 * 0x1e35a7bdu) >> (32 - 14)) << 1) + i];
    uint32_t matchlen = 0;
    if (item != 0) {
      const uint32_t wlen = item & 0x1f;
      if (wlen <= max_length) matchlen = br_match_len(t.dict_data + t.dict_offsets_by_length[wlen] + wlen * (item >> 5), t.text + cur, wlen);
    }
    *item_out = item;
    *matchlen_out = matchlen;
  });
  return out;
}
#endif

// one record of ChainTables::search_log
BR_DEV void br_log_search(uint32_t* rec, const int32_t* cache, const SearchResult& found_before_dictionary) {
  for (int i = 0; i < 4; ++i) rec[i] = (uint32_t)cache[i];
  rec[4] = found_before_dictionary.len;
  rec[5] = found_before_dictionary.distance;
  rec[6] = found_before_dictionary.score;
  rec[7] = (found_before_dictionary.found ? 1u : 0u) | (found_before_dictionary.stored ? 2u : 0u);
}
BR_DEV bool br_same_as_logged(const uint32_t* rec, const SearchResult& r) {
  return rec[4] == r.len && rec[5] == r.distance && rec[6] == r.score && rec[7] == ((r.found ? 1u : 0u) | (r.stored ? 2u : 0u));
}

// search(x) for the parse loop: reuses the speculative second slot when it is still valid
template <bool kH9, bool kRows, bool kLive = false, bool kDeep = false>
BR_DEV SearchResult br_search(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, kRows, kDeep>& s, ProbeMeta& m, DictState& ds,
                              uint32_t x, const int32_t* cache, uint32_t cache_version, uint32_t blk_end, const LiveRing* live = nullptr) {
#if defined(BR_CHAIN_PROFILE)
  const unsigned long long t0 = BR_TICK();
#endif
#if BR_SCALAR
  if constexpr (kRows && !kLive && !kH9) return br_search_rows_scalar(P, t, ds, m.no_dict != 0, x, cache, blk_end);
#endif
  uint32_t w = 1;
  if (!(m.pos != 0xffffffffu && m.version == cache_version && x == m.pos + 1)) {
    if constexpr (!kRows) BR_SYNC();  // every lane is done reading the previous probe (register probes of the candidate-row and
                                      // live chains: nothing shared is read; a wavefront's LDS operations run in program order)
    br_probe_pair<kH9, kRows, kLive>(P, t, s, m, x, cache, cache_version, blk_end, live);
    w = 0;
#if defined(BR_CHAIN_PROFILE)
    m.t_probe += BR_TICK() - t0;
    m.n_probe++;
#endif
  }
#if defined(BR_CHAIN_PROFILE)
  const unsigned long long t1 = BR_TICK();
#endif
  SearchResult r;
  if constexpr (!kRows || kLive) {
    SearchResult pre;
    r = br_fold_probe<kH9, kRows>(P, t, s, m, w, ds, blk_end, &pre);
    if (m.log_on && BR_LANE == 0) br_log_search(t.search_log + (size_t)x * kSearchLogWords, cache, pre);
  } else {
    r = br_fold_probe<kH9, kRows>(P, t, s, m, w, ds, blk_end);
  }
  // FindLongestMatch files the position it has just searched (mod.rs:1794-1795)
  if constexpr (kLive) br_live_insert(*live, m.live_key[w], m.live_n[w], x);
#if defined(BR_CHAIN_PROFILE)
  m.t_fold += BR_TICK() - t1;
  m.n_fold++;
#endif
  return r;
}

// adv_prepare_distance_cache, mod.rs:632-651
BR_DEV void br_prepare_distance_cache(int32_t* dc, uint32_t ndist) {
  if (ndist > 4) {
    const int32_t last = dc[0];
    dc[4] = last - 1;
    dc[5] = last + 1;
    dc[6] = last - 2;
    dc[7] = last + 2;
    dc[8] = last - 3;
    dc[9] = last + 3;
    if (ndist > 10) {
      const int32_t next_last = dc[1];
      dc[10] = next_last - 1;
      dc[11] = next_last + 1;
      dc[12] = next_last - 2;
      dc[13] = next_last + 2;
      dc[14] = next_last - 3;
      dc[15] = next_last + 3;
    }
  }
}

// Repeats the cache and ring stages of the search a chain ran at position p -- with the distance cache it had then
// (ChainTables::search_log) and the candidate lists of NOW -- and says whether they find what they found then.  Used by
// the validation after a flag change (lz77_recheck_searches): a segment is parsed again only if one of its searches
// comes out differently, not whenever one of its candidate lists was touched.
template <bool kH9, bool kDeep = false>
BR_DEV bool br_recheck_search(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, false, kDeep>& s, uint32_t p, uint32_t blk_end) {
  const uint32_t* rec = t.search_log + (size_t)p * kSearchLogWords;
  BR_SYNC();
  int32_t* dc = s.dc;
  for (int i = 0; i < 4; ++i) dc[i] = (int32_t)BR_UNIFORM(rec[i]);
  for (int i = 4; i < 16; ++i) dc[i] = 0;
  br_prepare_distance_cache(dc, P.ndist);
  ProbeMeta m;
  m.pos = 0xffffffffu;
  m.version = 0;
  m.win_base = 0xffffff00u;
  m.r_prev = 0xffffffffu;
  m.r_len = 0;
  m.no_dict = 1;  // (the dictionary stage comes after what is compared here)
  m.log_on = 0;
  DictState ds;
  ds.lookups = ds.lookups0 = ds.matches = ds.matches0 = 0;
  ds.mode = 0;
  ds.maxdef = 0;
  ds.vlookups = 0;
  ds.vwould = 0;
  ds.vmaxdef = 0;
  BR_SYNC();
  br_probe_pair<kH9, false>(P, t, s, m, p, dc, 0, blk_end);
  SearchResult now;
  br_fold_probe<kH9, false>(P, t, s, m, 0, ds, blk_end, &now, true);
  return br_same_as_logged(rec, now);
}

struct FlagWriter {
  uint8_t* next;
  bool enabled;
  uint32_t hi;       // writes are clipped to positions < hi (the end of the chain's own segment)
  uint32_t tail_lo;  // positions >= tail_lo inside the block get the stitch flag
  uint8_t tail_value;
  BR_DEV void put(uint32_t q, uint8_t v) {
    if (q < hi) next[q] = v;
  }
  BR_DEV void one(uint32_t q, uint8_t v) {  // uniform call: lane 0 writes
    if (BR_LANE == 0 && enabled) put(q, v);
  }
  BR_DEV uint8_t unstored(uint32_t q) const { return q >= tail_lo ? tail_value : (uint8_t)0; }
  // [a, b) := v for q < split, static "not stored by the main loop" value for q >= split
  BR_DEV void range(uint32_t a, uint32_t b, uint32_t split) {
    if (!enabled) return;
    if (b > hi) b = hi;
    for (uint32_t q = a + BR_LANE; q < b; q += BR_NLANES) put(q, q < split ? (uint8_t)1 : unstored(q));
  }
  // What StoreRange(first, min(copy_end, store_end)) of a copy leaves at position q >= first = copy start + 2
  // (mod.rs:2516-2521): stored up to store_end; the H5 family files the first 4 * floor(n / 4) positions of a range of
  // n >= 8 through StoreRangeOptBatch, which writes masked positions (mod.rs:1163-1232) -- see kFlagMasked.  kMasked: the
  // chain models them (live chains; every other chain runs where no masked entry can exist, Lz77Params::masked_from).
  // (masked_from = Lz77Params::masked_from, handed in at the call so that it does not occupy a register through the parse loop)
  template <bool kMasked>
  BR_DEV uint8_t copy_value(uint32_t q, uint32_t first, uint32_t copy_end, uint32_t store_end, uint32_t masked_from) const {
    if constexpr (kMasked) {
      const uint32_t last = copy_end < store_end ? copy_end : store_end;
      if (q >= last) return unstored(q);
      // (a quad is filed as (start & mask) + 0..3: it is the START of the quad that decides -- the one quad that straddles the
      // end of the first ring-buffer revolution keeps true positions for all four)
      if (last >= first + 8 && q < first + ((last - first) & ~3u) && first + ((q - first) & ~3u) >= masked_from) return (uint8_t)(kFlagStored | kFlagMasked);
      return kFlagStored;
    } else {
      return q < store_end ? (uint8_t)1 : unstored(q);  // (q < copy_end at every call)
    }
  }
  // the StoreRange part [first, copy_end) of a copy
  template <bool kMasked>
  BR_DEV void copy_range(uint32_t first, uint32_t copy_end, uint32_t store_end, uint32_t masked_from) {
    if (!enabled) return;
    const uint32_t b = copy_end > hi ? hi : copy_end;
    for (uint32_t q = first + BR_LANE; q < b; q += BR_NLANES) put(q, copy_value<kMasked>(q, first, copy_end, store_end, masked_from));
  }
  // the part [a, b) of the step described by (kind, base, p1) -- see HeadKind; for a copy b is where it ends
  template <bool kH9, bool kMasked>
  BR_DEV void head(uint32_t kind, uint32_t base, uint32_t p1, uint32_t a, uint32_t b, uint32_t store_end, uint32_t masked_from) {
    if (!enabled || kind == kHeadNone) return;
    const uint32_t step_end = b;
    if (b > hi) b = hi;
    for (uint32_t q = a + BR_LANE; q < b; q += BR_NLANES) {
      uint8_t v;
      if (kind == kHeadCopy) {
        // q <= base: lazily delayed literals and the position the match starts at (all searched and stored)
        // p1 bit 0: base + 1 was probed; bit 2: ... but not inserted; bits 8 + j: base - j was searched but not inserted
        // (the H9 ring-end case, see SearchResult::stored)
        if (q <= base) v = (kH9 && base - q < 8 && ((p1 >> (8 + base - q)) & 1u)) ? kFlagSearched : (uint8_t)(kFlagStored | kFlagSearched);
        else if (q == base + 1) v = (p1 & 1u) ? ((kH9 && (p1 & 4u)) ? kFlagSearched : (uint8_t)(kFlagStored | kFlagSearched)) : unstored(q);
        else v = copy_value<kMasked>(q, base + 2, step_end, store_end, masked_from);
      } else if (kind == kHeadUnstored) {
        v = unstored(q);
      } else if (kind == kHeadVec4) {
        v = ((q - base) & 3) == 0;
      } else {
        v = ((q - base) & 1) == 0;
      }
      put(q, v);
    }
  }
};

// One chain: parses segment `seg` from `entry`, writes commands, flags and `exit`.
// `next` receives (in every lane) the entry state this parse hands to the following segment.
// what a live chain needs of a block's parse to enter the next block by itself (extend_last_command, encode.rs:2435-2437)
struct BlockTail {
  uint32_t n_cmds, n_lits, insert_len, ext_len, last_dist_code, last_copy_len;
};

// How a list launch takes a segment (checkpoints, see Checkpoint).
struct Reparse {
  uint32_t mode;     // 0: from its entry to its end.  1: from its entry, may stop at a checkpoint (the entry changed, or a chain
                     // walked into it).  2: the entry is the one it was parsed with last: restart in front of rows_lo, may stop.
  uint32_t rows_lo;  // lowest / highest searched position of the segment whose candidate row changed since it was parsed
  uint32_t rows_hi;  // last (0xffffffff / 0: none)
};
static constexpr uint32_t kSpliceGap = 32;  // commands of room between a new head and the old tail (br_parse_segment)

// lane 0 writes one checkpoint record
BR_DEV void br_write_checkpoint(Checkpoint* rec, const Checkpoint& c) {
  if (BR_LANE == 0) *rec = c;
}

template <bool kH9, bool kRows, bool kLive = false, bool kSplice = false, bool kDeep = false>
BR_DEV uint32_t br_parse_segment(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, kRows, kDeep>& s, const Segment& seg_in,
                                 const SegEntry& entry, SegExit& exit_out, SegEntry& next, const LiveRing* live = nullptr,
                                 BlockTail* tail = nullptr, const Reparse* rp = nullptr) {
  const uint32_t pos_end = BR_UNIFORM(seg_in.blk_end);
  Segment seg;
  seg.start = BR_UNIFORM(seg_in.start);
  seg.end = BR_UNIFORM(seg_in.end);
  seg.blk_start = BR_UNIFORM(seg_in.blk_start);
  seg.blk_end = pos_end;
  seg.flags = BR_UNIFORM(seg_in.flags);
  seg.cmd_base = BR_UNIFORM(seg_in.cmd_base);
  seg.block_index = 0;
  seg.cmd_cap = BR_UNIFORM(seg_in.cmd_cap);
  const uint32_t htl = P.htl;
  const uint32_t window = P.spree_window;
  uint32_t position = BR_UNIFORM(entry.pos);
  uint32_t apply = BR_UNIFORM(entry.apply);
  uint32_t insert_length = 0;  // literals carried in are added by the host fix-up (CmdPatch kind 2)
  int32_t* dc = s.dc;
  for (int i = 0; i < 4; ++i) dc[i] = (int32_t)BR_UNIFORM(entry.cache[i]);
  for (int i = 4; i < 16; ++i) dc[i] = 0;
  DictState ds;
  ds.lookups = ds.lookups0 = BR_UNIFORM(entry.dict_lookups);
  ds.matches = ds.matches0 = BR_UNIFORM(entry.dict_matches);
  ds.mode = 0;
  ds.maxdef = -(1 << 30);
  ds.vlookups = 0;
  ds.vwould = 0;
  ds.vmaxdef = -(1 << 30);
  FlagWriter fw;
  fw.next = t.flags_next;
  fw.enabled = !(seg.flags & kSegWarmup);
  fw.hi = seg.end;
  uint32_t tail_kind = kHeadNone, tail_base = 0, tail_p1 = 0;
  fw.tail_lo = pos_end - 3;
  fw.tail_value = (seg.flags & kSegTailStitched) ? 1 : 0;
  uint32_t n_cmds = 0, n_lits = 0, n_searches = 0, ext_len = 0, n_pushes = 0, n_bad = 0;
  uint32_t last_dist_code = 0xffffffffu, last_copy_len = 0;
  uint32_t cache_version = 0;
  ProbeMeta probe;
  probe.pos = 0xffffffffu;
  probe.version = 0;
  probe.win_base = 0xffffff00u;
  probe.r_prev = 0xffffffffu;
  probe.r_len = 0;
  probe.kwin_base = 0xffffff00u;
  probe.kwin = 0;
  // Once the throttle (matches < lookups >> 7, mod.rs:1957-1960) has tripped it stays tripped: nothing is looked up any
  // more, so neither counter moves.  With exact counters at the entry the chain need not even keep the virtual books.
  probe.no_dict = (P.use_dictionary && BR_UNIFORM(entry.dict_exact) && ds.matches < (ds.lookups >> 7)) ? 1u : 0u;
  probe.log_on = ((!kRows || kLive) && t.search_log != nullptr && fw.enabled) ? 1u : 0u;
#if defined(BR_CHAIN_PROFILE)
  probe.t_probe = probe.t_fold = probe.n_probe = probe.n_fold = probe.t_setup = probe.t_refill = probe.n_refill = 0;
  const unsigned long long t_begin = BR_TICK();
#endif
  Command* cmds = t.cmds + (size_t)seg.cmd_base;

  if (seg.flags & kSegFirstInBlock) {
    position = seg.blk_start;
    if (entry.ext_allowed) {
      // extend_last_command, encode.rs:360-400: the previous copy continues while bytes keep matching
      const uint32_t d = (uint32_t)dc[0];
      uint32_t n = 0;
      const uint32_t limit = pos_end - position;
      // lane-parallel byte compare in strides of the wave width
      for (;;) {
        uint32_t step = limit - n;
        if (step == 0) break;
        const uint32_t m = br_match_len(t.text + position + n, t.text + position + n - d, step < 64 ? step : 64);
        n += m;
        if (m < 64) break;
      }
      ext_len = n;
      fw.range(position, position + n, 0);
      tail_kind = kHeadUnstored;
      tail_base = position;
      position += n;
    }
    apply = position + window;
  }
  const uint32_t num_bytes = pos_end - (seg.blk_start + ((seg.flags & kSegFirstInBlock) ? ext_len : 0));
  (void)num_bytes;
  // store_end, mod.rs:2397-2404.  For segments after the first one the extension length is unknown,
  // but whenever (pos_end - start) < lookahead the loop below cannot run anyway.
  const uint32_t store_end = pos_end >= htl ? pos_end - htl + 1 : 0;
  br_prepare_distance_cache(dc, P.ndist);
  if (!(seg.flags & kSegFirstInBlock)) {
    // the part of the previous chain's last step that lies in this segment
    tail_kind = BR_UNIFORM(entry.head_kind);
    tail_base = BR_UNIFORM(entry.head_base);
    tail_p1 = BR_UNIFORM(entry.head_p1);
    if (position > seg.start) fw.template head<kH9, kLive>(tail_kind, tail_base, tail_p1, seg.start, position, store_end, kH9 ? kNeverMasked : P.masked_from);
  }

  // ---- checkpoints (see Checkpoint)
  constexpr bool kCheckpoints = kRows && !kLive && !kH9;
  const bool cp_on = kCheckpoints && t.checkpoints != nullptr && fw.enabled;
  uint32_t next_cp = cp_on ? (seg.start / kCheckpointStride + 1u) * kCheckpointStride : 0xffffffffu;  // first boundary behind the start
  bool spliced = false;
  uint32_t sp_rec_cmds = 0, sp_rec_lits = 0, sp_rec_searches = 0, sp_rec_pushes = 0, sp_rec_bad = 0, sp_rec_lookups = 0, sp_rec_matches = 0,
           sp_rec_entry_lookups = 0, sp_rec_entry_matches = 0;  // the old record at the splice point
  uint32_t splice_room = 0;     // kSplice: how far the old commands were moved up; 0 = this parse cannot be spliced
  uint32_t old_n_cmds = 0;      // kSplice: commands of the parse being replaced
  uint32_t walked_from = BR_UNIFORM(entry.pos);
  if constexpr (kSplice && kCheckpoints) {
    if (cp_on && rp != nullptr && rp->mode != 0) {
      old_n_cmds = BR_UNIFORM(exit_out.n_cmds);
      uint32_t keep_cmds = 0;  // commands of the old parse that stay where they are (restart)
      const bool no_restart = (t.splice_off & 1u) != 0, no_stop = (t.splice_off & 2u) != 0;  // (diagnosis: BROTLI_MI355X_SPLICE_OFF)
      if (rp->mode == 2 && rp->rows_lo != 0xffffffffu && !no_restart) {
        // same entry, rows changed from rows_lo on: everything the old parse did in front of the last checkpoint that lies
        // at or in front of rows_lo would come out again
        uint32_t b = (rp->rows_lo < seg.end ? rp->rows_lo : seg.end - 1u) / kCheckpointStride * kCheckpointStride;
        for (; b > seg.start; b -= kCheckpointStride) {
          const Checkpoint* rec = t.checkpoints + b / kCheckpointStride;
          if (BR_UNIFORM(rec->valid) != kCheckpointValid || BR_UNIFORM(rec->pos) > rp->rows_lo) continue;
#if defined(BROTLI_HOST_EMU) && defined(BR_DEBUG_SPLICE)
          {
            // shadow: a parse from the entry must arrive at the record's position in the record's state
            ChainTables t2 = t;
            t2.checkpoints = nullptr;
            Segment g2 = seg_in;
            g2.end = rec->pos;
            g2.flags &= ~(uint32_t)kSegLastInBlock;
            std::vector<Command> keep(cmds, cmds + exit_out.n_cmds + 40);
            SegExit sx{};
            SegEntry sn{};
            br_parse_segment<kH9, kRows, false, false>(P, t2, s, g2, entry, sx, sn);
            for (size_t i = 0; i < keep.size(); ++i) cmds[i] = keep[i];
            bool ok = sx.pos == rec->pos && sx.apply == rec->apply && sx.insert_len == rec->insert_len && sx.n_cmds == rec->n_cmds && sx.n_lits == rec->n_lits;
            for (int i = 0; i < 4; ++i) ok = ok && sx.cache[i] == rec->dc[i];
            if (!ok)
              fprintf(stderr, "RESTART RECORD STALE seg [%u,%u) rows %u..%u record at %u: pos %u apply %u ins %u cmds %u lits %u dc %d %d %d %d entryLM %u/%u | parse now: pos %u apply %u ins %u cmds %u lits %u dc %d %d %d %d | entry pos %u apply %u cache %d %d %d %d LM %u/%u exact %u head %u/%u\n",
                      seg.start, seg.end, rp->rows_lo, rp->rows_hi, b, rec->pos, rec->apply, rec->insert_len, rec->n_cmds, rec->n_lits, rec->dc[0], rec->dc[1], rec->dc[2], rec->dc[3],
                      rec->entry_lookups, rec->entry_matches, sx.pos, sx.apply, sx.insert_len, sx.n_cmds, sx.n_lits, sx.cache[0], sx.cache[1], sx.cache[2], sx.cache[3], entry.pos, entry.apply,
                      entry.cache[0], entry.cache[1], entry.cache[2], entry.cache[3], entry.dict_lookups, entry.dict_matches, entry.dict_exact, entry.head_kind, entry.head_base);
          }
#endif
          position = BR_UNIFORM(rec->pos);
          insert_length = BR_UNIFORM(rec->insert_len);
          apply = BR_UNIFORM(rec->apply);
          BR_SYNC();
          for (int i = 0; i < 4; ++i) dc[i] = (int32_t)BR_UNIFORM(rec->dc[i]);
          BR_SYNC();
          br_prepare_distance_cache(dc, P.ndist);
          n_cmds = BR_UNIFORM(rec->n_cmds);
          n_lits = BR_UNIFORM(rec->n_lits);
          n_searches = BR_UNIFORM(rec->n_searches);
          n_pushes = BR_UNIFORM(rec->n_pushes);
          n_bad = BR_UNIFORM(rec->n_bad);
          last_dist_code = BR_UNIFORM(rec->last_dist_code);
          last_copy_len = BR_UNIFORM(rec->last_copy_len);
          ext_len = BR_UNIFORM(rec->ext_len);
          tail_kind = BR_UNIFORM(rec->tail_kind);
          tail_base = BR_UNIFORM(rec->tail_base);
          tail_p1 = BR_UNIFORM(rec->tail_p1);
          // the throttle counters are absolute: they move with the entry (the host has judged the old parse valid under
          // the entry of now, DictTracker::Consume)
          if (probe.no_dict) {
            // off for good under the entry of now (the old parse may have run with a guess and looked things up): nothing is
            // looked up, the counters stay where the entry has them, and a consult makes the parse one that "ran blind"
            ds.mode = BR_UNIFORM(rec->d_mode) != 0 ? 4u : 0u;
          } else {
            ds.lookups = BR_UNIFORM(rec->d_lookups) + (ds.lookups0 - BR_UNIFORM(rec->entry_lookups));
            ds.matches = BR_UNIFORM(rec->d_matches) + (ds.matches0 - BR_UNIFORM(rec->entry_matches));
            ds.mode = BR_UNIFORM(rec->d_mode);
          }
          ds.maxdef = (int32_t)BR_UNIFORM(rec->d_maxdef);
          ds.vlookups = BR_UNIFORM(rec->d_vlookups);
          ds.vwould = BR_UNIFORM(rec->d_vwould);
          ds.vmaxdef = (int32_t)BR_UNIFORM(rec->d_vmaxdef);
          keep_cmds = n_cmds;
          next_cp = b + kCheckpointStride;
          walked_from = position;
          break;
        }
      }
      // Room for the new commands: the old ones from keep_cmds on move up by kSpliceGap, so that a head that comes out a few
      // commands longer than the old one does not run over the tail it may be spliced with.  (Highest chunk first: a chunk
      // is read whole before it is written, and lands only on slots that have been moved already.)
      if (!no_stop && old_n_cmds >= keep_cmds && old_n_cmds + kSpliceGap <= seg.cmd_cap) {
        Command* slab = cmds;
        for (uint32_t hi = old_n_cmds; hi > keep_cmds;) {
          const uint32_t lo = hi - keep_cmds > BR_NLANES ? hi - BR_NLANES : keep_cmds;
          const uint32_t i = lo + BR_LANE;
          Command v = slab[i < hi ? i : lo];
          BR_SYNC();
          if (i < hi) {
            // (device-scope stores: the splice below reads these slots back through the L2)
            uint32_t* dst = (uint32_t*)(slab + i + kSpliceGap);
            BR_LIVE_ST32(dst, v.insert_len_);
            BR_LIVE_ST32(dst + 1, v.copy_len_);
            BR_LIVE_ST32(dst + 2, v.dist_extra_);
            BR_LIVE_ST32(dst + 3, (uint32_t)v.cmd_prefix_ | ((uint32_t)v.dist_prefix_ << 16));
          }
          hi = lo;
        }
        BR_SYNC();
        splice_room = kSpliceGap;
      }
    }
  }

  while (position + htl < pos_end && position < seg.end) {
    if constexpr (kCheckpoints) {
      // the first loop-top position at or behind a boundary: stop here if this is where the old parse was, in the same state;
      // otherwise leave the record for the next parse
      while (next_cp <= position && next_cp < seg.end) {
        Checkpoint* rec = t.checkpoints + next_cp / kCheckpointStride;
        if constexpr (kSplice) {
          if (splice_room != 0 && (rp->rows_hi == 0 || rp->rows_hi < position) && BR_UNIFORM(rec->valid) == kCheckpointValid &&
              BR_UNIFORM(rec->pos) == position && BR_UNIFORM(rec->insert_len) == insert_length && BR_UNIFORM(rec->apply) == apply &&
              n_cmds <= BR_UNIFORM(rec->n_cmds) + splice_room) {
            bool same = true;
            for (int i = 0; i < 4; ++i) same = same && (int32_t)BR_UNIFORM(rec->dc[i]) == dc[i];
            // the static dictionary: does the old parse of the rest hold under the throttle state this parse arrives with?
            const uint32_t old_mode = BR_UNIFORM(exit_out.dict_mode);
            if (same && P.use_dictionary) {
              if (probe.no_dict) {
                // off for good: the old rest must not have used a dictionary match
                same = old_mode == 0 || old_mode == 2 || old_mode == 4 ||
                       (old_mode == 1 && BR_UNIFORM(exit_out.dict_matches) == BR_UNIFORM(exit_out.dict_entry_matches));
              } else if (BR_UNIFORM(entry.dict_exact) && ds.mode <= 1 && old_mode <= 1 && !BR_UNIFORM(rec->no_dict)) {
                // on, with the true counters: the old rest found it on at every consult; it stays on under the counters of
                // now if they cover the largest deficit (lookups - 128 matches) the rest can have run up from here
                if (old_mode == 1) {
                  const int64_t d_old_here = (int64_t)(BR_UNIFORM(rec->d_lookups) - BR_UNIFORM(rec->entry_lookups)) -
                                             128ll * (int64_t)(BR_UNIFORM(rec->d_matches) - BR_UNIFORM(rec->entry_matches));
                  const int64_t need = (int64_t)(int32_t)BR_UNIFORM(exit_out.dict_maxdef) - d_old_here;
                  same = 128ll * (int64_t)ds.matches - (int64_t)ds.lookups + 127 >= need;
                }
              } else {
                same = false;
              }
            }
            if (same) spliced = true;  // (the record below is still written: same state, but the counts of the head parsed now)
          }
        }
        if (spliced) {
          // what the splice needs of the old record
          sp_rec_cmds = BR_UNIFORM(rec->n_cmds);
          sp_rec_lits = BR_UNIFORM(rec->n_lits);
          sp_rec_searches = BR_UNIFORM(rec->n_searches);
          sp_rec_pushes = BR_UNIFORM(rec->n_pushes);
          sp_rec_bad = BR_UNIFORM(rec->n_bad);
          sp_rec_lookups = BR_UNIFORM(rec->d_lookups);
          sp_rec_matches = BR_UNIFORM(rec->d_matches);
          sp_rec_entry_lookups = BR_UNIFORM(rec->entry_lookups);
          sp_rec_entry_matches = BR_UNIFORM(rec->entry_matches);
        }
        if (cp_on) {
          Checkpoint c;
          c.pos = position;
          c.insert_len = insert_length;
          c.apply = apply;
          for (int i = 0; i < 4; ++i) c.dc[i] = dc[i];
          c.n_cmds = n_cmds;
          c.n_lits = n_lits;
          c.n_searches = n_searches;
          c.n_pushes = n_pushes;
          c.n_bad = n_bad;
          c.last_dist_code = last_dist_code;
          c.last_copy_len = last_copy_len;
          c.ext_len = ext_len;
          c.tail_kind = tail_kind;
          c.tail_base = tail_base;
          c.tail_p1 = tail_p1;
          c.d_lookups = ds.lookups;
          c.d_matches = ds.matches;
          c.d_mode = ds.mode;
          c.d_maxdef = ds.maxdef;
          c.d_vlookups = ds.vlookups;
          c.d_vwould = ds.vwould;
          c.d_vmaxdef = ds.vmaxdef;
          c.entry_lookups = ds.lookups0;
          c.entry_matches = ds.matches0;
          c.no_dict = probe.no_dict;
          c.valid = kCheckpointValid;
          c.pad[0] = c.pad[1] = c.pad[2] = 0;
          br_write_checkpoint(rec, c);
        }
        if (spliced) break;
        next_cp += kCheckpointStride;
      }
      if (spliced) break;
    }
    SearchResult sr = br_search<kH9, kRows, kLive>(P, t, s, probe, ds, position, dc, cache_version, pos_end, live);
    n_searches++;
    if (sr.found) {
      int delayed = 0;
      bool next_probed, next_stored = true;
      uint32_t special = 0;  // bit j: the search j positions before the match start was not inserted (H9 ring end)
      for (;;) {
        SearchResult sr2 = br_search<kH9, kRows, kLive>(P, t, s, probe, ds, position + 1, dc, cache_version, pos_end, live);
        n_searches++;
        next_probed = true;
        if (kH9) next_stored = sr2.stored;
        if (sr2.found && sr2.score >= sr.score + 175) {
          fw.one(position, (!kH9 || sr.stored) ? (uint8_t)(kFlagStored | kFlagSearched) : kFlagSearched);
          if (kH9) special = (special << 1) | (sr.stored ? 0u : 2u);
          position++;
          insert_length++;
          sr = sr2;
          next_probed = false;
          if (++delayed < 4 && position + htl < pos_end) continue;
        }
        break;
      }
      if (kH9) special |= sr.stored ? 0u : 1u;
      apply = position + 2 * sr.len + window;
      const uint32_t max_distance = position < P.max_backward_limit ? position : P.max_backward_limit;
      const uint32_t distance_code = br_compute_distance_code(sr.distance, max_distance, dc);
      if (sr.distance <= max_distance && distance_code > 0) {
        dc[3] = dc[2];
        dc[2] = dc[1];
        dc[1] = dc[0];
        dc[0] = (int32_t)sr.distance;
        n_pushes++;
        cache_version++;
        br_prepare_distance_cache(dc, P.ndist);
      }
      if (BR_LANE == 0 && n_cmds < seg.cmd_cap && fw.enabled) cmds[n_cmds] = br_raw_command(insert_length, sr.len, sr.len ^ sr.len_x_code, distance_code);
      n_cmds++;
      if (sr.len < 2) n_bad++;
      n_lits += insert_length;
      insert_length = 0;
      last_dist_code = distance_code;
      last_copy_len = sr.len;
      // hash-table side effects: position searched, position+1 only if probed, then StoreRange
      tail_kind = kHeadCopy;
      tail_base = position;
      tail_p1 = next_probed ? 1u : 0u;
      if (kH9) tail_p1 |= ((next_probed && !next_stored) ? 4u : 0u) | ((special & 0xffu) << 8);
      fw.one(position, (!kH9 || sr.stored) ? (uint8_t)(kFlagStored | kFlagSearched) : kFlagSearched);
      if (sr.len > 1)
        fw.one(position + 1, next_probed ? ((!kH9 || next_stored) ? (uint8_t)(kFlagStored | kFlagSearched) : kFlagSearched) : fw.unstored(position + 1));
      if (sr.len > 2) {
        fw.template copy_range<kLive>(position + 2, position + sr.len, store_end, kH9 ? kNeverMasked : P.masked_from);
        if constexpr (kLive) br_live_store_copy(*live, position + 2, position + sr.len < store_end ? position + sr.len : store_end, P.masked_from, probe.kwin, probe.kwin_base);
      }
      position += sr.len;
    } else {
      fw.one(position, (!kH9 || sr.stored) ? (uint8_t)(kFlagStored | kFlagSearched) : kFlagSearched);
      insert_length++;
      position++;
      if (position > apply) {
        const uint32_t margin = htl - 1 > 4 ? htl - 1 : 4;
        if (position + 16 >= pos_end - margin) {
          tail_kind = kHeadUnstored;
          tail_base = position;
          fw.range(position, pos_end, 0);
          insert_length += pos_end - position;
          position = pos_end;
        } else if (position > apply + 4 * window) {
          // Store4Vec4: position, +4, +8, +12
          tail_kind = kHeadVec4;
          tail_base = position;
          for (uint32_t q = position + BR_LANE; fw.enabled && q < position + 16; q += BR_NLANES) {
            const uint8_t v = ((q - position) & 3) == 0;
            fw.put(q, v);
          }
          if constexpr (kLive) br_live_store(*live, position, 4, 4, 1, 0);
          insert_length += 16;
          position += 16;
        } else {
          // StoreEvenVec4: position, +2, +4, +6
          tail_kind = kHeadEven4;
          tail_base = position;
          for (uint32_t q = position + BR_LANE; fw.enabled && q < position + 8; q += BR_NLANES) {
            const uint8_t v = ((q - position) & 1) == 0;
            fw.put(q, v);
          }
          if constexpr (kLive) br_live_store(*live, position, 2, 4, 1, 0);
          insert_length += 8;
          position += 8;
        }
      }
    }
  }
  const uint32_t walked_to = position;
  (void)walked_to;
  bool dict_spliced = false;
  uint32_t sp_lookups = 0, sp_matches = 0, sp_mode = 0;
  int32_t sp_maxdef = 0;
  if constexpr (kCheckpoints) {
    if (cp_on && !spliced) {
      // boundaries this parse never reached at a loop top (a copy or a jump carried it past them, or the block ended): whatever
      // record sits there belongs to an older parse
      for (; next_cp < seg.end; next_cp += kCheckpointStride)
        if (BR_LANE == 0) t.checkpoints[next_cp / kCheckpointStride].valid = 0;
    }
  }
  if constexpr (kSplice && kCheckpoints) {
    if (spliced) {
      // ---- splice: the head parsed now + the rest of the old parse from the record at next_cp on
      const uint32_t rec_cmds = sp_rec_cmds;
      const uint32_t tail_cmds = old_n_cmds - rec_cmds;
      {
        // the old rest sits at [rec_cmds + room, old_n_cmds + room) and goes to [n_cmds, ...): downwards or nowhere, lowest chunk first
        Command* slab = cmds;
        const uint32_t from = rec_cmds + splice_room;
        if (from != n_cmds) {
          for (uint32_t done = 0; done < tail_cmds; done += BR_NLANES) {
            const uint32_t i = done + BR_LANE;
            // (read from the L2: these slots were written a moment ago, by this chain's move above, and a line of the
            // vector L1 may still hold what was there before)
            const uint32_t* src = (const uint32_t*)(slab + from + (i < tail_cmds ? i : 0u));
            uint32_t w0 = BR_LIVE_LD32(src), w1 = BR_LIVE_LD32(src + 1), w2 = BR_LIVE_LD32(src + 2), w3 = BR_LIVE_LD32(src + 3);
            BR_SYNC();
            if (i < tail_cmds) {
              uint32_t* dst = (uint32_t*)(slab + n_cmds + i);
              dst[0] = w0;
              dst[1] = w1;
              dst[2] = w2;
              dst[3] = w3;
            }
          }
        }
      }
      // records behind the splice point count from the old head: dropped (the next full parse writes them again)
      for (uint32_t b = next_cp + kCheckpointStride; b < seg.end; b += kCheckpointStride)
        if (BR_LANE == 0) t.checkpoints[b / kCheckpointStride].valid = 0;
      const uint32_t old_mode = BR_UNIFORM(exit_out.dict_mode);
      if (P.use_dictionary) {
        dict_spliced = true;
        if (probe.no_dict) {
          sp_lookups = ds.lookups;
          sp_matches = ds.matches;
          sp_mode = (ds.mode == 0 && old_mode == 0) ? 0u : 4u;
          sp_maxdef = ds.maxdef;
        } else {
          const int64_t d_old_here = (int64_t)(sp_rec_lookups - sp_rec_entry_lookups) - 128ll * (int64_t)(sp_rec_matches - sp_rec_entry_matches);
          const int64_t d_new_here = (int64_t)(ds.lookups - ds.lookups0) - 128ll * (int64_t)(ds.matches - ds.matches0);
          // (counts relative to the entry each was taken with: the record may stem from an older parse than the exit -- a restart
          // leaves the records in front of it alone)
          sp_lookups = ds.lookups + ((BR_UNIFORM(exit_out.dict_lookups) - BR_UNIFORM(exit_out.dict_entry_lookups)) - (sp_rec_lookups - sp_rec_entry_lookups));
          sp_matches = ds.matches + ((BR_UNIFORM(exit_out.dict_matches) - BR_UNIFORM(exit_out.dict_entry_matches)) - (sp_rec_matches - sp_rec_entry_matches));
          sp_mode = (ds.mode | old_mode) ? 1u : 0u;
          int64_t md = ds.mode == 1 ? (int64_t)ds.maxdef : -(1ll << 30);
          if (old_mode == 1) {
            const int64_t rest = d_new_here + ((int64_t)(int32_t)BR_UNIFORM(exit_out.dict_maxdef) - d_old_here);  // (an upper bound: the old maximum may lie in the old head)
            if (rest > md) md = rest;
          }
          sp_maxdef = (int32_t)(md < -(1ll << 30) ? -(1ll << 30) : (md > (1ll << 30) ? (1ll << 30) : md));
          ds.lookups = sp_lookups;
          ds.matches = sp_matches;
        }
      }
      n_lits += BR_UNIFORM(exit_out.n_lits) - sp_rec_lits;
      n_searches += BR_UNIFORM(exit_out.n_searches) - sp_rec_searches;
      n_bad += BR_UNIFORM(exit_out.bad_commands) - sp_rec_bad;
      n_pushes += BR_UNIFORM(exit_out.n_pushes_all) - sp_rec_pushes;
      if (tail_cmds != 0) {
        last_dist_code = BR_UNIFORM(exit_out.last_dist_code);
        last_copy_len = BR_UNIFORM(exit_out.last_copy_len);
      }
      n_cmds += tail_cmds;
      position = BR_UNIFORM(exit_out.pos);
      apply = BR_UNIFORM(exit_out.apply);
      insert_length = BR_UNIFORM(exit_out.insert_len);
      tail_kind = BR_UNIFORM(exit_out.tail_kind);
      tail_base = BR_UNIFORM(exit_out.tail_base);
      tail_p1 = BR_UNIFORM(exit_out.tail_p1);
      const int32_t c0 = (int32_t)BR_UNIFORM(exit_out.cache[0]), c1 = (int32_t)BR_UNIFORM(exit_out.cache[1]),
                    c2 = (int32_t)BR_UNIFORM(exit_out.cache[2]), c3 = (int32_t)BR_UNIFORM(exit_out.cache[3]);
      BR_SYNC();
      dc[0] = c0;
      dc[1] = c1;
      dc[2] = c2;
      dc[3] = c3;
      BR_SYNC();
    }
  }
  if (seg.flags & kSegLastInBlock) {
    if (position < pos_end) fw.range(position, pos_end, 0);
    insert_length += pos_end - position;
    position = pos_end;
  }
#if !defined(BROTLI_HOST_EMU)
  if (BR_LANE == 0 && t.work) {
    atomicAdd(t.work + 0, (unsigned long long)((spliced ? walked_to : position) - walked_from));
    atomicAdd(t.work + 1, (unsigned long long)n_searches);
    if (fw.enabled) atomicAdd(t.work + 2, (unsigned long long)n_cmds);
  }
#endif
  if (BR_LANE == 0) {
    exit_out.pos = position;
    exit_out.apply = apply;
    for (int i = 0; i < 4; ++i) exit_out.cache[i] = dc[i];
    exit_out.insert_len = insert_length;
    exit_out.n_cmds = n_cmds;
    exit_out.n_lits = n_lits;
    exit_out.ext_len = ext_len;
    // (a chain that only ever saw the dictionary off reports its virtual bookkeeping instead, see DictState)
    exit_out.dict_lookups = ds.mode == 2 ? ds.lookups + ds.vlookups : ds.lookups;
    exit_out.dict_matches = ds.mode == 2 ? ds.matches + ds.vwould : ds.matches;
    exit_out.last_dist_code = last_dist_code;
    exit_out.bad_commands = n_bad;
    exit_out.n_searches = n_searches;
    exit_out.last_copy_len = last_copy_len;
    exit_out.dict_mode = ds.mode;
    exit_out.dict_maxdef = ds.mode == 2 ? ds.vmaxdef : ds.maxdef;
    exit_out.n_pushes = n_pushes < 4 ? n_pushes : 4u;
    exit_out.n_pushes_all = n_pushes;
    exit_out.dict_entry_lookups = ds.lookups0;
    exit_out.dict_entry_matches = ds.matches0;
    if (dict_spliced) {
      exit_out.dict_lookups = sp_lookups;
      exit_out.dict_matches = sp_matches;
      exit_out.dict_mode = sp_mode;
      exit_out.dict_maxdef = sp_maxdef;
    }
    exit_out.tail_kind = position > seg.end ? tail_kind : (uint32_t)kHeadNone;
    exit_out.tail_base = position > seg.end ? tail_base : 0u;
    exit_out.tail_p1 = position > seg.end ? tail_p1 : 0u;
  }
#if defined(BR_CHAIN_PROFILE)
  if (BR_LANE == 0) {
    atomicAdd(&g_chain_prof[0], BR_TICK() - t_begin);
    atomicAdd(&g_chain_prof[1], probe.t_probe);
    atomicAdd(&g_chain_prof[2], probe.t_fold);
    atomicAdd(&g_chain_prof[3], probe.n_probe);
    atomicAdd(&g_chain_prof[4], probe.n_fold);
    atomicAdd(&g_chain_prof[5], 1ull);
    atomicAdd(&g_chain_prof[6], (unsigned long long)n_cmds);
    atomicAdd(&g_chain_prof[7], probe.t_setup);
    atomicAdd(&g_chain_prof[8], probe.t_refill);
    atomicAdd(&g_chain_prof[9], probe.n_refill);
    atomicAdd(&g_chain_prof[10], (unsigned long long)n_searches);
  }
#endif
  if (tail != nullptr) {
    tail->n_cmds = n_cmds;
    tail->n_lits = n_lits;
    tail->ext_len = ext_len;
    tail->insert_len = insert_length;
    tail->last_dist_code = last_dist_code;
    tail->last_copy_len = last_copy_len;
  }
  next.pos = position;
  next.apply = apply;
  for (int i = 0; i < 4; ++i) next.cache[i] = dc[i];
  next.insert_len = 0;
  next.ext_allowed = 0;
  next.dict_lookups = ds.lookups;
  next.dict_matches = ds.matches;
  next.ext_max_distance = 0;
  next.dict_exact = entry.dict_exact;
  next.head_kind = position > seg.end ? tail_kind : (uint32_t)kHeadNone;
  next.head_base = position > seg.end ? tail_base : 0u;
  next.head_p1 = position > seg.end ? tail_p1 : 0u;
  next.pad = 0;
#if defined(BROTLI_HOST_EMU) && defined(BR_DEBUG_SPLICE)
  if constexpr (kSplice && kCheckpoints) {
    if (cp_on && rp != nullptr && (spliced || walked_from != entry.pos)) {
      // shadow: the same segment parsed from its entry to its end must give the same exit and the same commands
      std::vector<Command> got(cmds, cmds + n_cmds);
      const SegExit merged = exit_out;
      ChainTables t2 = t;
      t2.checkpoints = nullptr;
      SegExit sx{};
      SegEntry sn{};
      br_parse_segment<kH9, kRows, false, false>(P, t2, s, seg_in, entry, sx, sn);
      bool same = sx.pos == merged.pos && sx.apply == merged.apply && sx.insert_len == merged.insert_len && sx.n_cmds == merged.n_cmds && sx.n_lits == merged.n_lits &&
                  sx.n_searches == merged.n_searches && sx.last_dist_code == merged.last_dist_code && sx.last_copy_len == merged.last_copy_len &&
                  sx.n_pushes_all == merged.n_pushes_all && sx.tail_kind == merged.tail_kind && sx.tail_base == merged.tail_base && sx.tail_p1 == merged.tail_p1 &&
                  sx.ext_len == merged.ext_len && sx.dict_lookups == merged.dict_lookups && sx.dict_matches == merged.dict_matches && sx.dict_mode == merged.dict_mode &&
                  sx.dict_maxdef <= merged.dict_maxdef;
      for (int i = 0; i < 4; ++i) same = same && sx.cache[i] == merged.cache[i];
      bool cmds_same = sx.n_cmds == got.size();
      for (size_t i = 0; cmds_same && i < got.size(); ++i) cmds_same = memcmp(&got[i], &cmds[i], sizeof(Command)) == 0;
      if (!same || !cmds_same)
        fprintf(stderr, "SHADOW MISMATCH seg [%u,%u) mode %u rows %u..%u restart from %u spliced %d at %u: merged pos %u cmds %u lits %u searches %u pushes %u dict %u/%u mode %u | full pos %u cmds %u lits %u searches %u pushes %u dict %u/%u mode %u | cmds same %d\n",
                seg.start, seg.end, rp->mode, rp->rows_lo, rp->rows_hi, walked_from, (int)spliced, walked_to, merged.pos, merged.n_cmds, merged.n_lits, merged.n_searches,
                merged.n_pushes_all, merged.dict_lookups, merged.dict_matches, merged.dict_mode, sx.pos, sx.n_cmds, sx.n_lits, sx.n_searches, sx.n_pushes_all, sx.dict_lookups,
                sx.dict_matches, sx.dict_mode, (int)cmds_same);
      exit_out = merged;
      for (size_t i = 0; i < got.size(); ++i) cmds[i] = got[i];
    }
  }
#endif
  return (n_searches < 0x0fffffffu ? n_searches : 0x0fffffffu) | ((n_pushes < 4 ? n_pushes : 4u) << 28);  // what the segment cost | distances pushed
}

// Parses segment k and -- in list rounds (sched != nullptr) -- keeps going into the following segments of the
// same input block for as long as the state it arrives with differs from the entry their last parse used and
// they are not scheduled themselves: a change that would otherwise creep forward one segment per round (each
// round costing a full chain latency) is absorbed in one launch.  A continued segment gets its new entry
// written to entries[] so that the host resolver sees what it was parsed with.
template <bool kH9, bool kRows, bool kSplice = false, bool kDeep = false>
BR_DEV void br_parse_chain(const Lz77Params& P, const ChainTables& t, ChainScratchT<kH9, kRows, kDeep>& s, const Segment* segments,
                           SegEntry* entries, SegExit* exits, uint32_t k, uint8_t* sched, uint32_t max_continuation) {
  if constexpr (kRows) {
    if (BR_LANE < 32) s.dict_off[BR_LANE] = BR_LANE < 25 ? t.dict_offsets_by_length[BR_LANE] : 0u;
#if BR_SCALAR
    for (uint32_t i = 1; i < 32; ++i) s.dict_off[i] = i < 25 ? t.dict_offsets_by_length[i] : 0u;
#endif
    BR_SYNC();
  }
  SegEntry entry = entries[k];
  // a chain takes on at most max_continuation further segments' worth of searches (one search every other byte is
  // the going rate): what it leaves behind is picked up in the next round by a chain of its own, so that a launch with
  // many chains never lasts much longer than a handful of segment parses (launches with only a few chains are latency
  // bound anyway and let them run to the end of the block).  Counting searches rather than segments lets a chain run
  // through a long stretch of incompressible data -- one search every 9 or 17 bytes (literal spree, mod.rs:2529-2546),
  // and a state that no dry run can guess because it depends on where the stretch began -- in one launch.
  uint32_t budget = 0;
  bool first = true;
  // A segment entered because only the distance cache differed, whose parse then comes out as before (same pushes on
  // top of another inherited tail), merely passes the cache along.  The host resolver composes that through any number
  // of segments at once and has them verified side by side in the next round: the chain stops walking.  It keeps
  // walking where its arrival really changes parses (data whose parse hangs on the cache contents).
  bool watch = false, passes_along = false;
  for (;;) {
    const Segment seg = segments[k];
    SegEntry next;
    uint32_t ret;
    if constexpr (kSplice) {
      Reparse rp;
      rp.mode = 0;
      rp.rows_lo = 0xffffffffu;
      rp.rows_hi = 0;
      if (sched != nullptr && t.rows_changed_lo != nullptr) {
        rp.rows_lo = BR_UNIFORM(t.rows_changed_lo[k]);
        rp.rows_hi = BR_UNIFORM(t.rows_changed_hi[k]);
        // (the first segment: as the scheduler marked it; a segment walked into: other entry than last time)
        rp.mode = (first && BR_UNIFORM(sched[k]) == kSchedOwnRows) ? 2u : 1u;
      }
      ret = br_parse_segment<kH9, kRows, false, true>(P, t, s, seg, entry, exits[k], next, nullptr, nullptr, &rp);
    } else {
      ret = br_parse_segment<kH9, kRows>(P, t, s, seg, entry, exits[k], next);
    }
    const uint32_t cost = ret & 0x0fffffffu, np = ret >> 28;
    if (first) {
      const uint32_t per_segment = (seg.end - seg.start) / 2;
      budget = max_continuation > 0xffffffffu / (per_segment + 1) ? 0xffffffffu : max_continuation * per_segment;
      first = false;
    } else {
      budget = budget > cost ? budget - cost : 0;
    }
    if (watch) {
      bool unchanged = np < 4 && np == s.keep[2] && next.pos == s.keep[0] && next.apply == s.keep[1] && next.head_kind == s.keep[3] &&
                       next.head_base == s.keep[4] && next.head_p1 == s.keep[5];
      for (uint32_t i = 0; i < 4; ++i) unchanged = unchanged && (i >= np || (uint32_t)next.cache[i] == s.keep[6 + i]);
      passes_along = BR_UNIFORM(unchanged ? 1u : 0u) != 0;
    }
    if (!sched || (seg.flags & (kSegLastInBlock | kSegWarmup))) break;
    const uint32_t mark = sched[k + 1];
    if (mark == kSchedOwn || mark == kSchedOwnRows || mark == kSchedWalked) break;  // has its own chain in this launch
    if (budget == 0) break;
    const bool forced = mark == 2;     // left to this chain, must be redone whatever state we arrive with
    const SegEntry old = entries[k + 1];
    bool same = old.pos == next.pos && old.apply == next.apply && old.head_kind == next.head_kind && old.head_base == next.head_base &&
                old.head_p1 == next.head_p1;
    const bool rest_same = same;
    for (int i = 0; i < 4; ++i) same = same && old.cache[i] == next.cache[i];
    // The old parse of k + 1 found nothing to copy: the host does better than walking through it.  A changed position
    // it carries through a whole literal spree by arithmetic, giving every segment behind a chain of its own in the
    // next round (PredictLiteralRun); a changed distance cache it checks against all searched positions in parallel
    // (lz77_check_cache).
    if (!kH9 && !same && !forced && exits[k + 1].n_cmds == 0 && exits[k + 1].n_searches != 0 && exits[k + 1].ext_len == 0) break;
    if (same && P.use_dictionary && next.dict_exact) {
      // static-dictionary throttle (mod.rs:1957-1960): would the old parse of k + 1 have seen the dictionary in the
      // same state under the counters this chain arrives with?  (mirrors DictTracker::Consume on the host)
      const uint32_t mode = exits[k + 1].dict_mode;
      const int32_t maxdef = exits[k + 1].dict_maxdef;
      const bool dead = next.dict_matches < (next.dict_lookups >> 7);
      const bool no_match = exits[k + 1].dict_matches == old.dict_matches;  // none used (mode 1) / none possible (mode 2)
      const bool stays_alive = 128ll * (long long)next.dict_matches - (long long)next.dict_lookups + 127 >= (long long)maxdef;
      if (mode != 0) {
        if (mode == 4) {
          same = dead;  // parsed blind: only good for a dictionary that is off
        } else if (dead) {
          same = mode == 2 || (mode == 1 && no_match);
        } else if (old.dict_lookups == next.dict_lookups && old.dict_matches == next.dict_matches) {
          same = true;
        } else if (mode == 1) {
          same = stays_alive || no_match;
        } else {
          same = mode == 2 && no_match;
        }
      }
    }
    if (same && !forced) break;
    if (passes_along && !forced) break;
    watch = rest_same && !forced;
    BR_SYNC();
    if (watch && BR_LANE == 0) {
      const SegExit& ox = exits[k + 1];
      s.keep[0] = ox.pos;
      s.keep[1] = ox.apply;
      s.keep[2] = ox.n_pushes;
      s.keep[3] = ox.tail_kind;
      s.keep[4] = ox.tail_base;
      s.keep[5] = ox.tail_p1;
      for (int i = 0; i < 4; ++i) s.keep[6 + i] = (uint32_t)ox.cache[i];
    }
    ++k;
    if (BR_LANE == 0) {
      entries[k] = next;
      sched[k] = 3;
    }
    entry = next;
  }
}

// One live chain (lz77_live.h): parses the input blocks [first, last) -- live chains are cut one segment per block -- on
// private copy `table` of the bucket rings, which the launcher has materialised for the start of block `first`.  From
// the second block on the chain derives the entry by itself, the way Lz77Stage::Resolve does: the state it arrives
// with, the meta-block flush rule (encode.rs:2454-2477) and extend_last_command (encode.rs:2435-2437) on its own books
// (LiveBlockState).  What it cannot know -- a meta-block that ends up stored uncompressed hands the distance cache of
// its start to the next one (encode.rs:1994, 2142) -- shows as a wrong entry when the host resolver replays the exits,
// and the stream is parsed again from that block.  The entries used go to entries[], the books to live_state[].
template <bool kRows, bool kDeep = false>
BR_DEV void br_parse_live(const Lz77Params& P, const ChainTables& t, ChainScratchT<false, kRows, kDeep>& s, const Segment* segments, SegEntry* entries,
                          SegExit* exits, uint32_t first, uint32_t last, uint32_t table, uint32_t* histo /* 256 words of scratch */) {
  LiveRing lr;
  const size_t keys_per_table = (size_t)1 << P.bucket_bits;
  lr.num = t.live_num + (size_t)table * keys_per_table;
  lr.buckets = t.live_buckets + (((size_t)table * keys_per_table) << P.block_bits);
  lr.keys = t.keys;
  lr.bits = P.block_bits;
  SegEntry entry = entries[first];
  SegEntry next;
  LiveBlockState st = t.live_state[first];
  uint32_t carry = BR_UNIFORM(entry.insert_len);  // literals pending at the entry of the block
  int32_t saved_cache[4];  // the distance cache at the start of the open meta-block
  for (int i = 0; i < 4; ++i) saved_cache[i] = (int32_t)BR_UNIFORM(t.live_state[first].saved_cache[i]);
  const uint32_t max_mb = P.max_metablock_bytes, limit = max_mb / 8;
  for (uint32_t j = first; j < last; ++j) {
    const Segment seg = segments[j];
    BlockTail tail;
    br_parse_segment<false, kRows, true>(P, t, s, seg, entry, exits[j], next, &lr, &tail);
    if (j + 1 == last) break;
    if (P.reset_pos != 0 && BR_UNIFORM(seg.blk_end) == P.reset_pos) br_live_reset(lr, P.bucket_bits);  // the block that starts there searches an empty table
    if (seg.flags & kSegTailStitched) br_live_store(lr, seg.blk_end - 3u, 1, 3, 1, 0);  // StitchToPreviousBlock, mod.rs:210-222
    // ---- the books, as in Lz77Stage::Resolve
    if (tail.ext_len != 0 && st.last_valid) st.last_copy_len += tail.ext_len;
    st.mb_cmds += tail.n_cmds;
    st.mb_lits += tail.n_lits;
    if (tail.n_cmds != 0) {
      st.mb_lits += carry;
      carry = tail.insert_len;
      st.last_valid = 1;
      st.last_dist_code = tail.last_dist_code;
      st.last_copy_len = tail.last_copy_len;
    } else {
      carry += tail.insert_len;
    }
    const uint32_t be = BR_UNIFORM(seg.blk_end);
    const bool next_fits = (uint64_t)(be - st.mb_start) + P.block_bytes <= (uint64_t)max_mb;
    if (!(next_fits && st.mb_lits < limit && st.mb_cmds < limit)) {  // the meta-block is closed here
      // A meta-block that should_compress (encode.rs:1325-1354) sends out uncompressed hands the distance cache of its START
      // to the next one (encode.rs:1994): few commands, nearly all literals, and an every-13th-byte histogram whose f32
      // entropy says "incompressible".  (What the chain cannot know is the size fallback of encode.rs:2141-2163; the host
      // resolver finds the entry wrong then and sends the chain back.)
      const uint32_t bytes = be - st.mb_start;
      const uint32_t cmds_all = st.mb_cmds + (carry != 0 ? 1u : 0u), lits_all = st.mb_lits + carry;  // with the trailing insert-only command
      bool compress = true;
      if (cmds_all < (bytes >> 8) + 2 && (float)lits_all > 0.99f * (float)bytes) {
        BR_SYNC();
        for (uint32_t i = BR_LANE; i < 256; i += BR_NLANES) histo[i] = 0;
        BR_SYNC();
        for (uint32_t q = st.mb_start + 13u * (uint32_t)BR_LANE; q < be; q += 13u * BR_NLANES) BR_ATOMIC_INC(&histo[t.text[q]]);
        BR_SYNC();
        const float threshold = (float)bytes * 7.92f / 13.0f;
        compress = !(br_bits_entropy(t.logs, histo, 256) > threshold);
      }
      if (!compress)
        for (int i = 0; i < 4; ++i) next.cache[i] = saved_cache[i];
      for (int i = 0; i < 4; ++i) saved_cache[i] = next.cache[i];
      st.mb_start = be;
      st.mb_cmds = st.mb_lits = 0;
      st.last_valid = 0;
      carry = 0;  // (pending literals went into its trailing insert-only command)
    }
    // ---- the entry of the next block
    const uint32_t was_exact = entry.dict_exact;
    entry = next;
    entry.pos = be;
    entry.insert_len = carry;
    entry.dict_exact = was_exact;
    entry.head_kind = kHeadNone;
    entry.head_base = entry.head_p1 = 0;
    entry.ext_allowed = 0;
    if (st.mb_cmds != 0 && carry == 0 && st.last_valid) {
      const uint64_t cmd_dist = (uint64_t)(int64_t)next.cache[0];
      if (st.last_dist_code < 16 || (uint64_t)st.last_dist_code - 15 == cmd_dist) {
        const uint64_t lpp = (uint64_t)be - st.last_copy_len;
        const uint64_t max_distance = lpp < P.max_backward_limit ? lpp : P.max_backward_limit;
        if (cmd_dist <= max_distance) entry.ext_allowed = 1;
      }
    }
    BR_SYNC();
    if (BR_LANE == 0) {
      entries[j + 1] = entry;
      t.live_state[j + 1] = st;
    }
  }
}

// Repeats the cache and ring stages of the search a live chain ran at position p -- with the distance cache it had then
// (ChainTables::search_log) and the ring that the flags behind `ix` imply for that position -- and says whether they find
// what they found then (lz77_live_verify).
template <bool kRows, bool kDeep = false>
BR_DEV bool br_verify_search(const Lz77Params& P, const ChainTables& t, const LiveIndex& ix, ChainScratchT<false, kRows, kDeep>& s, uint32_t p,
                             uint32_t blk_end) {
  const uint32_t* rec = t.search_log + (size_t)p * kSearchLogWords;
  BR_SYNC();
  int32_t* dc = s.dc;
  for (int i = 0; i < 4; ++i) dc[i] = (int32_t)BR_UNIFORM(rec[i]);
  for (int i = 4; i < 16; ++i) dc[i] = 0;
  br_prepare_distance_cache(dc, P.ndist);
  ProbeMeta m;
  m.pos = p;
  m.version = 0;
  m.win_base = 0xffffff00u;
  m.no_dict = 1;  // (the dictionary stage comes after what is compared here)
  m.log_on = 0;
  m.g[0] = m.g[1] = 0;
  DictState ds;
  ds.lookups = ds.lookups0 = ds.matches = ds.matches0 = 0;
  ds.mode = 0;
  ds.maxdef = 0;
  ds.vlookups = 0;
  ds.vwould = 0;
  ds.vmaxdef = 0;
  const uint32_t depth = 1u << P.block_bits;
  const uint32_t key = BR_UNIFORM(t.keys[p]);
  const LiveRingAt ring = br_live_ring_at_slot(ix, key, p, BR_UNIFORM(ix.slot_of[p]), depth);
  const uint32_t ndist = P.ndist;
  const uint32_t max_length = blk_end - p;
  const uint32_t max_backward = p < P.max_backward_limit ? p : P.max_backward_limit;
  const uint8_t* cur_data = t.text + p;
  BR_SYNC();
#if !BR_SCALAR
  if constexpr (kRows) {
    // the lane layout of br_probe_pair_rows, first half only
    const uint32_t lane = (uint32_t)BR_LANE, c = lane & 31u;
    const bool half0 = lane < 32u;
    const bool is_cache = half0 && c < ndist;
    const uint32_t i = c - ndist;
    const bool is_ring = half0 && !is_cache && i < ring.visible;
    uint32_t e = kLiveBreak;
    if (is_ring) e = br_live_ring_entry(ix, ring, i);
    const bool brk = is_ring && (e >= kLiveBreak || p - e > max_backward);
    const uint32_t brk_half = (uint32_t)__ballot(brk);
    const uint32_t first_brk = brk_half ? (uint32_t)__ffs((int)brk_half) - 1u : 32u;
    uint32_t prev = 0xffffffffu;
    if (is_cache) {
      const int64_t b = (int64_t)dc[c];
      if (b > 0 && b <= (int64_t)max_backward) prev = p - (uint32_t)b;
    } else if (is_ring && c < first_brk) {
      prev = e;
    }
    m.r_len = prev != 0xffffffffu ? br_match_len_wide(t.text + prev, cur_data, max_length, t.run_end, prev, p) : 0u;
    m.r_prev = prev;
    m.nbucket[0] = m.nbucket[1] = kRowEntries;
  } else
#endif
  {
    // candidates through LDS, like br_probe_pair
    m.nbucket[0] = ring.visible;
    m.nbucket[1] = 0;
    const uint32_t total = ndist + ring.visible;
    for (uint32_t c = BR_LANE; c < total; c += BR_NLANES) {
      uint32_t prev = 0xffffffffu;
      if (c < ndist) {
        const int64_t b = (int64_t)dc[c];
        if (b > 0 && b <= (int64_t)max_backward) prev = p - (uint32_t)b;
      } else {
        const uint32_t q = br_live_ring_entry(ix, ring, c - ndist);
        if (q < kLiveBreak && p - q <= max_backward) prev = q;  // else: marks the point where the bucket walk breaks
      }
      s.cand_prev[0][c] = prev;
      s.cand_len[0][c] = prev != 0xffffffffu ? br_match_len_wide(t.text + prev, cur_data, max_length, t.run_end, prev, p) : 0u;
    }
    BR_SYNC();
  }
  SearchResult now;
  br_fold_probe<false, kRows>(P, t, s, m, 0, ds, blk_end, &now, true);
  return br_same_as_logged(rec, now);
}

// ---- bursts (device_api.h): what the device decides between two list launches, per segment --------------------------------
// Segment k was parsed in the launch just finished (own chain: sched 1; walked into: sched 3): does the state its exit hands
// to segment k + 1 (same input block) equal the entry k + 1 was last parsed with?  The plain chaining rule of
// Lz77Stage::Resolve -- position, spree countdown, distance cache, the step that reaches into k + 1; the dictionary
// counters and the books stay what k + 1's entry holds (the host judges those).
BR_DEV void br_chain_check(const Segment* segments, const SegEntry* entries, const SegExit* exits, uint32_t num_segments, uint32_t k,
                           const uint8_t* sched, uint8_t* touched, uint8_t* entry_dirty, SegEntry* new_entries, uint32_t* rows_changed_lo,
                           uint32_t* rows_changed_hi, uint8_t* stale = nullptr) {
  if (sched[k] != kSchedOwn && sched[k] != kSchedOwnRows && sched[k] != kSchedWalked) return;
  touched[k] = 1;
  if (stale != nullptr) stale[k] = 1;
  if (rows_changed_lo != nullptr) {  // parsed: what is marked from now on happened after this parse
    rows_changed_lo[k] = 0xffffffffu;
    rows_changed_hi[k] = 0;
  }
  if (k + 1 >= num_segments || (segments[k + 1].flags & kSegFirstInBlock)) return;
  const SegExit& x = exits[k];
  const SegEntry& u = entries[k + 1];
  bool same = u.pos == x.pos && u.apply == x.apply && u.head_kind == x.tail_kind && u.head_base == x.tail_base && u.head_p1 == x.tail_p1;
  for (int i = 0; i < 4; ++i) same = same && u.cache[i] == x.cache[i];
  if (same) return;
  SegEntry n = u;
  n.pos = x.pos;
  n.apply = x.apply;
  for (int i = 0; i < 4; ++i) n.cache[i] = x.cache[i];
  n.head_kind = x.tail_kind;
  n.head_base = x.tail_base;
  n.head_p1 = x.tail_p1;
  n.ext_allowed = 0;
  new_entries[k + 1] = n;
  entry_dirty[k + 1] = 1;
}
// drops the checkpoint records of one segment
BR_DEV void br_drop_checkpoints(Checkpoint* checkpoints, const Segment& seg) {
  for (uint32_t b = (seg.start / kCheckpointStride + 1u) * kCheckpointStride; b < seg.end; b += kCheckpointStride) checkpoints[b / kCheckpointStride].valid = 0;
}
// Does segment k go into the next launch?  Takes its new entry if it has one and clears its marks.
BR_DEV bool br_burst_schedule_one(SegEntry* entries, uint32_t k, uint8_t* sched, uint8_t* cand_dirty, uint8_t* entry_dirty, const SegEntry* new_entries) {
  const bool go = cand_dirty[k] != 0 || entry_dirty[k] != 0 || sched[k] == kSchedLeft;
  // (a segment that only had rows change keeps its entry: its chain may restart from a checkpoint, see Reparse)
  const uint8_t how = (entry_dirty[k] != 0 || sched[k] == kSchedLeft) ? kSchedOwn : kSchedOwnRows;
  if (entry_dirty[k]) entries[k] = new_entries[k];
  cand_dirty[k] = 0;
  entry_dirty[k] = 0;
  sched[k] = go ? how : 0;
  return go;
}

}  // namespace brotli_mi355x
#endif
