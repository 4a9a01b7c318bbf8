// quick_api.h -- device seam of qualities 2..4: the BasicHasher family H2 / H3 / H4 / H54
// (backward_references/mod.rs:237-596) under CreateBackwardReferences (mod.rs:2376-2552), block by block.
// Same conventions as device_api.h: the product implements these with HIP kernels (quick_kernels.hip), the tests link the
// host driver against the serial emulation in tests/emu.
#ifndef BROTLI_MI355X_QUICK_API_H_
#define BROTLI_MI355X_QUICK_API_H_

#include <stddef.h>
#include <stdint.h>

#include "device_api.h"

namespace brotli_mi355x {

// The hasher of one stream.  A BasicHasher is ONE table of text positions: a position goes into slot key + ((ix >> 3) % sweep)
// (mod.rs:322-327), a search looks at the `sweep` slots key .. key + sweep - 1 (mod.rs:391-440).  The table lives in device
// memory from the first block of a stream to its last; behind the slots sit the two counters of the static-dictionary
// throttle (dict_num_lookups / dict_num_matches, mod.rs:1957-1960), so that one copy takes a snapshot of the whole hasher.
struct QuickJob {
  uint32_t kind = 2;         // hasher type: 2, 3, 4 or 54 (encode.rs:834-893)
  uint32_t bucket_bits = 16; // 16 (H2, H3), 17 (H4), 20 (H54)                  mod.rs:437-560
  uint32_t sweep = 1;        // BUCKET_SWEEP: 1, 2, 4, 4
  uint32_t hash_len = 5;     // bytes hashed: 5, 7 for H54
  uint32_t use_dictionary = 0;  // params.use_dictionary and a hasher that consults it (H2, H4)
  uint32_t* table = nullptr; // [quick_table_words()]: slots, then the books
};
constexpr uint32_t quick_slots(const QuickJob& J) { return (1u << J.bucket_bits) + J.sweep; }
constexpr uint32_t quick_books_at(const QuickJob& J) { return (quick_slots(J) + 15u) & ~15u; }  // [0] lookups, [1] matches
constexpr uint32_t quick_table_words(const QuickJob& J) { return quick_books_at(J) + 16u; }

// a fresh hasher: every slot 0 (the reference allocates zeroed tables and relies on it, encode.rs:1147), books 0
void lz77_quick_init(const QuickJob& J);
// the hasher of the piece in front (table_src, as QuickJob::table) becomes this job's: copied, every position moved down by
// `delta` (this piece's text starts that much further into the stream, a multiple of the ring-buffer size); entries in front
// of the new text lie beyond every window and become 0, which no position of the new text can reach either
void lz77_quick_import(const QuickJob& J, const uint32_t* table_src, uint32_t delta);
// HasherPrependCustomDictionary (encode.rs:1163-1194, mod.rs:224-229): positions [0, dict_bytes - 7) of the text are stored
void lz77_quick_prepend(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, uint32_t dict_bytes);
// input block `block` (= segment `block`: one segment per block) with the entry B.entries[block]: StitchToPreviousBlock,
// extend_last_command, CreateBackwardReferences; commands into its slab, exit into B.exits[block].  In stream order.
void lz77_quick_block(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, uint32_t block);

// ---- the speculative path of these qualities (quick_spec.h, round 6): segments of a block side by side ----------------------
// What a BasicHasher slot holds when position p is searched is a function of the text and of WHICH positions were filed before p
// (and, for a quad of StoreRange, under which of two slot offsets): the value of the latest filing into that slot.  Every
// filing goes by position order (a search files its own position after it has looked, a copy files its range before the next
// search).  So, given a flag byte per position -- filed / searched / filed by a quad and as which of its four -- the `sweep`
// candidates of every position are one sort away, like the candidate rows of quality 5: the potential filings ("events": one
// per position under its own slot offset, for sweep > 1 a second one under the offset a quad that starts in the 8-byte group in
// front would give it) are sorted by (slot, position) once; per round the active ones are marked from the flags, an
// exclusive max-scan gives every event the latest active event in front of it, and the candidate of (p, j) is that of p's rank in
// slot key(p) + j.  Chains (one wavefront per segment, quick_spec.h) parse with these candidates and write flags; the host
// resolver chains their exits (Lz77Stage::Resolve, as for qualities 5-9); a segment is parsed again when its entry changed or a
// candidate of a position it searched did.  At the fixed point the parse IS the sequential one.
// Flag byte of a position (QuickSpec::flags):
static constexpr uint8_t kQsStored = 1, kQsSearched = 2, kQsQuad = 4;  // bits 3-4: index of the position in its quad
struct QuickSpec {
  uint32_t n = 0;           // text positions (P.total_bytes)
  uint32_t events = 0;      // n (sweep 1) or 2 n
  uint32_t slots = 0;       // quick_slots(J)
  uint32_t* ev_slot = nullptr;   // [events] slot of event i, ascending
  uint32_t* ev_id = nullptr;     // [events] its event id: position (sweep 1) or 2 * position + (1 if the displaced one)
  uint32_t* ev_of = nullptr;     // [events] inverse: index of event id
  uint32_t* slot_first = nullptr;  // [slots + 2] first event of every slot
  uint32_t* qrank = nullptr;     // [n * sweep] index of the first event of slot key(p) + j whose position is >= p
  uint32_t* actraw = nullptr;    // [events] index + 1 of the active events, 0 for the others (from the flags)
  uint32_t* act = nullptr;       // [events] actraw max-scanned (inclusive): the latest active event at or in front of every event
  uint32_t* val = nullptr;       // [events] what an active event filed
  uint8_t* flags_prev = nullptr; // [n + 64] the flags actraw / act / cand stand for
  uint32_t* chg_list = nullptr;  // [chg_cap] incremental update: the events whose activity or value changed
  uint32_t* chg_range = nullptr; // [3 * chg_cap] per listed event: its slot and the positions (lo, hi] whose rank in it lies behind a changed scan entry
  uint32_t* chg_count = nullptr; // [16] [0] changed events (may exceed chg_cap: then the list is useless), [1] a walk gave up
  uint32_t chg_cap = 0;
  uint32_t* cand = nullptr;      // [n * sweep] candidates of every position
  uint8_t* flags = nullptr;      // [n + 64]
  const uint32_t* base = nullptr;  // [slots] what the table held in front of the text (a later piece of a stream); null: zeroed
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  uint32_t* sort_keys_tmp = nullptr;  // [events] scratch of the sort (keys in, ids in)
  uint32_t* sort_ids_tmp = nullptr;
  uint32_t* scan_tmp = nullptr;  // [events / 1024 + 4096]
};
size_t lz77_qspec_sort_tmp_bytes(uint32_t events);
// once per text: events, their order, qrank
void lz77_qspec_index(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S);
// first guess of the flags: a custom dictionary is filed position by position but for its last 7 bytes, the last three positions of
// a block in front of a block of >= 7 bytes are filed when that block starts (StitchToPreviousBlock), the last 7 of a block not
// otherwise; everything else is assumed filed by a search
// (prefix_is_dictionary = false: the prefix is the stream so far, whose filings are in S.base)
void lz77_qspec_init_flags(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t first_block_start, bool prefix_is_dictionary);
// cand from the flags, every position (flags_prev := flags); geo != nullptr: dirty[k] = 1 for the chains that searched a position one of
// whose candidates changed in a way that can matter (qs_change_matters)
void lz77_qspec_candidates(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const SegGeometry* geo, uint8_t* dirty_dev);
// The same after a launch over a few segments, in two steps, with work in proportion to the changes instead of the text.
// lz77_qspec_diff: the flags of the listed segments against flags_prev -- the events whose activity or value changed are switched
// and listed (S.chg_count[0], S.chg_list), flags_prev brought up to date.  lz77_qspec_repair(count): for every listed event the scan
// entries from it up to the next active event of its slot are rewritten (from actraw alone: repairs that overlap write the same
// values), then the candidates of the positions whose rank in that slot lies in the rewritten stretch -- found in the 2 sweep - 1
// slots that hold the own-offset events of positions looking into it -- are derived again and compared.  A walk over more than
// kQsWalkCap inactive events gives up (S.chg_count[1]): the caller takes the pass over everything then, as when the list overflowed.
void lz77_qspec_diff(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list_dev, uint32_t count);
void lz77_qspec_repair(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t changed, const SegGeometry& geo, uint8_t* dirty_dev);
static constexpr uint32_t kQsWalkCap = 1u << 14;
// the listed segments (list_dev == nullptr: all of them), each from B.entries[k]: flags, commands, B.exits[k]
// own_tables != nullptr: chain i of the launch reads and files into the table own_tables + i * own_stride instead of looking at the
// candidates (QsTables::own, quick_spec.h; one chain per block)
void lz77_qspec_parse(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list_dev, uint32_t count,
                      uint32_t* own_tables = nullptr, uint32_t own_stride = 0);
// tables[i * stride + slot], i < count: what the slot holds where segment list[i] (list_dev == nullptr: segment i) starts, under the
// flags the candidates stand for (lz77_qspec_table with upto = the segment's start, for many segments at once)
void lz77_qspec_block_tables(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const uint32_t* list_dev, uint32_t count,
                             uint32_t* tables, uint32_t stride);
// *out_dev = the lowest position of the listed segments whose FILING differs between S.flags and S.flags_prev (the flags the
// candidates and the block tables stand for), 0xffffffff if there is none
void lz77_qspec_first_change(const Lz77Buffers& B, const QuickSpec& S, const uint32_t* list_dev, uint32_t count, uint32_t* out_dev);
// the same chains over `count` segments and entries of the caller's (device arrays), exits to exits_dev: the dry runs of the warm-up
// (Segment::flags with kSegWarmup: nothing but the exit is written)
void lz77_qspec_parse_custom(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, const Segment* segments_dev, const SegEntry* entries_dev,
                             SegExit* exits_dev, uint32_t count);
// out[i] = B.exits[list[i]] for i < count (list rounds move only what was parsed)
void lz77_qspec_gather_exits(const Lz77Buffers& B, const uint32_t* list_dev, uint32_t count, SegExit* out_dev);
// the slots of the hasher as the reference holds them when it has filed every position in front of `upto` that it files (from the
// flags the candidates stand for; the books of the throttle are the caller's): out[slots].  out may be S.base.
void lz77_qspec_table(const Lz77Params& P, const Lz77Buffers& B, const QuickJob& J, const QuickSpec& S, uint32_t upto, uint32_t* out);

}  // namespace brotli_mi355x
#endif
