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

}  // namespace brotli_mi355x
#endif
