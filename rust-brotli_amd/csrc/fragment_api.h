// fragment_api.h -- device seam of qualities 0 and 1: compress_fragment (one pass, quality 0) and
// compress_fragment_two_pass (quality 1), the encoders BrotliEncoderCompressStream runs instead of the ring-buffer path
// (encode.rs:2706-2861).  Same conventions as device_api.h.
//
// Round 5: the fragments of one call run SIDE BY SIDE.  Every fragment starts on a hash table zeroed for it
// (encode.rs:1643-1700), so what one fragment hands the next is small: the bit position (three bits of phase, which only the
// byte alignment of a stored meta-block ever looks at) and, at quality 0, the command prefix code -- which a fragment rebuilds from
// the commands of its own LAST block (compress_fragment.rs:1033-1044), whatever code it came in with.  So: every fragment is
// compressed into a slot of its own from bit 0, quality 0 in two passes (pass A yields the code each fragment leaves behind,
// pass B compresses with the right incoming codes), and the slots are joined bit-exactly afterwards (frag_join).
#ifndef BROTLI_MI355X_FRAGMENT_API_H_
#define BROTLI_MI355X_FRAGMENT_API_H_

#include <stddef.h>
#include <stdint.h>

namespace brotli_mi355x {

// What a stream carries from one fragment to the next at quality 0: the command prefix code as the last fragment left it
// (encode.rs:172-176, 627-659).  (The bit position travels in FragmentJob / FragmentResult.)
struct FragmentState {
  uint32_t cmd_code_numbits;
  uint32_t pad;
  uint8_t cmd_depths[128];
  uint16_t cmd_bits[128];
  uint8_t cmd_code[512];
};

// One fragment of a batch.
struct FragmentJob {
  uint32_t in_offset;   // where the fragment starts in the batch's input
  uint32_t in_size;
  uint32_t is_last;     // the stream ends with this fragment (ISLAST + ISLASTEMPTY behind it)
  uint32_t table_bits;  // as HashTableSize chose them
  uint64_t out_offset;  // byte offset of the fragment's output slot (a multiple of 8; >= 2 * in_size + 520 bytes, first byte zero)
  uint32_t start_bits;  // 0..7: the bit phase the fragment starts with inside its slot
  uint32_t state_in;    // index into states_in of the command code it comes in with (quality 0)
};

// What a fragment reports.  Bit positions count from the start of the slot (start_bits included).
struct FragmentResult {
  uint64_t end_bits;        // where the fragment's bits end
  uint64_t first_align;     // position of the first jump to a byte boundary (before the padding), ~0 = none: everything behind
                            // it is byte aligned whatever the phase, everything in front of it moves with the phase
  uint64_t decision_bits;   // bits the fragment had produced when "larger than stored raw?" was asked (the one decision that
                            // counts output bits, compress_fragment_two_pass.rs:697-702 / compress_fragment.rs:1166-1171) ...
  uint64_t decision_align;  // ... and the first alignment point as of then (its padding is what the phase changes)
  uint32_t fell_back;       // the answer was yes: the whole fragment is one stored meta-block
  uint32_t bad;             // never expected
};

// Scratch of a batch in device memory: per fragment a hash table (zeroed by frag_compress_batch), and for quality 1 the command
// and literal buffers of one 128 KiB block (compress_fragment_two_pass.rs:646-703).  Fragment j uses [j * stride, (j + 1) * stride) of each.
struct FragmentBuffers {
  uint32_t* table = nullptr;     // n x table_stride words, table_stride >= 1 << (largest table_bits of the batch)
  uint32_t* commands = nullptr;  // n x cmd_stride words (quality 1), cmd_stride >= min(largest fragment, 1 << 17)
  uint8_t* literals = nullptr;   // n x lit_stride bytes (quality 1), lit_stride >= min(largest fragment, 1 << 17) + 64
  size_t table_stride = 0, cmd_stride = 0, lit_stride = 0;
};

// n fragments side by side, one wavefront each: input[job.in_offset, + in_size) (device memory, >= 64 readable bytes behind the
// last one) is compressed as the reference compresses one fragment and its bits go to out + job.out_offset from bit
// job.start_bits on.  states_in / states_out: quality 0 only (states_out[j] = the code fragment j leaves behind; may alias nothing).
void frag_compress_batch(int quality, const uint8_t* input, const FragmentJob* jobs_dev, uint32_t n, const FragmentBuffers& B,
                         const FragmentState* states_in_dev, FragmentState* states_out_dev, FragmentResult* results_dev, uint8_t* out);

// dst (zeroed) |= bits [src_bit, src_bit + nbits) of src, placed at dst_bit, for every piece.  Pieces do not overlap in dst.
struct FragmentPiece {
  uint64_t src_bit, dst_bit, nbits;
};
void frag_join(const uint8_t* src, const FragmentPiece* pieces_dev, uint32_t n, uint8_t* dst);

}  // namespace brotli_mi355x
#endif
