// fragment_api.h -- device seam of qualities 0 and 1: compress_fragment (one pass, quality 0) and
// compress_fragment_two_pass (quality 1), the encoders BrotliEncoderCompressStream runs instead of the ring-buffer path
// (encode.rs:2706-2861).  Same conventions as device_api.h.
#ifndef BROTLI_MI355X_FRAGMENT_API_H_
#define BROTLI_MI355X_FRAGMENT_API_H_

#include <stddef.h>
#include <stdint.h>

namespace brotli_mi355x {

// What a stream carries from one fragment to the next (device memory while a call runs, host memory between calls).
struct FragmentState {
  uint64_t storage_ix;        // bit position in the output buffer of the running call
  uint32_t bad;               // set by the device code if it meets a state it cannot continue from (never expected)
  uint32_t cmd_code_numbits;  // quality 0: the command prefix code as the last fragment left it (encode.rs:172-176, 627-659)
  uint8_t cmd_depths[128];
  uint16_t cmd_bits[128];
  uint8_t cmd_code[512];
};

// Scratch of one stream in device memory (owned by the caller): the hash table (zeroed for every fragment, encode.rs:1643-1700)
// and, for quality 1, the command and literal buffers of one 128 KiB block (compress_fragment_two_pass.rs:646-703).
struct FragmentBuffers {
  uint32_t* table = nullptr;     // [1 << 17]
  uint32_t* commands = nullptr;  // [1 << 17]
  uint8_t* literals = nullptr;   // [1 << 17]
  FragmentState* state = nullptr;
};

// One fragment: input[0, input_size) (device memory, >= 64 readable bytes behind it) is compressed as the reference compresses one
// fragment -- table_bits as HashTableSize chose them -- and its bits are appended to `out` (device memory, >= 2 * input_size + 503
// bytes behind the current position) at state->storage_ix, which is advanced.  The byte under the cursor holds the bits written so
// far (the caller places the stream's open byte there before the first fragment of a call).  Fragments of a stream in order.
void frag_compress(int quality, const uint8_t* input, uint32_t input_size, bool is_last, uint32_t table_bits, const FragmentBuffers& B, uint8_t* out);

}  // namespace brotli_mi355x
#endif
