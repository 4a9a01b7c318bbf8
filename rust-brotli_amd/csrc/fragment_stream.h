// fragment_stream.h -- host side of qualities 0 and 1: what BrotliEncoderCompressStream does instead of the ring-buffer path
// (BrotliEncoderCompressStreamFast, encode.rs:2706-2861) -- the input of every call is cut into fragments of at most 1 << lgwin
// bytes, each compressed on its own hash table by the device (fragment_api.h); the stream carries the open byte of the output and,
// at quality 0, the command prefix code from fragment to fragment.
#ifndef BROTLI_MI355X_FRAGMENT_STREAM_H_
#define BROTLI_MI355X_FRAGMENT_STREAM_H_

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "encoder_params.h"
#include "fragment_api.h"

namespace brotli_mi355x {

struct FragmentStream {
  bool started = false;
  uint16_t last_bytes = 0;      // the open byte(s) of the output: window bits at first, then what the last piece left
  uint8_t last_bytes_bits = 0;
  FragmentState state{};        // quality 0: the command prefix code, host copy between calls
  // catable streams only (FragmentRingCompress): what the reference keeps in its ring buffer and books
  std::vector<uint8_t> pending;  // input copied in, not yet compressed (less than a block, or the block that has just filled up)
  uint32_t first_mb = 0;         // is_first_mb, encode.rs:2261-2333: 0 nothing written, 1 magic-number block, 2 one raw byte, 3 both
  size_t size_hint = 0;          // update_size_hint at the first encode_data (the magic-number block carries it)
  bool saw_input = false;
  uint64_t input_seen = 0;       // input_pos_: bytes copied in so far
  uint64_t flushed_raw = 0;      // last_flush_pos_: the quality 0 / 1 branch of encode_data moves it by the raw first bytes only (encode.rs:2283-2389)
};

// true for the parameter sets that take this path in the reference: quality 0 / 1 and not catable (encode.rs:2929-2937)
bool IsFragmentStream(const EncoderParams& user_params);
// quality 0 / 1 AND catable (which a custom dictionary and the shards of compress_multi turn on at these qualities): the stream goes
// through the reference's ring-buffer path after all, where encode_data hands every input block of 1 << lgwin bytes to the same
// fragment compressors (encode.rs:2335-2389) -- behind the two raw first bytes of a catable stream and the magic-number block
bool IsFragmentRing(const EncoderParams& user_params);
// One BrotliEncoderCompressStream call on such a stream (the generic loop of compress_stream, encode.rs:2938-2995).
void FragmentRingCompress(const EncoderParams& user_params, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, bool flush,
                          std::vector<uint8_t>* out);
// One BrotliEncoderCompressStream call with `size` bytes of input: finish = BROTLI_OPERATION_FINISH, flush = BROTLI_OPERATION_FLUSH
// (the byte-alignment block behind the data, encode.rs:1541-1566), neither = PROCESS.  Whole bytes of output are appended to *out.
void FragmentStreamCompress(const EncoderParams& user_params, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, bool flush,
                            std::vector<uint8_t>* out);
// BROTLI_OPERATION_EMIT_METADATA on a catable stream of these qualities: process_metadata (encode.rs:2579-2685) has encode_data flush
// "pending" input until input_pos_ == last_flush_pos_ -- which the quality 0 / 1 branch moves by the raw first bytes only.  True if
// the reference returns from that loop: everything received so far is (or, after the forced flush, will be) raw first bytes.
bool FragmentRingMetadataReturns(const FragmentStream& fs);
// BROTLI_OPERATION_EMIT_METADATA on such a stream: the header of a metadata block of `size` bytes behind the open byte
// (write_metadata_header, encode.rs:2545-2575); the caller appends the payload
void FragmentStreamMetadataHeader(const EncoderParams& user_params, FragmentStream* fs, size_t size, std::vector<uint8_t>* out);

}  // namespace brotli_mi355x
#endif
