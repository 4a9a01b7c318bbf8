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
  FragmentState state{};        // host copy between calls (storage_ix is per call)
};

// true for the parameter sets that take this path in the reference: quality 0 / 1 and not catable (encode.rs:2929-2937)
bool IsFragmentStream(const EncoderParams& user_params);
// One BrotliEncoderCompressStream call with `size` bytes of input: finish = BROTLI_OPERATION_FINISH, flush = BROTLI_OPERATION_FLUSH
// (the byte-alignment block behind the data, encode.rs:1541-1566), neither = PROCESS.  Whole bytes of output are appended to *out.
void FragmentStreamCompress(const EncoderParams& user_params, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, bool flush,
                            std::vector<uint8_t>* out);
// BROTLI_OPERATION_EMIT_METADATA on such a stream: the header of a metadata block of `size` bytes behind the open byte
// (write_metadata_header, encode.rs:2545-2575); the caller appends the payload
void FragmentStreamMetadataHeader(const EncoderParams& user_params, FragmentStream* fs, size_t size, std::vector<uint8_t>* out);

}  // namespace brotli_mi355x
#endif
