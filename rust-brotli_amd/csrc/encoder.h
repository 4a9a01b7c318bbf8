// encoder.h -- one-stream encoder: host driver that turns (params, optional LZ77 prefix, input) into a
// brotli stream using the device stages (lz77_stage + meta-block kernels).
//
// It plays the role of the reference's encode_data / WriteMetaBlockInternal loop
// (src/enc/encode.rs:1941-2167, 2214-2543) for a COMPLETE input: the stream API of the C ABI buffers
// its input and calls this once at FINISH.
#ifndef BROTLI_MI355X_ENCODER_H_
#define BROTLI_MI355X_ENCODER_H_

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "encoder_params.h"
#include "lz77_stage.h"

namespace brotli_mi355x {

struct EncodeStats {
  uint32_t lz77_rounds = 0;
  uint64_t searches = 0, commands = 0, literals = 0;
  uint32_t metablocks = 0, uncompressed_metablocks = 0;
  uint32_t fallback_retries = 0;
  double ms_lz77 = 0, ms_metablock = 0, ms_total = 0;
  double ms_phase[16] = {0};
  // dominant kernel (k_parse_segments) timed with HIP events on its stream
  double parse_kernel_ms = 0;
  uint32_t parse_launches = 0;
  uint64_t parse_segments = 0;
  uint32_t num_segments = 0, segment_bytes = 0;
  // what those launches did, all chains and re-parses included: positions walked, searches, commands written
  uint64_t parse_walked = 0, parse_searches = 0, parse_commands = 0;
};

struct EncodeRequest {
  EncoderParams params;         // as set by the user (not yet finalized)
  const uint8_t* input = nullptr;
  size_t input_size = 0;
  bool input_on_device = false;  // input points to device memory (bench path: data resident in HBM)
  const uint8_t* prefix = nullptr;  // custom LZ77 dictionary (already clipped to the last (1<<lgwin)-16 bytes)
  size_t prefix_size = 0;
  bool prefix_is_file_continuation = false;  // compress_multi semantics: prev bytes come from the prefix
  bool hasher_chosen_before_size_hint = false;  // custom dictionary path picks the hasher early ...
  bool has_hasher_size_hint = false;            // ... with the size hint as it stood at that moment (else: params.size_hint)
  size_t hasher_size_hint = 0;
  uint32_t segment_bytes = 0;  // bytes per parse chain; 0 = chosen from the input size (ChooseSegmentBytes)
  // optional: the stream is written straight into this buffer instead of `out` (one device-to-host copy, no
  // intermediate vector); too small a buffer is an error
  uint8_t* direct_out = nullptr;
  size_t direct_capacity = 0;
  size_t* direct_size = nullptr;
  bool direct_out_on_device = false;  // direct_out is device memory: the stream stays in HBM (multi-GPU gather)
  // BROTLI_OPERATION_FLUSH support: `prefix` is the stream encoded so far, *carry_in its state (may be !valid for the
  // first piece); finish = false leaves the stream open (no ISLAST, padded to a byte boundary like the reference's
  // injected flush, encode.rs:1541-1566); the state for the next piece is written to *carry_out
  const StreamCarry* carry_in = nullptr;
  StreamCarry* carry_out = nullptr;
  bool finish = true;
  // Bounded-memory streaming (BROTLI_OPERATION_PROCESS with a lot of input buffered): with finish = false and partial =
  // true the piece is NOT flushed.  Only the meta-blocks that the reference's flush rule closes by itself within the
  // input are emitted (whole input blocks must be handed over); *consumed_out tells how much of the input they cover --
  // the rest has to be offered again, in front of more input.  The output ends on a bit, not a byte boundary: the
  // incomplete last byte travels in the carry.  *keep_from_out: how many leading bytes of prefix + input the caller may
  // drop -- the next piece's prefix starts there (StreamCarry::stream_base).  Both also work for flushed pieces.
  bool partial = false;
  bool last_block_processed_early = false;  // see Lz77Stage::SetStreamState
  size_t* consumed_out = nullptr;
  size_t* keep_from_out = nullptr;
  // BROTLI_OPERATION_EMIT_METADATA (with finish = false): instead of the flush padding, the header of a metadata block
  // of metadata_size bytes is written (encode.rs:2545-2575); the caller appends the bytes themselves
  bool emit_metadata = false;
  size_t metadata_size = 0;
};

// Compresses one stream.  Output is appended to `out`.  Throws std::runtime_error on device errors or
// unsupported parameters (there is no CPU fallback).
// Segment size of the speculative parse: short segments give low latency (few, short chains finish quickly), long ones
// less per-segment overhead (warm-up, resolver work) on big inputs.
uint32_t ChooseSegmentBytes(size_t input_bytes);
// stream position of the reference's first hasher reset (encode.rs:1623-1631): 3 GiB
uint64_t FirstPositionWrap();

void EncodeStream(const EncodeRequest& req, std::vector<uint8_t>* out, EncodeStats* stats);

}  // namespace brotli_mi355x
#endif
