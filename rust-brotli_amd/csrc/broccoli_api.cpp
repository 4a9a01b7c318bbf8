// broccoli_api.cpp -- the BroCatli streaming C API of the reference (c/brotli/broccoli.h, src/ffi/broccoli.rs:55-175):
// concatenation of brotli streams that were encoded `catable` / `appendable`, fed and drained piecewise.
//
// The stitching rules are those of csrc/concat.cpp (ChunkStitcher = src/concat/mod.rs:274-608 restated).  Like the
// reference's state machine this front end keeps nothing of a file but the bytes of its header until they can be judged and
// its last two bytes: input is taken in slices of at most 64 KiB, and only while the stitched bytes of the slice before
// have been handed out -- memory stays constant whatever the sizes of the files and of the caller's buffers.
// Host-only code: concatenation touches two bytes per junction, there is nothing in it for a GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/brotli_mi355x.h"
#include "concat.h"

using namespace brotli_mi355x;

namespace {
constexpr size_t kSlice = 64 << 10;
struct Broccoli {
  ChunkStitcher stitcher;
  std::vector<uint8_t> pending;  // stitched output not yet handed out (at most one slice + the junction bytes)
  size_t pending_pos = 0;
  int error = 0;                 // sticky BroccoliResult (>= 124)
  bool finished = false;
  bool file_open = false;        // BroccoliNewBrotliFile has been called at least once
};

Broccoli* Get(BroccoliState* s) {
  Broccoli* b;
  memcpy(&b, s->data, sizeof(b));
  return b;
}
BroccoliState Wrap(Broccoli* b) {
  BroccoliState s;
  memset(&s, 0, sizeof(s));
  memcpy(s.data, &b, sizeof(b));
  return s;
}
BroccoliResult Drain(Broccoli* b, size_t* available_out, uint8_t** out, BroccoliResult when_empty) {
  const size_t have = b->pending.size() - b->pending_pos;
  const size_t n = have < *available_out ? have : *available_out;
  if (n) {
    memcpy(*out, b->pending.data() + b->pending_pos, n);
    *out += n;
    *available_out -= n;
    b->pending_pos += n;
  }
  if (b->pending_pos == b->pending.size()) {
    b->pending.clear();
    b->pending_pos = 0;
    return when_empty;
  }
  return BroccoliNeedsMoreOutput;
}
}  // namespace

extern "C" {

BroccoliState BroccoliCreateInstance(void) { return Wrap(new (std::nothrow) Broccoli()); }

BroccoliState BroccoliCreateInstanceWithWindowSize(uint8_t window_size) {
  Broccoli* b = new (std::nothrow) Broccoli();
  // (an invalid size falls back to the plain constructor, like src/ffi/broccoli.rs:60-65)
  if (b && !b->stitcher.InitWithWindowSize(window_size)) b->stitcher = ChunkStitcher();
  return Wrap(b);
}

void BroccoliDestroyInstance(BroccoliState state) { delete Get(&state); }

void BroccoliNewBrotliFile(BroccoliState* state) {
  Broccoli* b = Get(state);
  if (!b) return;
  b->file_open = true;
  b->stitcher.BeginFile();
}

BroccoliResult BroccoliConcatStream(BroccoliState* state, size_t* available_in, const uint8_t** input_buf_ptr, size_t* available_out,
                                    uint8_t** output_buf_ptr) {
  Broccoli* b = Get(state);
  if (!b) return BroccoliBrotliFileNotCraftedForConcatenation;
  for (;;) {
    if (b->error) return (BroccoliResult)b->error;
    // what the slice before produced goes out first; no new input while the caller has not taken it
    const BroccoliResult r = Drain(b, available_out, output_buf_ptr, BroccoliNeedsMoreInput);
    if (r != BroccoliNeedsMoreInput) return r;
    if (*available_in == 0) return BroccoliNeedsMoreInput;
    if (!b->file_open) {
      // Bytes without a BroccoliNewBrotliFile in front of them: the reference's state machine has no file to put them in
      // (concat/mod.rs:450-468 runs into its unwrap).  They are taken for the start of a first file -- header, window and
      // catable checks included -- never forwarded unjudged.
      b->file_open = true;
      b->stitcher.BeginFile();
    }
    const size_t n = *available_in < kSlice ? *available_in : kSlice;
    ByteSink sink(&b->pending);
    if (!b->stitcher.Feed(*input_buf_ptr, n, &sink)) b->error = BroccoliBrotliFileNotCraftedForConcatenation;
    *input_buf_ptr += n;
    *available_in -= n;
  }
}

BroccoliResult BroccoliConcatStreaming(BroccoliState* state, size_t* available_in, const uint8_t* input_buf, size_t* available_out,
                                       uint8_t* output_buf) {
  return BroccoliConcatStream(state, available_in, &input_buf, available_out, &output_buf);
}

BroccoliResult BroccoliConcatFinish(BroccoliState* state, size_t* available_out, uint8_t** output_buf) {
  Broccoli* b = Get(state);
  if (!b) return BroccoliBrotliFileNotCraftedForConcatenation;
  if (!b->finished) {
    if (b->error == 0 && !b->stitcher.Finish(&b->pending)) b->error = BroccoliBrotliFileNotCraftedForAppend;
    b->finished = true;
  }
  if (b->error) return (BroccoliResult)b->error;
  return Drain(b, available_out, output_buf, BroccoliSuccess);
}

BroccoliResult BroccoliConcatFinished(BroccoliState* state, size_t* available_out, uint8_t* output_buf) {
  return BroccoliConcatFinish(state, available_out, &output_buf);
}
}
