// broccoli_api.cpp -- the BroCatli streaming C API of the reference (c/brotli/broccoli.h, src/ffi/broccoli.rs:55-175):
// concatenation of brotli streams that were encoded `catable` / `appendable`, fed and drained piecewise.
//
// The stitching rules are those of csrc/concat.cpp (ChunkStitcher = src/concat/mod.rs:274-608 restated for whole
// files).  This front end collects the bytes of the file that is being fed and runs the stitcher whenever a file is
// complete (BroccoliNewBrotliFile / BroccoliConcatFinish), so its memory grows with the largest input file instead of
// staying constant like the reference's byte-at-a-time state machine; the bytes that come out are the same.
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
struct Broccoli {
  ChunkStitcher stitcher;
  std::vector<uint8_t> file;     // bytes of the file being fed
  bool file_open = false;
  std::vector<uint8_t> pending;  // stitched output not yet handed out
  size_t pending_pos = 0;
  int error = 0;                 // sticky BroccoliResult (>= 124)
  bool finished = false;
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
void CloseFile(Broccoli* b) {
  if (!b->file_open) return;
  b->file_open = false;
  if (!b->stitcher.Append(b->file.data(), b->file.size(), &b->pending) && b->error == 0)
    b->error = BroccoliBrotliFileNotCraftedForConcatenation;
  b->file.clear();
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
  CloseFile(b);
  b->file_open = true;
}

BroccoliResult BroccoliConcatStream(BroccoliState* state, size_t* available_in, const uint8_t** input_buf_ptr, size_t* available_out,
                                    uint8_t** output_buf_ptr) {
  Broccoli* b = Get(state);
  if (!b) return BroccoliBrotliFileNotCraftedForConcatenation;
  if (b->error) return (BroccoliResult)b->error;
  if (*available_in) {
    if (!b->file_open) b->file_open = true;  // (the reference panics without new_brotli_file; be lenient)
    b->file.insert(b->file.end(), *input_buf_ptr, *input_buf_ptr + *available_in);
    *input_buf_ptr += *available_in;
    *available_in = 0;
  }
  return Drain(b, available_out, output_buf_ptr, BroccoliNeedsMoreInput);
}

BroccoliResult BroccoliConcatStreaming(BroccoliState* state, size_t* available_in, const uint8_t* input_buf, size_t* available_out,
                                       uint8_t* output_buf) {
  return BroccoliConcatStream(state, available_in, &input_buf, available_out, &output_buf);
}

BroccoliResult BroccoliConcatFinish(BroccoliState* state, size_t* available_out, uint8_t** output_buf) {
  Broccoli* b = Get(state);
  if (!b) return BroccoliBrotliFileNotCraftedForConcatenation;
  if (!b->finished) {
    CloseFile(b);
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
