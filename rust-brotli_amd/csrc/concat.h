// concat.h -- stitching of independently compressed chunks into one brotli stream.
// Restates BroCatli (reference src/concat/mod.rs:39-123, 274-608): for whole chunks and an unbounded output, the way
// CompressMulti drives it (src/enc/threading/mod.rs:565-660): new_brotli_file(); stream(chunk); ...; finish() -- and, for
// the BroCatli C API, fed in pieces of any size with two bytes of state per junction (BeginFile / Feed).
#ifndef BROTLI_MI355X_CONCAT_H_
#define BROTLI_MI355X_CONCAT_H_
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace brotli_mi355x {

// Where the stitched stream goes: a growing vector or a caller-owned buffer (the bytes of a chunk are then copied
// exactly once; `overflow` is set and further bytes are dropped when the buffer is too small).
class ByteSink {
 public:
  explicit ByteSink(std::vector<uint8_t>* v) : vec_(v) {}
  ByteSink(uint8_t* buffer, size_t capacity) : buf_(buffer), cap_(capacity) {}
  void push_back(uint8_t b) {
    if (vec_) vec_->push_back(b);
    else if (size_ < cap_) buf_[size_++] = b;
    else overflow_ = true;
  }
  void append(const uint8_t* first, const uint8_t* last);
  // leaves n bytes of a caller-owned buffer unwritten (the caller fills them); returns their offset
  size_t skip(size_t n) {
    const size_t at = size_;
    if (n > cap_ - size_) overflow_ = true;
    else size_ += n;
    return at;
  }
  // (a caller-owned buffer that has overflowed, or is still empty, has no last byte to look at or take back)
  uint8_t back() const { return vec_ ? vec_->back() : ((overflow_ || size_ == 0) ? (uint8_t)0 : buf_[size_ - 1]); }
  void pop_back() {
    if (vec_) vec_->pop_back();
    else if (!overflow_ && size_ != 0) --size_;
  }
  size_t size() const { return vec_ ? vec_->size() : size_; }
  bool overflow() const { return overflow_; }

 private:
  std::vector<uint8_t>* vec_ = nullptr;
  uint8_t* buf_ = nullptr;
  size_t cap_ = 0, size_ = 0;
  bool overflow_ = false;
};

// A compressed chunk as the stitcher sees it: either all of its bytes, or only its first and last few (the body then
// stays where it is -- e.g. in device memory -- and the stitcher reports where it belongs in the output).
struct ChunkView {
  const uint8_t* full = nullptr;
  const uint8_t* head = nullptr;  // first min(size, head_len) bytes
  const uint8_t* tail = nullptr;  // last min(size, tail_len) bytes
  size_t head_len = 0, tail_len = 0;
  size_t size = 0;
  uint8_t at(size_t i) const { return full ? full[i] : (i < head_len ? head[i] : tail[i - (size - tail_len)]); }
};
struct BodyCopy {
  size_t dst_offset, src_offset, size;
};

class ChunkStitcher {
 public:
  // body != nullptr: the bytes of the chunk body are not copied; *body says which range of the chunk belongs where
  bool Append(const ChunkView& chunk, ByteSink* out, BodyCopy* body);
  // returns false when the chunk cannot be concatenated (not appendable / not catable / window too large)
  bool Append(const uint8_t* chunk, size_t size, ByteSink* out);
  bool Finish(ByteSink* out);
  // The same file by file, piece by piece (BroCatli::new_brotli_file / stream, concat/mod.rs:274-276, 450-566): nothing of
  // a file is kept but the (at most five) bytes of its header until they can be judged and the last two bytes handed over.
  void BeginFile();
  bool Feed(const uint8_t* piece, size_t size, ByteSink* out);
  bool Append(const uint8_t* chunk, size_t size, std::vector<uint8_t>* out) {
    ByteSink sink(out);
    return Append(chunk, size, &sink);
  }
  bool Finish(std::vector<uint8_t>* out) {
    ByteSink sink(out);
    return Finish(&sink);
  }

 private:
  bool FlushPreviousStream(ByteSink* out);
  // shift_and_check_new_stream_header (concat/mod.rs:332-449) + the take-back of the last byte written
  bool EmitNewStreamHeader(const uint8_t* header, size_t num_read, ByteSink* out);
  bool new_file_pending_ = false;  // BeginFile seen, nothing of the file yet
  bool in_header_ = false;
  uint8_t header_[5] = {0, 0, 0, 0, 0};
  uint8_t header_len_ = 0;
  uint8_t last_bytes_[2] = {0, 0};
  uint8_t last_bytes_len_ = 0;
  bool last_byte_sanitized_ = false;
  bool any_bytes_emitted_ = false;
  uint8_t last_byte_bit_offset_ = 0;
  uint8_t window_size_ = 0;

 public:
  // BroCatli::try_new_with_window_size (concat/mod.rs:232-272): the output stream starts with a header that announces
  // this window and an empty last meta-block, which the first real file then replaces; false = invalid window size
  bool InitWithWindowSize(uint8_t log_window_size);
};

}  // namespace brotli_mi355x
#endif
