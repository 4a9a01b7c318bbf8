// concat.h -- stitching of independently compressed chunks into one brotli stream.
// Restates BroCatli (reference src/concat/mod.rs:39-123, 274-608) for whole chunks and an unbounded output,
// the way CompressMulti drives it (src/enc/threading/mod.rs:565-660): new_brotli_file(); stream(chunk); ...; finish().
#ifndef BROTLI_MI355X_CONCAT_H_
#define BROTLI_MI355X_CONCAT_H_
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace brotli_mi355x {

class ChunkStitcher {
 public:
  // returns false when the chunk cannot be concatenated (not appendable / not catable / window too large)
  bool Append(const uint8_t* chunk, size_t size, std::vector<uint8_t>* out);
  bool Finish(std::vector<uint8_t>* out);

 private:
  bool FlushPreviousStream(std::vector<uint8_t>* out);
  uint8_t last_bytes_[2] = {0, 0};
  uint8_t last_bytes_len_ = 0;
  bool last_byte_sanitized_ = false;
  bool any_bytes_emitted_ = false;
  uint8_t last_byte_bit_offset_ = 0;
  uint8_t window_size_ = 0;
};

}  // namespace brotli_mi355x
#endif
