// concat.cpp -- see concat.h
#include "concat.h"

#include <string.h>

namespace brotli_mi355x {

namespace {

// parse_window_size, concat/mod.rs:39-74
bool ParseWindowSize(const uint8_t* b, uint8_t* window_size, size_t* offset) {
  if ((b[0] & 1) == 0) {
    *window_size = 16;
    *offset = 1;
    return true;
  }
  const uint8_t low = b[0] & 15;
  if (low == 0x3 || low == 0x5 || low == 0x7 || low == 0x9 || low == 0xb || low == 0xd || low == 0xf) {
    *window_size = (uint8_t)(18 + (low - 3) / 2);
    *offset = 4;
    return true;
  }
  switch (b[0] & 127) {
    case 0x71: *window_size = 15; *offset = 7; return true;
    case 0x61: *window_size = 14; *offset = 7; return true;
    case 0x51: *window_size = 13; *offset = 7; return true;
    case 0x41: *window_size = 12; *offset = 7; return true;
    case 0x31: *window_size = 11; *offset = 7; return true;
    case 0x21: *window_size = 10; *offset = 7; return true;
    case 0x01: *window_size = 17; *offset = 7; return true;
    default: break;
  }
  if (b[0] & 0x80) return false;
  const uint8_t ret = b[1] & 0x3f;
  if (ret < 10 || ret > 30) return false;
  *window_size = ret;
  *offset = 14;
  return true;
}

// detect_varlen_offset, concat/mod.rs:76-123: bit offset just behind the first meta-block header, which
// must be either an empty last block, a metadata block or an uncompressed block (byte aligned payload)
bool DetectVarlenOffset(const uint8_t* b, size_t n, size_t* out) {
  uint8_t ws;
  size_t offset;
  if (!ParseWindowSize(b, &ws, &offset)) return false;
  uint64_t bytes = 0;
  for (size_t i = 0; i < n; ++i) bytes |= (uint64_t)b[i] << (i * 8);
  bytes >>= offset;
  offset += 1;
  if (bytes & 1) {  // ISLAST
    bytes >>= 1;
    offset += 1;
    if (bytes & 1) {  // ISLASTEMPTY
      *out = offset;
      return true;
    }
  }
  bytes >>= 1;
  uint64_t mnibbles = bytes & 3;
  bytes >>= 2;
  offset += 2;
  if (mnibbles == 3) {  // metadata block
    if (bytes & 1) return false;
    bytes >>= 1;
    offset += 1;
    const uint64_t mskipbytes = bytes & 3;
    offset += 2;
    offset += (size_t)mskipbytes * 8;
    *out = offset;
    return true;
  }
  mnibbles += 4;
  offset += (size_t)mnibbles * 4;
  bytes >>= mnibbles * 4;
  offset += 1;
  if ((bytes & 1) == 0) return false;  // must be UNCOMPRESSED
  *out = offset;
  return true;
}

}  // namespace

// flush_previous_stream, concat/mod.rs:277-330: drops the ISLAST/ISLASTEMPTY bits that end the previous chunk
void ByteSink::append(const uint8_t* first, const uint8_t* last) {
  if (vec_) {
    vec_->insert(vec_->end(), first, last);
    return;
  }
  const size_t n = (size_t)(last - first);
  if (n > cap_ - size_) {
    overflow_ = true;
    return;
  }
  memcpy(buf_ + size_, first, n);
  size_ += n;
}

bool ChunkStitcher::FlushPreviousStream(ByteSink* out) {
  if (last_byte_sanitized_) return true;
  if (last_bytes_len_ == 0) {
    last_byte_sanitized_ = true;
    return true;
  }
  uint16_t last_bytes = (uint16_t)(last_bytes_[0] + (last_bytes_[1] << 8));
  const uint8_t max = (uint8_t)(last_bytes_len_ * 8);
  uint8_t index = (uint8_t)(max - 1);
  for (uint8_t i = 0; i < max; ++i) {
    index = (uint8_t)(max - 1 - i);
    if ((1u << index) & last_bytes) break;
  }
  if (index == 0) return false;
  if ((last_bytes >> (index - 1)) != 3) return false;
  index -= 1;
  last_bytes &= (uint16_t)((1u << index) - 1);
  last_bytes_[0] = (uint8_t)last_bytes;
  last_bytes_[1] = (uint8_t)(last_bytes >> 8);
  if (index >= 8) {
    out->push_back(last_bytes_[0]);
    last_bytes_[0] = last_bytes_[1];
    any_bytes_emitted_ = true;
    index -= 8;
    last_bytes_len_ -= 1;
  }
  last_byte_bit_offset_ = index;
  last_byte_sanitized_ = true;
  return true;
}

bool ChunkStitcher::EmitNewStreamHeader(const uint8_t* header, size_t num_read, ByteSink* out) {
  // shift_and_check_new_stream_header, concat/mod.rs:332-449
  uint8_t pending[6];
  size_t pending_len;
  uint8_t window_size;
  size_t window_offset;
  if (!ParseWindowSize(header, &window_size, &window_offset)) return false;
  if (window_size_ == 0) {
    window_size_ = window_size;
    memcpy(pending, header, num_read);
    pending_len = num_read;
  } else {
    if (window_size > window_size_) return false;
    uint8_t realigned[6] = {last_bytes_[0], 0, 0, 0, 0, 0};
    size_t varlen_offset;
    if (!DetectVarlenOffset(header, num_read, &varlen_offset)) return false;
    uint64_t bytes_so_far = 0;
    for (size_t i = 0; i < num_read; ++i) bytes_so_far |= (uint64_t)header[i] << (i * 8);
    bytes_so_far >>= window_offset;
    bytes_so_far &= (1ull << (varlen_offset - window_offset)) - 1;
    const size_t var_len_bytes = ((varlen_offset - window_offset) + 7) / 8;
    for (size_t byte_index = 0; byte_index < var_len_bytes; ++byte_index) {
      const uint64_t cur_byte = bytes_so_far >> (byte_index * 8);
      realigned[byte_index] |= (uint8_t)((cur_byte & ((1u << (8 - last_byte_bit_offset_)) - 1)) << last_byte_bit_offset_);
      realigned[byte_index + 1] = (uint8_t)(cur_byte >> (8 - last_byte_bit_offset_));
    }
    const size_t whole_byte_destination = ((size_t)last_byte_bit_offset_ + varlen_offset - window_offset + 7) / 8;
    const size_t whole_byte_source = (varlen_offset + 7) / 8;
    if (whole_byte_source > num_read) return false;
    const size_t num_whole_bytes_to_copy = num_read - whole_byte_source;
    for (size_t i = 0; i < num_whole_bytes_to_copy; ++i) realigned[whole_byte_destination + i] = header[whole_byte_source + i];
    pending_len = whole_byte_destination + num_whole_bytes_to_copy;
    memcpy(pending, realigned, pending_len);
  }
  out->append(pending, pending + pending_len);
  if (out->overflow()) return false;  // caller-owned buffer too small: nothing sensible can follow
  any_bytes_emitted_ = true;
  // the last byte may still change (next chunk / end of stream): take it back
  last_byte_sanitized_ = false;
  last_byte_bit_offset_ = 0;
  last_bytes_[0] = out->back();
  last_bytes_[1] = 0;
  last_bytes_len_ = 1;
  out->pop_back();
  return true;
}

void ChunkStitcher::BeginFile() {
  new_file_pending_ = true;  // (a header that never became long enough to judge is dropped with its file)
  in_header_ = false;
  header_len_ = 0;
}

bool ChunkStitcher::Feed(const uint8_t* p, size_t n, ByteSink* out) {
  size_t i = 0;
  if (new_file_pending_) {
    if (n == 0) return true;
    if (!FlushPreviousStream(out)) return false;
    new_file_pending_ = false;
    in_header_ = true;
    header_len_ = 0;
  }
  if (in_header_) {
    while (i < n && header_len_ < 5) header_[header_len_++] = p[i++];  // (as much of the five as this piece has)
    const bool sufficient = (header_len_ == 4 && (127 & header_[0]) != 17) || header_len_ == 5;
    if (!sufficient) return true;
    if (!EmitNewStreamHeader(header_, header_len_, out)) return false;
    in_header_ = false;
  }
  for (; i < n; ++i) {  // the body, two bytes behind
    if (last_bytes_len_ < 2) {
      last_bytes_[last_bytes_len_++] = p[i];
    } else {
      out->push_back(last_bytes_[0]);
      last_bytes_[0] = last_bytes_[1];
      last_bytes_[1] = p[i];
    }
  }
  return !out->overflow();
}

bool ChunkStitcher::Append(const uint8_t* in, size_t in_len, ByteSink* out) {
  ChunkView view;
  view.full = in;
  view.size = in_len;
  return Append(view, out, nullptr);
}

bool ChunkStitcher::Append(const ChunkView& in, ByteSink* out, BodyCopy* body) {
  const size_t in_len = in.size;
  if (body) *body = BodyCopy{0, 0, 0};
  // new_brotli_file() + stream(), concat/mod.rs:274-276, 450-566
  if (!FlushPreviousStream(out)) return false;
  uint8_t header[5] = {0, 0, 0, 0, 0};
  size_t num_read = in_len < 5 ? in_len : 5;
  for (size_t i = 0; i < num_read; ++i) header[i] = in.at(i);
  size_t in_offset = num_read;
  const bool sufficient = (num_read == 4 && (127 & header[0]) != 17) || num_read == 5;
  if (!sufficient) return true;  // the reference waits for more input that never comes: the chunk is dropped
  if (!EmitNewStreamHeader(header, num_read, out)) return false;
  // body: keep the last two bytes back
  while (last_bytes_len_ != 2) {
    if (in_offset == in_len) return true;
    last_bytes_[last_bytes_len_++] = in.at(in_offset++);
  }
  const size_t to_copy = in_len - in_offset;
  if (to_copy == 0) return true;
  if (to_copy == 1) {
    out->push_back(last_bytes_[0]);
    last_bytes_[0] = last_bytes_[1];
    last_bytes_[1] = in.at(in_offset);
    return true;
  }
  out->push_back(last_bytes_[0]);
  out->push_back(last_bytes_[1]);
  if (body) {
    *body = BodyCopy{out->skip(to_copy - 2), in_offset, to_copy - 2};
  } else {
    out->append(in.full + in_offset, in.full + in_offset + to_copy - 2);
  }
  last_bytes_[0] = in.at(in_offset + to_copy - 2);
  last_bytes_[1] = in.at(in_offset + to_copy - 1);
  return true;
}

bool ChunkStitcher::InitWithWindowSize(uint8_t ws) {
  uint8_t b0 = 0, b1 = 0, len = 0;
  if (ws > 24) {
    b0 = 17;
    b1 = (uint8_t)(ws | 64 | 128);
    len = 2;
  } else if (ws == 16) {
    b0 = 1 | 2 | 4;
    len = 1;
  } else if (ws > 17) {
    b0 = (uint8_t)((3 + (ws - 18) * 2) | (16 | 32));
    len = 1;
  } else {
    switch (ws) {
      case 15: b0 = 0x71 | 0x80; break;
      case 14: b0 = 0x61 | 0x80; break;
      case 13: b0 = 0x51 | 0x80; break;
      case 12: b0 = 0x41 | 0x80; break;
      case 11: b0 = 0x31 | 0x80; break;
      case 10: b0 = 0x21 | 0x80; break;
      case 17: b0 = 0x1 | 0x80; break;
      default: return false;
    }
    b1 = 1;
    len = 2;
  }
  last_bytes_[0] = b0;
  last_bytes_[1] = b1;
  last_bytes_len_ = len;
  last_byte_bit_offset_ = 0;
  last_byte_sanitized_ = false;
  any_bytes_emitted_ = false;
  window_size_ = ws;
  return true;
}

// finish, concat/mod.rs:567-608
bool ChunkStitcher::Finish(ByteSink* out) {
  if (last_byte_sanitized_ && last_bytes_len_ != 0) {
    uint16_t last_bytes = (uint16_t)(last_bytes_[0] | (last_bytes_[1] << 8));
    const uint8_t bit_end = (uint8_t)((last_bytes_len_ - 1) * 8 + last_byte_bit_offset_);
    last_bytes |= (uint16_t)(3u << bit_end);
    last_bytes_[0] = (uint8_t)last_bytes;
    last_bytes_[1] = (uint8_t)(last_bytes >> 8);
    last_byte_sanitized_ = false;
    last_byte_bit_offset_ += 2;
    if (last_byte_bit_offset_ >= 8) {
      last_byte_bit_offset_ -= 8;
      last_bytes_len_ += 1;
    }
  }
  while (last_bytes_len_ != 0) {
    out->push_back(last_bytes_[0]);
    last_bytes_len_ -= 1;
    last_bytes_[0] = last_bytes_[1];
    any_bytes_emitted_ = true;
  }
  if (!any_bytes_emitted_) {
    any_bytes_emitted_ = true;
    out->push_back(';');
  }
  return true;
}

}  // namespace brotli_mi355x
