// fragment_stream.cpp -- see fragment_stream.h
#include "fragment_stream.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>

#include "device_api.h"

namespace brotli_mi355x {

namespace {

// InitCommandPrefixCodes, encode.rs:627-659
void InitCommandPrefixCodes(FragmentState* st) {
  static const uint8_t kDefaultCommandDepths[128] = {
      0,  4,  4,  5,  6,  6,  7,  7,  7,  7,  7,  8,  8,  8,  8,  8,  0,  0,  0,  4,  4,  4,  4,  4,  5,  5,
      6,  6,  6,  6,  7,  7,  7,  7,  10, 10, 10, 10, 10, 10, 0,  4,  4,  5,  5,  5,  6,  6,  7,  8,  8,  9,
      10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 5,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
      0,  0,  6,  6,  6,  6,  6,  6,  5,  5,  5,  5,  5,  5,  4,  4,  4,  4,  4,  4,  4,  5,  5,  5,  5,  5,
      5,  6,  6,  7,  7,  7,  8,  10, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 0,  0,  0,  0};
  static const uint16_t kDefaultCommandBits[128] = {
      0,   0,   8,   9,   3,    35,   7,    71,   39,   103,  23,   47,   175,  111,  239,  31,   0,  0,  0,  4,
      12,  2,   10,  6,   13,   29,   11,   43,   27,   59,   87,   55,   15,   79,   319,  831,  191, 703, 447, 959,
      0,   14,  1,   25,  5,    21,   19,   51,   119,  159,  95,   223,  479,  991,  63,   575,  127, 639, 383, 895,
      255, 767, 511, 1023, 14,  0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,   0,   0,   0,
      27,  59,  7,   39,  23,   55,   30,   1,    17,   9,    25,   5,    0,    8,    4,    12,   2,   10,  6,   21,
      13,  29,  3,   19,  11,   15,   47,   31,   95,   63,   127,  255,  767,  2815, 1791, 3839, 511, 2559, 1535, 3583,
      1023, 3071, 2047, 4095, 0, 0,   0,    0};
  static const uint8_t kDefaultCommandCode[57] = {
      0xff, 0x77, 0xd5, 0xbf, 0xe7, 0xde, 0xea, 0x9e, 0x51, 0x5d, 0xde, 0xc6, 0x70, 0x57, 0xbc, 0x58, 0x58, 0x58, 0xd8,
      0xd8, 0x58, 0xd5, 0xcb, 0x8c, 0xea, 0xe0, 0xc3, 0x87, 0x1f, 0x83, 0xc1, 0x60, 0x1c, 0x67, 0xb2, 0xaa, 0x06, 0x83,
      0xc1, 0x60, 0x30, 0x18, 0xcc, 0xa1, 0xce, 0x88, 0x54, 0x94, 0x46, 0xe1, 0xb0, 0xd0, 0x4e, 0xb2, 0xf7, 0x04, 0x00};
  memset(st, 0, sizeof(*st));
  memcpy(st->cmd_depths, kDefaultCommandDepths, sizeof(kDefaultCommandDepths));
  memcpy(st->cmd_bits, kDefaultCommandBits, sizeof(kDefaultCommandBits));
  memcpy(st->cmd_code, kDefaultCommandCode, sizeof(kDefaultCommandCode));
  st->cmd_code_numbits = 448;
}

// ensure_initialized as far as this path needs it (encode.rs:657-707): the window bits become the open byte(s)
void Start(const EncoderParams& p, FragmentStream* fs) {
  if (fs->started) return;
  fs->started = true;
  const int lgwin = std::max(p.lgwin, 18);  // (quality 0 / 1, encode.rs:683-685)
  uint32_t bits = 0, n = 0;
  if (p.catable && p.bare_stream) {  // (no stream header, encode.rs:686-688)
  } else if (p.large_window) {
    bits = (uint32_t)(((lgwin & 0x3F) << 8) | 0x11);
    n = 14;
  } else if (lgwin == 16) {
    bits = 0;
    n = 1;
  } else if (lgwin == 17) {
    bits = 1;
    n = 7;
  } else if (lgwin > 17) {
    bits = (uint32_t)(((lgwin - 17) << 1) | 1);
    n = 4;
  } else {
    bits = (uint32_t)(((lgwin - 8) << 4) | 1);
    n = 7;
  }
  fs->last_bytes = (uint16_t)bits;
  fs->last_bytes_bits = (uint8_t)n;
  InitCommandPrefixCodes(&fs->state);
}

struct DevBuf {
  void* p = nullptr;
  explicit DevBuf(size_t bytes, bool zero = false) : p(zero ? dev_alloc(bytes) : dev_alloc_uninit(bytes)) {}
  ~DevBuf() { dev_free(p); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

// HashTableSize / GetHashTable, encode.rs:1643-1700
uint32_t TableBits(int quality, size_t input_size) {
  const size_t max_table_size = quality == 0 ? ((size_t)1 << 15) : ((size_t)1 << 17);
  size_t htsize = 256;
  while (htsize < max_table_size && htsize < input_size) htsize <<= 1;
  if (quality == 0 && (htsize & 0xaaaaa) == 0) htsize <<= 1;
  uint32_t bits = 0;
  while (((size_t)1 << bits) < htsize) ++bits;
  return bits;
}

}  // namespace

bool IsFragmentStream(const EncoderParams& user_params) {
  EncoderParams p = user_params;
  FinalizeParams(&p);
  return (p.quality == 0 || p.quality == 1) && !p.catable;
}

namespace {

// The fragments of `size` bytes (cut at 1 << lgwin, as compress_stream_fast cuts one call's input; one block of the ring-buffer path
// is a single fragment), one after the other on the device; `finish`: the last one carries is_last.  They go in batches of whole
// fragments (about 64 MiB of input, at least one fragment) whose bits land in one buffer each: device memory stays bounded however
// much one call hands over.  Whole bytes are appended to *out, the open byte stays in *fs.
void RunFragments(const EncoderParams& p, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, std::vector<uint8_t>* out) {
  const size_t block_size_limit = (size_t)1 << p.lgwin;
  static const size_t batch_target = getenv("BROTLI_MI355X_FRAGMENT_BATCH") ? (size_t)strtoull(getenv("BROTLI_MI355X_FRAGMENT_BATCH"), nullptr, 10) : ((size_t)64 << 20);
  const size_t per_batch = std::max<size_t>(1, batch_target / block_size_limit);
  const size_t batch_bytes = per_batch * block_size_limit;
  const size_t in_cap = std::min(size, batch_bytes), cap = 2 * in_cap + 503 * per_batch + 64;
  DevBuf in(in_cap + 64, true), outb(cap + 64, true), table(((size_t)1 << 17) * 4 + 64), commands(((size_t)1 << 17) * 4 + 64), literals(((size_t)1 << 17) + 64),
      state(sizeof(FragmentState) + 64);
  FragmentBuffers B;
  B.table = (uint32_t*)table.p;
  B.commands = (uint32_t*)commands.p;
  B.literals = (uint8_t*)literals.p;
  B.state = (FragmentState*)state.p;
  size_t done = 0;
  bool more = true;
  std::vector<uint8_t> bytes;
  while (more) {
    const size_t here = std::min(size - done, batch_bytes);
    if (here) dev_h2d_bulk(in.p, input + done, here);
    uint8_t head[2] = {(uint8_t)fs->last_bytes, (uint8_t)(fs->last_bytes >> 8)};
    dev_h2d(outb.p, head, 2);
    fs->state.storage_ix = fs->last_bytes_bits;
    fs->state.bad = 0;
    dev_h2d(state.p, &fs->state, sizeof(FragmentState));
    size_t at = 0;
    for (;;) {
      const size_t block_size = std::min(block_size_limit, here - at);
      const bool is_last = (size - done - at == block_size) && finish;
      if (block_size == 0 && !is_last) break;
      frag_compress(p.quality, (const uint8_t*)in.p + at, (uint32_t)block_size, is_last, TableBits(p.quality, block_size), B, (uint8_t*)outb.p);
      at += block_size;
      if (is_last || at == here) break;
    }
    done += here;
    more = done < size;
    dev_d2h(&fs->state, state.p, sizeof(FragmentState));
    if (fs->state.bad) throw std::runtime_error("brotli_mi355x: fragment compressor failed");
    const uint64_t ix = fs->state.storage_ix;
    if ((ix >> 3) + 2 > cap) throw std::runtime_error("brotli_mi355x: fragment output ran over its bound");
    bytes.resize((size_t)(ix >> 3) + 2);
    dev_d2h_bulk(bytes.data(), outb.p, bytes.size());
    out->insert(out->end(), bytes.begin(), bytes.begin() + (ptrdiff_t)(ix >> 3));
    fs->last_bytes = (uint16_t)(bytes[(size_t)(ix >> 3)] | (bytes[(size_t)(ix >> 3) + 1] << 8));
    fs->last_bytes_bits = (uint8_t)(ix & 7);
    if (fs->last_bytes_bits != 0) fs->last_bytes &= (uint16_t)((1u << fs->last_bytes_bits) - 1u); else fs->last_bytes = 0;
    if (more) dev_memset(outb.p, 0, (size_t)(ix >> 3) + 16);  // (the bit writer ORs nothing, but the open byte is read back: a clean start)
  }
}

// inject_byte_padding_block, encode.rs:1541-1566: an empty metadata block seals the open byte
void InjectPadding(FragmentStream* fs, std::vector<uint8_t>* out) {
  if (fs->last_bytes_bits == 0) return;
  uint32_t seal = fs->last_bytes;
  uint32_t seal_bits = fs->last_bytes_bits;
  seal |= 0x6u << seal_bits;
  seal_bits += 6;
  out->push_back((uint8_t)seal);
  if (seal_bits > 8) out->push_back((uint8_t)(seal >> 8));
  if (seal_bits > 16) out->push_back((uint8_t)(seal >> 16));
  fs->last_bytes = 0;
  fs->last_bytes_bits = 0;
}

// bits composed on the host behind the open byte; whole bytes go to *out, the rest becomes the open byte again
struct HostBits {
  FragmentStream* fs;
  std::vector<uint8_t>* out;
  uint64_t acc;
  uint32_t n;
  HostBits(FragmentStream* f, std::vector<uint8_t>* o) : fs(f), out(o), acc(f->last_bytes), n(f->last_bytes_bits) {
    while (n >= 8) {  // (the window bits of a large-window stream are 14)
      out->push_back((uint8_t)acc);
      acc >>= 8;
      n -= 8;
    }
  }
  void put(uint32_t nbits, uint64_t bits) {
    for (uint32_t b = 0; b < nbits; ++b) {
      acc |= ((bits >> b) & 1ull) << n;
      if (++n == 8) flush_byte();
    }
  }
  void flush_byte() {
    out->push_back((uint8_t)acc);
    acc = 0;
    n = 0;
  }
  void align() {
    if (n != 0) flush_byte();
  }
  void bytes(const uint8_t* p, size_t count) { out->insert(out->end(), p, p + count); }  // (byte aligned)
  ~HostBits() {
    fs->last_bytes = (uint16_t)acc;
    fs->last_bytes_bits = (uint8_t)n;
  }
};

}  // namespace

void FragmentStreamCompress(const EncoderParams& user_params, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, bool flush,
                            std::vector<uint8_t>* out) {
  EncoderParams p = user_params;
  FinalizeParams(&p);
  Start(p, fs);
  if (size != 0 || finish) RunFragments(p, fs, input, size, finish, out);
  if (flush) InjectPadding(fs, out);
}

bool IsFragmentRing(const EncoderParams& user_params) {
  EncoderParams p = user_params;
  FinalizeParams(&p);
  return (p.quality == 0 || p.quality == 1) && p.catable;
}

void FragmentRingCompress(const EncoderParams& user_params, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, bool flush,
                          std::vector<uint8_t>* out) {
  EncoderParams p = user_params;
  FinalizeParams(&p);
  Start(p, fs);
  const size_t block = (size_t)1 << p.lgblock;  // (ComputeLgBlock: lgwin at these qualities)
  size_t avail = size;
  bool processing = true;
  while (processing) {
    // copy_input_to_ring_buffer: up to the end of the input block
    const size_t room = fs->pending.size() >= block ? 0 : block - fs->pending.size();
    if (room != 0 && avail != 0) {
      const size_t n = std::min(room, avail);
      fs->pending.insert(fs->pending.end(), input + (size - avail), input + (size - avail) + n);
      fs->saw_input = true;
      avail -= n;
      continue;
    }
    if (!(room == 0 || finish || flush)) break;  // PROCESS with a block that is not full yet
    const bool is_last = avail == 0 && finish;
    const bool force_flush = avail == 0 && flush;
    // ---- encode_data, encode.rs:2214-2389
    if (fs->size_hint == 0) {  // update_size_hint, encode.rs:1604-1620
      const uint64_t total = (uint64_t)fs->pending.size() + avail;
      fs->size_hint = p.size_hint != 0 ? p.size_hint : (size_t)std::min<uint64_t>(total, (uint64_t)1 << 30);
    }
    size_t bytes = fs->pending.size(), skip = 0;
    {
      HostBits hb(fs, out);
      if (fs->first_mb == 0 && p.magic_number) {
        // BrotliWriteMetadataMetaBlock, brotli_bit_stream.rs:2853-2896
        uint8_t b128[10];
        size_t count = 0;
        uint64_t value = fs->size_hint;
        for (size_t index = 0; index < 10; ++index) {
          b128[index] = (uint8_t)(value & 0x7f);
          value >>= 7;
          count = index + 1;
          if (value != 0) b128[index] |= 0x80; else break;
        }
        hb.put(1, 0);
        hb.put(2, 3);
        hb.put(1, 0);
        hb.put(2, 1);
        hb.put(8, 3 + count);
        hb.align();
        const uint8_t magic[4] = {0xe1, 0x97, (uint8_t)((p.catable && !p.use_dictionary) ? 0x81 : (p.appendable ? 0x82 : 0x80)), 1};
        hb.bytes(magic, 4);
        hb.bytes(b128, count);
        fs->first_mb = 1;
      }
      if (fs->first_mb != 3 && bytes != 0) {
        // the first two bytes of a catable stream go out raw, in a meta-block of their own (encode.rs:2283-2333)
        const uint32_t n = (uint32_t)std::min<size_t>(2, bytes);
        hb.put(1, 0);
        hb.put(2, 0);
        hb.put(16, n - 1);
        hb.put(1, 1);
        hb.align();
        hb.bytes(fs->pending.data(), n);
        skip = n;
        bytes -= n;
        fs->first_mb = n >= 2 ? 3 : (fs->first_mb == 2 ? 3 : 2);
      }
    }
    if (!(bytes == 0 && !is_last)) RunFragments(p, fs, fs->pending.data() + skip, bytes, is_last, out);
    fs->pending.clear();
    if (force_flush) {
      InjectPadding(fs, out);
      processing = false;
    }
    if (is_last) processing = false;
  }
}

void FragmentStreamMetadataHeader(const EncoderParams& user_params, FragmentStream* fs, size_t size, std::vector<uint8_t>* out) {
  EncoderParams p = user_params;
  FinalizeParams(&p);
  Start(p, fs);
  uint8_t header[16];
  memset(header, 0, sizeof(header));
  size_t ix = fs->last_bytes_bits;
  header[0] = (uint8_t)fs->last_bytes;
  header[1] = (uint8_t)(fs->last_bytes >> 8);
  fs->last_bytes = 0;
  fs->last_bytes_bits = 0;
  auto put = [&](uint32_t n, uint64_t bits) {
    for (uint32_t b = 0; b < n; ++b, ++ix)
      if ((bits >> b) & 1) header[ix >> 3] |= (uint8_t)(1u << (ix & 7));
  };
  put(1, 0);
  put(2, 3);
  put(1, 0);
  if (size == 0) {
    put(2, 0);
  } else {
    uint32_t nbits = 0;
    if (size > 1) {
      uint32_t v = (uint32_t)size - 1;
      while (v) {
        nbits++;
        v >>= 1;
      }
    }
    const uint32_t nbytes = (nbits + 7) / 8;
    put(2, nbytes);
    put(8 * nbytes, (uint64_t)size - 1);
  }
  out->insert(out->end(), header, header + ((ix + 7) >> 3));
}

}  // namespace brotli_mi355x
