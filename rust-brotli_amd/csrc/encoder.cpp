// encoder.cpp -- see encoder.h
#include "encoder.h"
#include "timeline.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <stdexcept>

#include "device_api.h"
#include "host_entropy.h"
#include "lz77_stage.h"
#include "metablock_api.h"
#include "metablock_device.h"
#include "metablock_hq.h"
#include "metablock_items.h"

namespace brotli_mi355x {

namespace {

struct DevMem {
  std::vector<void*> ptrs;
  // small buffers are carved out of zero-filled arenas (one fill per arena instead of one tiny fill kernel each)
  static constexpr size_t kArenaBytes = (size_t)4 << 20, kSmall = (size_t)256 << 10;
  uint8_t* arena = nullptr;
  size_t arena_used = kArenaBytes;
  ~DevMem() {
    for (void* p : ptrs) dev_free(p);
  }
  template <typename T>
  T* alloc(size_t count) {
    const size_t bytes = (count * sizeof(T) + 64 + 255) & ~(size_t)255;
    if (bytes <= kSmall) {
      if (arena_used + bytes > kArenaBytes) {
        arena = (uint8_t*)dev_alloc(kArenaBytes);
        ptrs.push_back(arena);
        arena_used = 0;
      }
      T* p = (T*)(arena + arena_used);
      arena_used += bytes;
      return p;
    }
    void* p = dev_alloc(bytes);
    ptrs.push_back(p);
    return (T*)p;
  }
};

// host-composed pieces of the stream (window bits, metadata block, uncompressed headers, tail blocks)
struct BitPiece {
  uint64_t pos;
  uint32_t nbits;
  uint32_t pad;
  uint64_t bits;
};

struct HostBits {
  uint64_t pos = 0;
  std::vector<BitPiece> pieces;
  void put(uint32_t nbits, uint64_t bits) {
    if (nbits == 0) return;
    pieces.push_back({pos, nbits, 0, bits});
    pos += nbits;
  }
  void jump_to_byte_boundary() { pos = (pos + 7) & ~(uint64_t)7; }
};

// BrotliWriteMetadataMetaBlock, brotli_bit_stream.rs:2853-2896
void WriteMetadataMetaBlock(const EncoderParams& p, HostBits* hb) {
  uint8_t b128[10];
  size_t count = 0;
  uint64_t value = p.size_hint;
  for (size_t index = 0; index < 10; ++index) {
    b128[index] = (uint8_t)(value & 0x7f);
    value >>= 7;
    count = index + 1;
    if (value != 0) {
      b128[index] |= 0x80;
    } else {
      break;
    }
  }
  hb->put(1, 0);
  hb->put(2, 3);
  hb->put(1, 0);
  hb->put(2, 1);
  hb->put(8, 3 + count);
  hb->jump_to_byte_boundary();
  uint8_t magic[3] = {0xe1, 0x97, 0x80};
  if (p.catable && !p.use_dictionary) {
    magic[2] = 0x81;
  } else if (p.appendable) {
    magic[2] = 0x82;
  }
  for (int i = 0; i < 3; ++i) hb->put(8, magic[i]);
  hb->put(8, 1);  // VERSION, src/lib.rs:67
  for (size_t i = 0; i < count; ++i) hb->put(8, b128[i]);
}

// BrotliStoreUncompressedMetaBlockHeader, brotli_bit_stream.rs:2743-2756
void WriteUncompressedHeader(uint32_t length, HostBits* hb) {
  const uint32_t lg = length == 1 ? 1 : (31u ^ (uint32_t)__builtin_clz(length - 1)) + 1;
  const uint32_t mnibbles = (lg < 16 ? 16 : lg + 3) / 4;
  hb->put(1, 0);
  hb->put(2, mnibbles - 4);
  hb->put(mnibbles * 4, length - 1);
  hb->put(1, 1);
  hb->jump_to_byte_boundary();
}

// WriteEmptyLastBlocksInternal, encode.rs:1928-1940
void WriteEmptyLastBlocks(const EncoderParams& p, HostBits* hb) {
  if (p.byte_align && (hb->pos & 7) != 0) {  // BrotliWritePaddingMetaBlock
    hb->put(6, 6);
    hb->jump_to_byte_boundary();
  }
  if (!p.bare_stream) {  // BrotliWriteEmptyLastMetaBlock
    hb->put(1, 1);
    hb->put(1, 1);
    hb->jump_to_byte_boundary();
  }
}

// ChooseContextMap (encode.rs:1717-1780), ShouldUseComplexStaticContextMap (:1802-1871),
// DecideOverLiteralContextModeling (:1873-1927) evaluated on the device-collected sample histograms.
void DecideContexts(const uint32_t* s, int quality, size_t size_hint, size_t length, uint32_t* num_contexts,
                    uint32_t* map_id) {
  *num_contexts = 1;
  *map_id = 0;
  if (quality < 5 || length < 64) return;
  if (size_hint >= (1u << 20)) {
    const uint32_t* combined = s + 16;
    const uint32_t* context = s + 48;
    const uint32_t total = s[480];
    float entropy[3];
    entropy[1] = HostShannonEntropy(combined, 32, nullptr);
    entropy[2] = 0.0f;
    for (size_t i = 0; i < 13; ++i) entropy[2] += HostShannonEntropy(context + 32 * i, 32, nullptr);
    entropy[0] = 1.0f / (float)total;
    entropy[1] *= entropy[0];
    entropy[2] *= entropy[0];
    if (!(entropy[2] > 3.0f || entropy[1] - entropy[2] < 0.2f)) {
      *num_contexts = 13;
      *map_id = 3;
      return;
    }
  }
  uint32_t bigram[9];
  memcpy(bigram, s, sizeof(bigram));
  uint32_t monogram[3] = {0, 0, 0};
  uint32_t two_prefix[6] = {0, 0, 0, 0, 0, 0};
  float entropy[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < 9; ++i) {
    monogram[i % 3] += bigram[i];
    two_prefix[i % 6] += bigram[i];
  }
  entropy[1] = HostShannonEntropy(monogram, 3, nullptr);
  entropy[2] = HostShannonEntropy(two_prefix, 3, nullptr) + HostShannonEntropy(two_prefix + 3, 3, nullptr);
  entropy[3] = 0.0f;
  for (size_t i = 0; i < 3; ++i) entropy[3] += HostShannonEntropy(bigram + 3 * i, 3, nullptr);
  const size_t total = (size_t)(monogram[0] + monogram[1] + monogram[2]);
  entropy[0] = 1.0f / (float)total;
  entropy[1] *= entropy[0];
  entropy[2] *= entropy[0];
  entropy[3] *= entropy[0];
  if (quality < 7) entropy[3] = entropy[1] * 10.0f;
  if (entropy[1] - entropy[2] < 0.2f && entropy[1] - entropy[3] < 0.2f) {
    *num_contexts = 1;
  } else if (entropy[2] - entropy[3] < 0.02f) {
    *num_contexts = 2;
    *map_id = 1;
  } else {
    *num_contexts = 3;
    *map_id = 2;
  }
}

struct Clock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double lap(bool sync, const char* what = nullptr) {
    if (what) timeline().stamp(what);
    if (sync) dev_sync();
    auto t1 = std::chrono::steady_clock::now();
    double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t0 = t1;
    return ms;
  }
};

}  // namespace

// tail of a piece that leaves the stream open: the injected flush (an empty metadata block, only when the stream is
// not byte aligned, encode.rs:1541-1566) or the header of a real metadata block (write_metadata_header, :2545-2575)
template <typename Bits>
void WriteOpenTail(const EncodeRequest& req, Bits* bits) {
  if (req.emit_metadata) {
    bits->put(1, 0);
    bits->put(2, 3);
    bits->put(1, 0);
    if (req.metadata_size == 0) {
      bits->put(2, 0);
    } else {
      uint32_t nbits = 0;
      if (req.metadata_size > 1) {
        uint32_t v = (uint32_t)req.metadata_size - 1;
        while (v) {
          nbits++;
          v >>= 1;
        }
      }
      const uint32_t nbytes = (nbits + 7) / 8;
      bits->put(2, nbytes);
      if (nbytes) bits->put(8 * nbytes, (uint64_t)req.metadata_size - 1);
    }
    bits->jump_to_byte_boundary();
  } else if ((bits->pos & 7) != 0) {
    bits->put(6, 6);
    bits->jump_to_byte_boundary();
  }
}

// The reference maps stream positions from 3 GiB on back into [1 GiB, 3 GiB) and empties its hash table whenever that
// mapping jumps backwards, i.e. when the input position passes 3, 5, 7 ... GiB (WrapPosition / update_last_processed_pos
// / HasherReset, encode.rs:1623-1631, 1705-1710, 2472-2474): the block that starts there searches an empty table (into
// which StitchToPreviousBlock has just put the three positions in front of it).
// BROTLI_MI355X_TEST_WRAP_SHIFT: log2 of the unit (30 = GiB); the tests scale it down together with the oracle's.
namespace {
int WrapShift() {
  static const int v = getenv("BROTLI_MI355X_TEST_WRAP_SHIFT") ? atoi(getenv("BROTLI_MI355X_TEST_WRAP_SHIFT")) : 30;
  return v;
}
// largest reset position <= pos (0 if there is none)
uint64_t LastHasherResetAtOrBelow(uint64_t pos) {
  const uint64_t unit = 1ull << WrapShift();
  if (pos < 3 * unit) return 0;
  return 3 * unit + (pos - 3 * unit) / (2 * unit) * (2 * unit);
}
}  // namespace

// stream position of the reference's first hasher reset (3 GiB; the tests scale it down, WrapShift)
uint64_t FirstPositionWrap() { return 3ull << WrapShift(); }

uint32_t ChooseSegmentBytes(size_t input_bytes) {
  const uint32_t forced = getenv("BROTLI_MI355X_SEGMENT_BYTES") ? (uint32_t)atoi(getenv("BROTLI_MI355X_SEGMENT_BYTES")) : 0u;  // (experiments, tests)
  if (forced) return forced;
  if (input_bytes <= ((size_t)512 << 10)) return 256;  // (a small call lasts as long as its chains: 152 KB at quality 5 2.51 -> 2.36 ms)
  if (input_bytes <= ((size_t)4 << 20)) return 512;
  if (input_bytes <= ((size_t)16 << 20)) return 1024;
  if (input_bytes <= ((size_t)128 << 20)) return 2048;
  return 4096;
}

void EncodeStream(const EncodeRequest& req, std::vector<uint8_t>* out, EncodeStats* stats_out) {
  EncodeStats stats;
  Clock total_clock;
  timeline().begin();
  struct TimelineEnd {
    ~TimelineEnd() { timeline().end(); }
  } timeline_end;
  const bool prof = getenv("BROTLI_MI355X_PROFILE") != nullptr;
  EncoderParams p = req.params;
  const size_t n = req.input_size;
  if (n >= (1ull << 31) || req.prefix_size >= (1ull << 30)) throw std::runtime_error("brotli_mi355x: streams of 2 GiB or more are not supported");
  // ---- parameters, in the order the reference fixes them (encode.rs:657-707, 1604-1620, 1125-1161)
  const bool continuing = req.carry_in != nullptr && req.carry_in->valid;  // a later piece of a flushed stream
  if (req.prefix_size && !continuing) p.use_dictionary = false;  // set_custom_dictionary, encode.rs:1213
  FinalizeParams(&p);
  if (continuing) {
    // both were fixed by the first encode_data of the stream (a size hint that came out 0 there -- a flush or a metadata block
    // before any input -- stays unset and is filled in by the next one, encode.rs:1604-1620)
    p.size_hint = req.carry_in->size_hint != 0 ? req.carry_in->size_hint : req.params.size_hint;
    p.hasher = req.carry_in->hasher;
  } else {
    if (req.hasher_chosen_before_size_hint) {
      // custom dictionary: hasher_setup ran inside set_custom_dictionary (encode.rs:1234, 1125-1161), with the size hint
      // the caller had set by then -- what update_size_hint fills in later does not reach the hasher parameters
      EncoderParams early = p;
      if (req.has_hasher_size_hint) early.size_hint = req.hasher_size_hint;
      ChooseHasher(&early);
      p.hasher = early.hasher;
    }
    if (p.size_hint == 0) p.size_hint = std::min<size_t>(n, (size_t)1 << 30);  // update_size_hint with everything offered at once
    if (!req.hasher_chosen_before_size_hint) ChooseHasher(&p);
  }
  if (continuing && !req.carry_in->use_dictionary) p.use_dictionary = false;  // (turned off by the stream's first piece)
  const char* why = nullptr;
  if (!IsAccelerated(p, &why)) throw std::runtime_error(std::string("brotli_mi355x: ") + why);
  bool catable = p.catable;
  bool appendable = p.appendable;
  if (!continuing && req.carry_in == nullptr && req.prefix != nullptr && (req.prefix_size <= 1)) {
    // set_custom_dictionary with a too-short dictionary: no priming, but catable + appendable (encode.rs:1237-1241)
    catable = true;
    appendable = true;
  }
  p.catable = catable;
  p.appendable = appendable;
  const uint32_t prefix_bytes = (continuing || req.prefix_size > 1) ? (uint32_t)req.prefix_size : 0;

  // context bytes in front of a meta-block read as 0 below this text position: a custom dictionary does not count as stream
  // until the first meta-block has been written (the reference sets prev_byte only behind a written meta-block)
  const uint32_t prev_floor = continuing ? req.carry_in->prev_floor : ((prefix_bytes && !req.prefix_is_file_continuation) ? prefix_bytes : 0u);
  HostBits hb;
  // stream header: window bits (EncodeWindowBits, encode.rs:603-625)
  if (!continuing && !(req.params.catable && p.bare_stream)) {
    const int lgwin = p.lgwin;
    if (p.large_window) {
      hb.put(14, (uint64_t)(((lgwin & 0x3F) << 8) | 0x11));
    } else if (lgwin == 16) {
      hb.put(1, 0);
    } else if (lgwin == 17) {
      hb.put(7, 1);
    } else if (lgwin > 17) {
      hb.put(4, (uint64_t)(((lgwin - 17) << 1) | 1));
    } else {
      hb.put(7, (uint64_t)(((lgwin - 8) << 4) | 1));
    }
  }
  if (continuing && req.carry_in->tail_nbits) hb.put(req.carry_in->tail_nbits, req.carry_in->tail_bits);  // the open last byte of the piece in front
  // The magic-number block is written by the first encode_data of the stream (encode.rs:2261-2282).  A metadata block asked for
  // before any input has arrived does not go through encode_data (process_metadata writes its header straight away,
  // encode.rs:2630-2640): the magic number then comes BEHIND it, with the first piece that does (StreamCarry::magic_owed).
  const bool metadata_before_anything = req.emit_metadata && !req.finish && n == 0;
  const bool magic_due = p.magic_number && (!continuing || req.carry_in->magic_owed);
  if (magic_due && !metadata_before_anything) WriteMetadataMetaBlock(p, &hb);
  if (!continuing && req.finish && n == 0 && p.byte_align && p.appendable && !p.catable && (hb.pos & 7) != 0) {
    hb.put(6, 6);
    hb.jump_to_byte_boundary();
  }

  // positions are 32-bit on the device (like the reference's u32 ring positions, which wrap instead); longer streams
  // go through BrotliEncoderCompressMulti / the chunk entry points
  if ((uint64_t)prefix_bytes + (uint64_t)n > 0xE0000000ull)
    throw std::runtime_error("brotli_mi355x: more than 3.5 GiB in one stream is not supported, split it into chunks");
  DevMem mem;
  const uint32_t M = prefix_bytes + (uint32_t)n;
  uint8_t* text = mem.alloc<uint8_t>((size_t)M + 64);
  if (prefix_bytes) dev_h2d_bulk(text, req.prefix + (req.prefix_size - prefix_bytes), prefix_bytes);
  if (n) {
    if (req.input_on_device) {
      dev_d2d(text + prefix_bytes, req.input, n);
    } else {
      dev_h2d_bulk(text + prefix_bytes, req.input, n);
    }
  }
  stats.ms_phase[0] = total_clock.lap(prof, "input-copied");

  struct RawCopy {
    uint64_t dst_byte;
    uint32_t src_pos, bytes;
  };
  std::vector<RawCopy> raw_copies;
  uint32_t raw_head = 0;
  // is_first_mb of the reference (encode.rs:2283-2333): nothing / one / both of the raw first bytes are out.  Every encode_data
  // before "both" stores min(2, bytes) raw -- a flush behind the very first byte makes it three raw bytes in all.
  const uint32_t raw_state = continuing ? req.carry_in->catable_raw_bytes : 0u;
  if (p.catable && n != 0 && raw_state < 2) {
    raw_head = (uint32_t)std::min<size_t>(2, n);
    WriteUncompressedHeader(raw_head, &hb);
    raw_copies.push_back({hb.pos >> 3, prefix_bytes, raw_head});
    hb.pos += (uint64_t)raw_head * 8;
  }
  const uint64_t head_bits = hb.pos;

  Lz77Stage lz;
  std::vector<uint8_t> result;
  bool wrote_direct = false;
  if (n - raw_head == 0) {
    // nothing left to search: only the trailing blocks (encode.rs:1979-1982)
    if (req.finish) {
      WriteEmptyLastBlocks(p, &hb);
    } else {
      WriteOpenTail(req, &hb);
    }
    if (req.carry_out) {
      StreamCarry& co = *req.carry_out;
      if (continuing) co = *req.carry_in;
      co.tail_bits = co.tail_nbits = 0;  // (this piece ends byte aligned: flush padding or the end of the stream)
      co.magic_owed = magic_due && metadata_before_anything;
      co.catable_raw_bytes = raw_head >= 2 ? 2u : (raw_head == 1 ? (raw_state == 1 ? 2u : 1u) : raw_state);
      co.use_dictionary = p.use_dictionary;
      co.prev_floor = prev_floor;  // (no meta-block written by this piece)
      if (!co.valid) {
        co.valid = true;
        co.hasher = p.hasher;
        co.size_hint = p.size_hint;
        const int32_t far_away = 0x7ffffff0;  // a catable stream starts without usable last distances (encode.rs:693-703)
        const int32_t d0[4] = {p.catable ? far_away : 4, p.catable ? far_away : 11, p.catable ? far_away : 15, p.catable ? far_away : 16};
        memcpy(co.dist_cache, d0, sizeof(d0));
        co.dict_break = prefix_bytes;
        if (prefix_bytes != 0) {
          // nothing searched yet behind a custom dictionary: the hash table holds what HasherPrependCustomDictionary put there,
          // every dictionary position but the last StoreLookahead - 1 (encode.rs:1196-1270, mod.rs:224-229)
          const uint32_t htl = IsH6Family(p.hasher.type) ? 8u : 4u;
          co.stored.assign(prefix_bytes, 0);
          if (prefix_bytes > htl - 1) std::fill(co.stored.begin(), co.stored.begin() + (prefix_bytes - (htl - 1)), (uint8_t)1);
        }
      }
    }
    const size_t total_bytes = (size_t)((hb.pos + 7) >> 3);
    result.assign(total_bytes + 8, 0);
    for (const BitPiece& bp : hb.pieces) {
      uint64_t v = bp.bits;
      for (uint32_t b = 0; b < bp.nbits; ++b, v >>= 1)
        if (v & 1) result[(bp.pos + b) >> 3] |= (uint8_t)(1u << ((bp.pos + b) & 7));
    }
    for (const RawCopy& rc : raw_copies) dev_d2h(result.data() + rc.dst_byte, text + rc.src_pos, rc.bytes);
    result.resize(total_bytes);
    if (req.direct_out) {
      if (result.size() > req.direct_capacity) throw std::runtime_error("brotli_mi355x: output buffer too small");
      if (req.direct_out_on_device) {
        dev_h2d(req.direct_out, result.data(), result.size());
        dev_sync();
      } else {
        memcpy(req.direct_out, result.data(), result.size());
      }
      *req.direct_size = result.size();
    } else if (out->empty()) {
      out->swap(result);
    } else {
      out->insert(out->end(), result.begin(), result.end());
    }
    if (stats_out) *stats_out = stats;
    return;
  }

  const uint32_t segment_bytes = req.segment_bytes ? req.segment_bytes : ChooseSegmentBytes(n);
  {
    Clock c;
    lz.SetStreamState(continuing ? req.carry_in : nullptr, req.finish, req.partial && !req.finish, req.last_block_processed_early);
    lz.Setup(p, text, prefix_bytes, (uint32_t)n, raw_head, segment_bytes);
    {
      // a hasher reset inside this input?  (One that coincides with its start has been applied to the carry already.)
      const uint64_t base = continuing ? req.carry_in->stream_base : 0;
      const uint64_t reset = LastHasherResetAtOrBelow(base + M - 1);
      if (reset > base + prefix_bytes) {
        if (LastHasherResetAtOrBelow(reset - 1) > base + prefix_bytes)
          throw std::runtime_error("brotli_mi355x: two hasher resets (position wraps) inside one piece of a stream");
        lz.SetHasherReset((uint32_t)(reset - base));
      }
    }
    stats.ms_phase[9] = c.lap(prof, "lz77-setup");
  }
  for (;;) {  // repeated only when a compressed meta-block turns out larger than its raw form
    HostBits bits = hb;  // stream position after the head pieces
    bits.pos = head_bits;
    Clock clk;
    lz.Run();
    stats.ms_lz77 += clk.lap(prof, "lz77-done");
    {
      double ms;
      uint32_t launches;
      uint64_t segs;
      uint64_t work[3] = {0, 0, 0};
      lz77_parse_timing(&ms, &launches, &segs, work);
      stats.parse_walked += work[0];
      stats.parse_searches += work[1];
      stats.parse_commands += work[2];
      stats.parse_kernel_ms += ms;
      stats.parse_launches += launches;
      stats.parse_segments += segs;
      stats.num_segments = lz.device_params().num_segments;
      stats.segment_bytes = segment_bytes;
    }
    stats.lz77_rounds += lz.stats().rounds;
    stats.searches = lz.stats().searches;
    const std::vector<MetaBlockPlan>& plans = lz.metablocks();
    const uint32_t n_mb = (uint32_t)plans.size();
    const bool partial = req.partial && !req.finish;
    if (partial && n_mb == 0) {
      // no meta-block closes within this input: nothing is emitted, nothing changes, the caller comes back with more
      if (req.consumed_out) *req.consumed_out = 0;
      if (req.keep_from_out) *req.keep_from_out = 0;
      if (req.direct_size) *req.direct_size = 0;
      if (stats_out) *stats_out = stats;
      return;
    }
    // (in a partial piece the commands of the meta-block that is still open are not used)
    const uint32_t K = partial ? plans.back().cmd_offset + plans.back().n_cmds : lz.num_commands();
    uint64_t L64 = 0;
    for (const MetaBlockPlan& mp : plans) L64 += mp.n_literals;
    const uint32_t L = (uint32_t)L64;
    // every meta-block stored uncompressed (incompressible input): nothing of the per-symbol machinery below is needed -- no
    // literal map of a gigabyte of literals, no scans, no symbol bits -- only the layout and the copies of the bytes
    bool all_raw = n_mb != 0;
    for (const MetaBlockPlan& mp : plans) all_raw = all_raw && mp.uncompressed;
    const size_t Ka = all_raw ? 0 : K, La = all_raw ? 0 : L;

    DevMem mm;
    MbBuffers B{};
    const DeviceTables& dt = dev_tables();
    B.text = text;
    B.cmds = lz.commands_dev();
    B.n_cmds = K;
    B.n_lits = L;
    B.n_mb = n_mb;
    B.text_base = plans.empty() ? 0 : plans[0].start;
    B.utf8_lut = dt.utf8_context_lookup;
    B.signed_lut = dt.signed_context_lookup;
    B.et.logs_16 = dt.logs_16;
    B.et.logs_8 = dt.logs_8;
    const bool hq = p.quality >= 10;  // the quality >= 10 meta-block builder behind the greedy chains ("9.5"), metablock_hq.h
    B.header_stride = hq ? kHqHeaderWords : kHeaderWords;
    B.descs = mm.alloc<MbDesc>(n_mb);
    B.results = mm.alloc<MbResult>(n_mb);
    B.cmd_lit_start = mm.alloc<uint32_t>(Ka + 1);
    B.cmd_pos = mm.alloc<uint32_t>(Ka + 1);
    B.cmd_dist_index = mm.alloc<uint32_t>(Ka + 1);
    B.lit_pos = mm.alloc<uint32_t>(La + 1);
    B.lit_cmd = mm.alloc<uint32_t>(La + 1);
    B.lit_nbits = mm.alloc<uint32_t>(La + 1);
    B.cmd_nbits = mm.alloc<uint32_t>(Ka + 1);
    B.cmd_own_bits = mm.alloc<uint32_t>(Ka + 1);
    void* scan_scratch = mm.alloc<uint8_t>(mb_scan_scratch_bytes(std::max<size_t>(Ka, La) + 2));

    std::vector<MbDesc> descs(n_mb);
    for (uint32_t m = 0; m < n_mb; ++m) {
      MbDesc& d = descs[m];
      memset(&d, 0, sizeof(d));
      const MetaBlockPlan& mp = plans[m];
      d.start = mp.start;
      d.end = mp.end;
      d.cmd_offset = mp.cmd_offset;
      d.n_cmds = mp.n_cmds;
      d.n_lits = mp.n_literals;
      d.context_mode = 2;  // ChooseContextMode: UTF8 below quality 10 unless forced (encode.rs:1357-1377)
      switch (p.mode) {
        case 3: d.context_mode = 0; break;
        case 4: d.context_mode = 1; break;
        case 6: d.context_mode = 3; break;
        default: break;
      }
      d.uncompressed = mp.uncompressed ? 1 : 0;
      const bool actual_last = mp.is_last;
      d.is_last = (actual_last && !(p.appendable || p.byte_align)) ? 1 : 0;
      d.num_distance_symbols = p.dist.alphabet_size;
      d.dist_postfix_bits = p.dist.distance_postfix_bits;
      d.num_direct_distance_codes = p.dist.num_direct_distance_codes;
      d.num_contexts = 1;
      // qualities 2 and 3 (store_meta_block_fast / _trivial, metablock_fast.h): one block type per kind, context mode bits 0
      d.simple = p.quality <= 2 ? kMbFast : (p.quality < 4 ? kMbTrivial : kMbGreedy);
      if (d.simple != kMbGreedy) d.context_mode = 0;
    }
    // prev bytes (encode.rs:2526-2534): bytes preceding the meta-block in the stream, 0 at the very start
    if (!all_raw) {
      std::vector<uint32_t> where((size_t)n_mb * 2, 0xffffffffu);
      for (uint32_t m = 0; m < n_mb; ++m) {
        const uint32_t s = descs[m].start;
        const uint32_t lo = prev_floor;
        if (s >= lo + 2) where[2 * m] = s - 2;
        if (s >= lo + 1) where[2 * m + 1] = s - 1;
      }
      uint32_t* where_dev = mm.alloc<uint32_t>((size_t)n_mb * 2);
      uint8_t* tails_dev = mm.alloc<uint8_t>((size_t)n_mb * 2);
      std::vector<uint8_t> tails((size_t)n_mb * 2, 0);
      dev_h2d(where_dev, where.data(), where.size() * 4);
      mb_gather_bytes(text, where_dev, n_mb * 2, tails_dev);
      dev_d2h(tails.data(), tails_dev, tails.size());
      for (uint32_t m = 0; m < n_mb; ++m) {
        descs[m].prev_byte = tails[2 * m + 1];
        descs[m].prev_byte2 = tails[2 * m];
      }
    }
    {
      uint32_t lit_base = 0;
      for (uint32_t m = 0; m < n_mb; ++m) {
        descs[m].lit_base = lit_base;
        lit_base += descs[m].n_lits;
      }
    }
    std::vector<MbResult> results(n_mb);
    std::vector<uint32_t> body_off(n_mb + 1);
    if (!all_raw) {
    dev_h2d(B.descs, descs.data(), n_mb * sizeof(MbDesc));
    uint32_t* boundary_words = mm.alloc<uint32_t>(n_mb + 1);
    mb_command_scans(B, scan_scratch);
    mb_literal_map(B);
    // distance symbol counts per meta-block
    {
      std::vector<uint32_t> di(n_mb + 1);
      mb_gather_at_metablock_starts(B, B.cmd_dist_index, boundary_words);
      dev_d2h(di.data(), boundary_words, (n_mb + 1) * 4);
      for (uint32_t m = 0; m < n_mb; ++m) {
        descs[m].dist_base = di[m];
        descs[m].n_dists = di[m + 1] - di[m];
      }
      B.n_dists = di[n_mb];
    }
    if (hq) {
      // ---- BrotliBuildMetaBlock (metablock.rs:133-307) on the device, metablock_hq.h
      for (uint32_t m = 0; m < n_mb; ++m) {
        descs[m].hq = 1u | (p.large_window ? 2u : 0u);
        descs[m].hq_no_context = p.disable_literal_context_modeling ? 1 : 0;
        descs[m].n_symbols[0] = descs[m].n_lits;
        descs[m].n_symbols[1] = descs[m].n_cmds;
        descs[m].n_symbols[2] = descs[m].n_dists;
      }
      {
        // the distance-parameter search re-codes the distance prefixes of a meta-block: on a copy, the LZ77 stage keeps its own
        Command* copy = mm.alloc<Command>((size_t)K + 1);
        dev_d2d(copy, B.cmds, (size_t)K * sizeof(Command));
        B.cmds = copy;
        B.cmds_rw = copy;
      }
      dev_h2d(B.descs, descs.data(), n_mb * sizeof(MbDesc));
      const bool census = !(p.mode == 3 || p.mode == 4 || p.mode == 5 || p.mode == 6);
      if (census) mb_hq_utf8_census(B);
      mb_hq_distance_params(B);
      dev_d2h(results.data(), B.results, n_mb * sizeof(MbResult));
      for (uint32_t m = 0; m < n_mb; ++m) {
        MbDesc& d = descs[m];
        if (census) d.context_mode = results[m].hq_mostly_utf8 ? 2 : 3;  // ChooseContextMode, encode.rs:1357-1377
        d.dist_postfix_bits = results[m].hq_postfix;
        d.num_direct_distance_codes = results[m].hq_ndirect;
        d.num_distance_symbols = 16 + d.num_direct_distance_codes + ((p.large_window ? 62u : 24u) << (d.dist_postfix_bits + 1));
      }
      B.hq_sym[0] = mm.alloc<uint16_t>((size_t)L + 8);
      B.hq_sym[1] = mm.alloc<uint16_t>((size_t)K + 8);
      B.hq_sym[2] = mm.alloc<uint16_t>((size_t)B.n_dists + 8);
      mb_hq_gather_symbols(B);
      // BrotliSplitBlock, block_splitter.rs:840-929: one job per (meta-block, kind)
      static const uint32_t kSymbolsPerHistogram[3] = {544, 530, 544}, kMaxHistograms[3] = {100, 50, 50}, kStride[3] = {70, 40, 40};
      static const float kSwitchCost[3] = {28.1f, 13.5f, 14.6f};
      uint8_t* block_ids[3] = {mm.alloc<uint8_t>((size_t)L + 8), mm.alloc<uint8_t>((size_t)K + 8), mm.alloc<uint8_t>((size_t)B.n_dists + 8)};
      std::vector<HqSplitJob> jobs;
      for (uint32_t m = 0; m < n_mb; ++m) {
        const MbDesc& d = descs[m];
        if (d.uncompressed) continue;
        const uint32_t base[3] = {d.lit_base, d.cmd_offset, d.dist_base};
        for (uint32_t k = 0; k < 3; ++k) {
          HqSplitJob j;
          memset(&j, 0, sizeof(j));
          j.m = m;
          j.kind = k;
          j.length = d.n_symbols[k];
          j.alphabet = kRowLen[k];
          j.num_histograms = std::min(j.length / kSymbolsPerHistogram[k] + 1, kMaxHistograms[k]);
          j.stride = kStride[k];
          j.iters = p.quality <= 11 ? 3 : 10;  // block_splitter.rs:794
          j.block_switch_cost = kSwitchCost[k];
          j.data = B.hq_sym[k] + base[k];
          j.block_ids = block_ids[k] + base[k];
          if (j.length >= 128) {
            const size_t nh = j.num_histograms, bitmaplen = (nh + 7) >> 3;
            j.histo_data = mm.alloc<uint32_t>((nh + 1) * j.alphabet);
            j.histo_total = mm.alloc<uint32_t>(nh + 1);
            j.insert_cost = mm.alloc<float>((size_t)j.alphabet * nh + nh + 8);
            j.cost = mm.alloc<float>(bitmaplen * 8 + 8);
            j.switch_signal = mm.alloc<uint8_t>((size_t)j.length * bitmaplen + 8);
            j.new_id = mm.alloc<uint16_t>(nh + 8);
          }
          jobs.push_back(j);
        }
      }
      HqSplitJob* jobs_dev = mm.alloc<HqSplitJob>(jobs.size() + 1);
      dev_h2d(jobs_dev, jobs.data(), jobs.size() * sizeof(HqSplitJob));
      mb_hq_find_blocks(B, jobs_dev, (uint32_t)jobs.size());
      dev_d2h(jobs.data(), jobs_dev, jobs.size() * sizeof(HqSplitJob));
      stats.ms_phase[1] += clk.lap(prof, "hq-find-blocks");
      // pools sized by what FindBlocks found; ClusterBlocks
      uint32_t block_total[3] = {0, 0, 0}, histo_total[3] = {0, 0, 0};
      for (uint32_t m = 0; m < n_mb; ++m)
        for (uint32_t k = 0; k < 3; ++k) {
          descs[m].histo_base[k] = histo_total[k];
          descs[m].max_histos[k] = 256;
          histo_total[k] += 256;
        }
      {
        // ClusterBlocks keeps one histogram per block FindBlocks found, twice (the batches' rows and the gathered clusters);
        // a pathological input (a switch every few symbols over a 16 MiB meta-block) would ask for more than the device
        // has.  Refuse with a message instead of dying in an allocation half-way through.
        uint64_t need = 0;
        for (const HqSplitJob& j : jobs)
          if (j.length >= 128) need += (uint64_t)(j.num_blocks + 2) * (2ull * j.alphabet * 4 + 64ull * sizeof(HqPair) + 64);
        if (need > (96ull << 30))
          throw std::runtime_error("brotli_mi355x: quality 9.5: clustering " + std::to_string(need >> 30) +
                                   " GiB of block histograms is more than this build sets aside; feed the stream in smaller pieces");
      }
      for (HqSplitJob& j : jobs) {
        MbDesc& d = descs[j.m];
        const uint32_t nb = j.num_blocks;
        d.block_base[j.kind] = block_total[j.kind];
        d.max_blocks[j.kind] = nb + 1;
        block_total[j.kind] += nb + 2;
        if (j.length >= 128) {
          const size_t n = nb;
          uint64_t max_pairs = std::min<uint64_t>(64ull * n, (uint64_t)(n / 2) * n);
          max_pairs = std::max<uint64_t>(max_pairs, kHqBatchPairs);
          j.histogram_symbols = mm.alloc<uint32_t>(n + 8);
          j.block_lengths = mm.alloc<uint32_t>(n + 8);
          j.block_pos = mm.alloc<uint32_t>(n + 8);
          j.batch_data = mm.alloc<uint32_t>((n + 2) * j.alphabet);
          j.batch_total = mm.alloc<uint32_t>(n + 8);
          j.batch_cost = mm.alloc<float>(n + 8);
          j.batch_sizes = mm.alloc<uint32_t>(n + 8);
          j.batch_clusters = mm.alloc<uint32_t>(n + 8);
          j.batch_symbols = mm.alloc<uint32_t>(n + 8);
          j.batch_count = mm.alloc<uint32_t>(n / kHqBatch + 8);
          j.all_data = mm.alloc<uint32_t>((n + 2) * j.alphabet);
          j.all_total = mm.alloc<uint32_t>(n + 2);
          j.all_cost = mm.alloc<float>(n + 2);
          j.cluster_size = mm.alloc<uint32_t>(n + 8);
          j.clusters = mm.alloc<uint32_t>(n + 2 * kHqBatch + 8);  // (also the per-batch sizes / cluster lists)
          j.new_index = mm.alloc<uint32_t>(n + 2 * kHqBatch + 8);
          j.pairs = mm.alloc<HqPair>(max_pairs + 2);
        }
      }
      for (uint32_t k = 0; k < 3; ++k) {
        B.n_granules[k] = 0;
        B.histo[k] = mm.alloc<uint32_t>((size_t)histo_total[k] * kRowLen[k] + 8);
        B.depth[k] = mm.alloc<uint8_t>((size_t)histo_total[k] * kRowLen[k] + 8);
        B.bits[k] = mm.alloc<uint16_t>((size_t)histo_total[k] * kRowLen[k] + 8);
        B.tree_bits[k] = mm.alloc<uint64_t>((size_t)histo_total[k] * kTreeBitsWords + 8);
        B.tree_nbits[k] = mm.alloc<uint32_t>(histo_total[k] + 8);
        B.block_types[k] = mm.alloc<uint8_t>(block_total[k] + 8);
        B.block_lengths[k] = mm.alloc<uint32_t>(block_total[k] + 8);
        B.block_start[k] = mm.alloc<uint32_t>(block_total[k] + 8);
        B.switch_bits[k] = mm.alloc<uint64_t>(block_total[k] + 8);
        B.switch_nbits[k] = mm.alloc<uint8_t>(block_total[k] + 8);
      }
      B.header_words = mm.alloc<uint64_t>((size_t)n_mb * kHqHeaderWords);
      B.ctxmap_scratch = mm.alloc<uint32_t>((size_t)n_mb * 2 * 256 * 64);
      B.mb_out_bit = mm.alloc<uint64_t>(n_mb + 1);
      dev_h2d(B.descs, descs.data(), n_mb * sizeof(MbDesc));
      dev_h2d(jobs_dev, jobs.data(), jobs.size() * sizeof(HqSplitJob));
      std::vector<HqBatchRef> batch_refs;
      for (uint32_t ji = 0; ji < jobs.size(); ++ji)
        if (jobs[ji].length >= 128)
          for (uint32_t b = 0; b * kHqBatch < jobs[ji].num_blocks; ++b) batch_refs.push_back({ji, b});
      HqBatchRef* batch_refs_dev = mm.alloc<HqBatchRef>(batch_refs.size() + 1);
      dev_h2d(batch_refs_dev, batch_refs.data(), batch_refs.size() * sizeof(HqBatchRef));
      mb_hq_cluster_blocks(B, jobs_dev, (uint32_t)jobs.size(), batch_refs_dev, (uint32_t)batch_refs.size());
      dev_d2h(results.data(), B.results, n_mb * sizeof(MbResult));
      stats.ms_phase[2] += clk.lap(prof, "hq-cluster-blocks");
      // context histograms and the clustered context maps (histogram.rs:465-534, cluster.rs:353-465)
      uint32_t rows_total[2] = {0, 0}, map_total[2] = {0, 0};
      std::vector<HqClusterJob> cjobs;
      for (uint32_t m = 0; m < n_mb; ++m) {
        MbDesc& d = descs[m];
        if (d.uncompressed) continue;
        const uint32_t nt[2] = {results[m].num_types[kSplitLiteral], results[m].num_types[kSplitDistance]};
        const uint32_t rows[2] = {d.hq_no_context ? nt[0] : nt[0] << 6, nt[1] << 2};
        const uint32_t maps[2] = {nt[0] << 6, nt[1] << 2};
        for (uint32_t w = 0; w < 2; ++w) {
          d.hq_ctx_row_base[w] = rows_total[w];
          d.hq_ctx_map_base[w] = map_total[w];
          rows_total[w] += rows[w];
          map_total[w] += maps[w];
        }
      }
      B.hq_ctx_histo[0] = mm.alloc<uint32_t>((size_t)rows_total[0] * 256 + 8);
      B.hq_ctx_histo[1] = mm.alloc<uint32_t>((size_t)rows_total[1] * kNumDistanceHistoSymbols + 8);
      B.hq_ctx_map[0] = mm.alloc<uint32_t>((size_t)map_total[0] + 8);
      B.hq_ctx_map[1] = mm.alloc<uint32_t>((size_t)map_total[1] + 8);
      for (uint32_t m = 0; m < n_mb; ++m) {
        const MbDesc& d = descs[m];
        if (d.uncompressed) continue;
        for (uint32_t w = 0; w < 2; ++w) {
          HqClusterJob j;
          memset(&j, 0, sizeof(j));
          j.m = m;
          j.kind = w == 0 ? kSplitLiteral : kSplitDistance;
          const uint32_t nt = results[m].num_types[j.kind];
          j.in_size = w == 0 ? (d.hq_no_context ? nt : nt << 6) : nt << 2;
          j.len = kRowLen[j.kind];
          j.expand64 = (w == 0 && d.hq_no_context) ? 1 : 0;
          j.in_data = B.hq_ctx_histo[w] + (size_t)d.hq_ctx_row_base[w] * j.len;
          const size_t n = j.in_size, re = std::min<size_t>(n, 256);
          j.in_total = mm.alloc<uint32_t>(n + 8);
          j.out_data = mm.alloc<uint32_t>((n + 2) * j.len);
          j.out_total = mm.alloc<uint32_t>(n + 2);
          j.out_cost = mm.alloc<float>(n + 2);
          j.cluster_size = mm.alloc<uint32_t>(n + 8);
          j.clusters = mm.alloc<uint32_t>(n + 8);
          j.symbols = mm.alloc<uint32_t>(n + 8);
          j.new_index = mm.alloc<uint32_t>(n + 8);
          j.batch_count = mm.alloc<uint32_t>(n / kHqBatch + 8);
          j.reindex_data = mm.alloc<uint32_t>(re * j.len + 8);
          j.reindex_total = mm.alloc<uint32_t>(re + 8);
          j.reindex_cost = mm.alloc<float>(re + 8);
          j.pairs = mm.alloc<HqPair>(std::max<size_t>(kHqBatchPairs, 64 * n) + 2);
          cjobs.push_back(j);
        }
      }
      dev_h2d(B.descs, descs.data(), n_mb * sizeof(MbDesc));
      mb_hq_context_histograms(B);
      HqClusterJob* cjobs_dev = mm.alloc<HqClusterJob>(cjobs.size() + 1);
      dev_h2d(cjobs_dev, cjobs.data(), cjobs.size() * sizeof(HqClusterJob));
      std::vector<HqBatchRef> cbatch_refs;
      for (uint32_t ji = 0; ji < cjobs.size(); ++ji)
        for (uint32_t b = 0; b * kHqBatch < cjobs[ji].in_size; ++b) cbatch_refs.push_back({ji, b});
      HqBatchRef* cbatch_refs_dev = mm.alloc<HqBatchRef>(cbatch_refs.size() + 1);
      dev_h2d(cbatch_refs_dev, cbatch_refs.data(), cbatch_refs.size() * sizeof(HqBatchRef));
      mb_hq_cluster_histograms(B, cjobs_dev, (uint32_t)cjobs.size(), cbatch_refs_dev, (uint32_t)cbatch_refs.size());
      dev_d2h(results.data(), B.results, n_mb * sizeof(MbResult));
      for (uint32_t m = 0; m < n_mb; ++m) results[m].num_histos[kSplitCommand] = results[m].num_types[kSplitCommand];
      dev_h2d(B.results, results.data(), n_mb * sizeof(MbResult));
      stats.ms_phase[3] += clk.lap(prof, "hq-context-maps");
    } else {
      // literal context modelling decision
      if (p.disable_literal_context_modeling == 0 && p.quality >= 5) {  // (DecideOverLiteralContextModeling leaves lower qualities alone)
        uint32_t* stats_dev = mm.alloc<uint32_t>((size_t)n_mb * kContextStatsWords);
        mb_context_stats(B, stats_dev);
        std::vector<uint32_t> cs((size_t)n_mb * kContextStatsWords);
        dev_d2h(cs.data(), stats_dev, cs.size() * 4);
        for (uint32_t m = 0; m < n_mb; ++m) {
          if (descs[m].uncompressed) continue;
          DecideContexts(cs.data() + (size_t)m * kContextStatsWords, p.quality, p.size_hint, descs[m].end - descs[m].start,
                         &descs[m].num_contexts, &descs[m].context_map_id);
        }
      }
      // pools
      uint32_t gran_total[3] = {0, 0, 0}, row_total[3] = {0, 0, 0}, block_total[3] = {0, 0, 0}, histo_total[3] = {0, 0, 0};
      for (uint32_t m = 0; m < n_mb; ++m) {
        MbDesc& d = descs[m];
        d.n_symbols[0] = d.n_lits;
        d.n_symbols[1] = d.n_cmds;
        d.n_symbols[2] = d.n_dists;
        for (uint32_t k = 0; k < 3; ++k) {
          const uint32_t gl = kGranuleLen[k];
          const uint32_t nc = k == 0 ? d.num_contexts : 1;
          d.granule_base[k] = gran_total[k];
          d.n_granules[k] = d.uncompressed ? 0 : (d.n_symbols[k] + gl - 1) / gl;
          d.gran_row_base[k] = row_total[k];
          d.block_base[k] = block_total[k];
          d.max_blocks[k] = d.n_symbols[k] / gl + 1;
          const uint32_t max_types = (k == 0 && nc > 1) ? 256 / nc : 256;
          d.histo_base[k] = histo_total[k];
          d.max_histos[k] = std::min(d.max_blocks[k], max_types + 1) * nc;
          gran_total[k] += d.n_granules[k];
          row_total[k] += d.n_granules[k] * nc;
          block_total[k] += d.max_blocks[k] + 1;
          histo_total[k] += d.max_histos[k];
        }
      }
      std::vector<uint32_t> gran_mb_host[3];
      for (uint32_t k = 0; k < 3; ++k) {
        gran_mb_host[k].resize(gran_total[k] + 1);
        for (uint32_t m = 0; m < n_mb; ++m)
          for (uint32_t g = 0; g < descs[m].n_granules[k]; ++g) gran_mb_host[k][descs[m].granule_base[k] + g] = m;
        B.n_granules[k] = gran_total[k];
        B.gran_mb[k] = mm.alloc<uint32_t>(gran_total[k] + 1);
        dev_h2d(B.gran_mb[k], gran_mb_host[k].data(), (size_t)gran_total[k] * 4);
        B.gran_hist[k] = mm.alloc<uint16_t>((size_t)row_total[k] * kRowLen[k] + 8);
        B.gran_block[k] = mm.alloc<uint16_t>(gran_total[k] + 8);
        B.histo[k] = mm.alloc<uint32_t>((size_t)histo_total[k] * kRowLen[k] + 8);
        B.depth[k] = mm.alloc<uint8_t>((size_t)histo_total[k] * kRowLen[k] + 8);
        B.bits[k] = mm.alloc<uint16_t>((size_t)histo_total[k] * kRowLen[k] + 8);
        B.tree_bits[k] = mm.alloc<uint64_t>((size_t)histo_total[k] * kTreeBitsWords + 8);
        B.tree_nbits[k] = mm.alloc<uint32_t>(histo_total[k] + 8);
        B.block_types[k] = mm.alloc<uint8_t>(block_total[k] + 8);
        B.block_lengths[k] = mm.alloc<uint32_t>(block_total[k] + 8);
        B.switch_bits[k] = mm.alloc<uint64_t>(block_total[k] + 8);
        B.switch_nbits[k] = mm.alloc<uint8_t>(block_total[k] + 8);
      }
      B.header_words = mm.alloc<uint64_t>((size_t)n_mb * kHeaderWords);
      B.ctxmap_scratch = mm.alloc<uint32_t>((size_t)n_mb * 2 * 256 * 64);
      B.mb_out_bit = mm.alloc<uint64_t>(n_mb + 1);
      dev_h2d(B.descs, descs.data(), n_mb * sizeof(MbDesc));
      stats.ms_phase[1] += clk.lap(prof, "mb1");
      mb_granule_histograms(B);
      stats.ms_phase[2] += clk.lap(prof, "mb2");
      bool wide = false;
      for (uint32_t m = 0; m < n_mb; ++m) wide = wide || (!descs[m].uncompressed && descs[m].num_contexts > 3);
      mb_split_chains(B, wide);
      dev_d2h(results.data(), B.results, n_mb * sizeof(MbResult));
      stats.ms_phase[3] += clk.lap(prof, "mb3");
    }
    // Huffman codes: one job per histogram
    std::vector<CodeJob> jobs;
    for (uint32_t m = 0; m < n_mb; ++m) {
      if (descs[m].uncompressed) continue;
      for (uint32_t k = 0; k < 3; ++k) {
        uint32_t mode = kCodeOptimized;
        if (descs[m].simple == kMbTrivial) mode = kCodePlain;
        // (quality 2: up to 128 commands are written with the static command and distance codes, brotli_bit_stream.rs:2619-2687)
        if (descs[m].simple == kMbFast) mode = (descs[m].n_cmds <= 128 && k != kSplitLiteral) ? kCodeStatic : kCodeFast;
        for (uint32_t i = 0; i < results[m].num_histos[k]; ++i) jobs.push_back({k, descs[m].histo_base[k] + i, descs[m].num_distance_symbols, mode});
      }
    }
    B.huff_scratch = mm.alloc<HuffmanScratch>(std::max<size_t>(jobs.size(), n_mb) + 1);
    CodeJob* jobs_dev = mm.alloc<CodeJob>(jobs.size() + 1);
    dev_h2d(jobs_dev, jobs.data(), jobs.size() * sizeof(CodeJob));
    mb_build_codes(B, jobs_dev, (uint32_t)jobs.size());
    stats.ms_phase[4] += clk.lap(prof, "mb4");
    mb_write_headers(B);
    stats.ms_phase[5] += clk.lap(prof, "mb5");
    mb_symbol_bits(B, scan_scratch);
    mb_gather_at_metablock_starts(B, B.cmd_nbits, boundary_words);
    dev_d2h_async(results.data(), B.results, n_mb * sizeof(MbResult));
    dev_d2h_async(body_off.data(), boundary_words, (n_mb + 1) * 4);
    dev_sync();
    if (getenv("BROTLI_MI355X_DEBUG_MB"))
      for (uint32_t m = 0; m < n_mb; ++m)
        fprintf(stderr, "  meta-block %u [%u,%u) cmds %u lits %u contexts %u | literal blocks %u types %u | command blocks %u types %u | distance blocks %u types %u | header %u bits\n",
                m, descs[m].start, descs[m].end, descs[m].n_symbols[1], descs[m].n_symbols[0], descs[m].num_contexts, results[m].num_blocks[0],
                results[m].num_types[0], results[m].num_blocks[1], results[m].num_types[1], results[m].num_blocks[2], results[m].num_types[2],
                results[m].header_bits);
    stats.ms_phase[6] += clk.lap(prof, "mb6");
    }  // !all_raw

    // ---- layout of the stream (WriteMetaBlockInternal, encode.rs:1941-2167)
    std::vector<uint64_t> mb_out_bit(n_mb + 1, 0);
    std::vector<RawCopy> copies = raw_copies;
    int fallback = -1;
    for (uint32_t m = 0; m < n_mb; ++m) {
      const MbDesc& d = descs[m];
      const bool actual_last = plans[m].is_last;
      const bool is_last = d.is_last != 0;
      const uint32_t bytes = d.end - d.start;
      if (d.uncompressed) {
        WriteUncompressedHeader(bytes, &bits);
        copies.push_back({bits.pos >> 3, d.start, bytes});
        bits.pos += (uint64_t)bytes * 8;
        if (is_last) {
          bits.put(1, 1);
          bits.put(1, 1);
          bits.jump_to_byte_boundary();
        }
      } else {
        const uint64_t start_bit = bits.pos;
        mb_out_bit[m] = start_bit;
        const uint64_t body_bits = body_off[m + 1] - body_off[m];
        bits.pos += results[m].header_bits + body_bits;
        if (is_last) bits.jump_to_byte_boundary();
        if ((uint64_t)bytes + 4 + (start_bit >> 3) < (bits.pos >> 3)) {  // encode.rs:2141-2163
          fallback = (int)m;
          break;
        }
      }
      if (actual_last != is_last) WriteEmptyLastBlocks(p, &bits);
    }
    if (fallback >= 0) {
      lz.ForceUncompressed((uint32_t)fallback);
      stats.fallback_retries++;
      continue;
    }
    if (req.finish && lz.needs_empty_last()) WriteEmptyLastBlocks(p, &bits);
    if (!req.finish && !partial) WriteOpenTail(req, &bits);
    // ---- emission
    const size_t all_bytes = (size_t)((bits.pos + 7) >> 3);
    // a partial piece hands out whole bytes only; its last, incomplete byte goes into the carry
    const uint32_t tail_nbits = partial ? (uint32_t)(bits.pos & 7) : 0u;
    const size_t total_bytes = tail_nbits ? all_bytes - 1 : all_bytes;
    const size_t out_words = all_bytes / 8 + 4;
    B.out_words = mm.alloc<uint64_t>(out_words);
    if (!all_raw) {
      dev_h2d(B.mb_out_bit, mb_out_bit.data(), (n_mb + 1) * 8);
      mb_emit(B);
    }
    {
      // the headers of the compressed meta-blocks, the bytes of the stored ones and what the host composed (stream header, the
      // headers of stored meta-blocks, tail blocks): one upload and three launches, however many meta-blocks there are -- an
      // incompressible gigabyte is 700 stored meta-blocks, and a launch + an upload + an allocation per piece was 100 ms of it
      std::vector<MbBitCopy> headers;
      for (uint32_t m = 0; m < n_mb; ++m)
        if (!descs[m].uncompressed) headers.push_back({mb_out_bit[m], (uint64_t)m * B.header_stride, results[m].header_bits});
      std::vector<MbRawCopy> raws;
      for (const RawCopy& rc : copies) raws.push_back({rc.dst_byte, rc.src_pos, rc.bytes});
      std::vector<MbBitPiece> pieces;
      for (const BitPiece& bp : bits.pieces) pieces.push_back({bp.pos, bp.nbits, 0, bp.bits});
      const size_t bytes = headers.size() * sizeof(MbBitCopy) + raws.size() * sizeof(MbRawCopy) + pieces.size() * sizeof(MbBitPiece);
      if (bytes != 0) {
        std::vector<uint8_t> blob(bytes);
        uint8_t* at = blob.data();
        memcpy(at, headers.data(), headers.size() * sizeof(MbBitCopy));
        at += headers.size() * sizeof(MbBitCopy);
        memcpy(at, raws.data(), raws.size() * sizeof(MbRawCopy));
        at += raws.size() * sizeof(MbRawCopy);
        memcpy(at, pieces.data(), pieces.size() * sizeof(MbBitPiece));
        uint8_t* blob_dev = mm.alloc<uint8_t>(bytes + 64);
        dev_h2d(blob_dev, blob.data(), bytes);
        const MbBitCopy* headers_dev = (const MbBitCopy*)blob_dev;
        const MbRawCopy* raws_dev = (const MbRawCopy*)(blob_dev + headers.size() * sizeof(MbBitCopy));
        const MbBitPiece* pieces_dev = (const MbBitPiece*)(blob_dev + headers.size() * sizeof(MbBitCopy) + raws.size() * sizeof(MbRawCopy));
        // (a grid dimension holds 65 535 items)
        for (size_t i = 0; i < headers.size(); i += 32768) mb_copy_bits_batch(B.out_words, B.header_words, headers_dev + i, (uint32_t)std::min<size_t>(32768, headers.size() - i));
        for (size_t i = 0; i < raws.size(); i += 32768) mb_raw_copies((uint8_t*)B.out_words, text, raws_dev + i, (uint32_t)std::min<size_t>(32768, raws.size() - i));
        mb_place_pieces(B.out_words, pieces_dev, (uint32_t)pieces.size());
        dev_sync();  // (blob is host memory that goes out of scope)
      }
    }
    stats.ms_phase[7] += clk.lap(prof, "mb7");
    if (req.direct_out) {
      if (total_bytes > req.direct_capacity) throw std::runtime_error("brotli_mi355x: output buffer too small");
      if (req.direct_out_on_device) {
        dev_d2d(req.direct_out, B.out_words, total_bytes);
        dev_sync();
      } else {
        dev_d2h_bulk(req.direct_out, B.out_words, total_bytes);
      }
      *req.direct_size = total_bytes;
      wrote_direct = true;
    } else {
      result.resize(total_bytes);
      dev_d2h_bulk(result.data(), B.out_words, total_bytes);
    }
    stats.ms_phase[8] += clk.lap(prof, "mb8");
    const uint32_t resume = partial ? lz.resume_pos() : M;  // text position where the next piece takes over
    if (req.consumed_out) *req.consumed_out = resume - prefix_bytes;
    if (req.carry_out) {
      // State for the next piece of the stream: distance cache, dictionary counters, what is in the hash table.  Only a
      // window of the stream so far is kept as the next prefix: at least one ring buffer of bytes (matches reach back
      // max_backward < ring size, and the byte one ring revolution back is what the reference's ring buffer holds behind
      // the end of a block), starting at a multiple of the ring size, so that ring-buffer indices stay what they were.
      StreamCarry co;
      co.valid = true;
      co.magic_owed = false;
      co.hasher = p.hasher;
      co.size_hint = p.size_hint;
      co.dict_break = continuing ? req.carry_in->dict_break : prefix_bytes;
      co.prev_floor = 0;
      co.use_dictionary = p.use_dictionary;
      co.catable_raw_bytes = raw_head >= 2 ? 2u : (raw_head == 1 ? (raw_state == 1 ? 2u : 1u) : raw_state);
      memcpy(co.dist_cache, plans.back().dist_cache_after, sizeof(co.dist_cache));
      if (partial) {
        co.dict_lookups = plans.back().dict_lookups_after;
        co.dict_matches = plans.back().dict_matches_after;
        co.dict_dead = plans.back().dict_dead_after;
      } else {
        lz.FinalDictState(&co.dict_lookups, &co.dict_matches, &co.dict_dead);
      }
      const uint64_t ring = (uint64_t)lz.device_params().ring_mask + 1;
      const uint64_t old_base = continuing ? req.carry_in->stream_base : 0;
      const uint64_t abs_resume = old_base + resume;
      uint64_t new_base = old_base;
      if (abs_resume >= 2 * ring && !getenv("BROTLI_MI355X_KEEP_WHOLE_STREAM")) new_base = (abs_resume - ring) / ring * ring;
      const uint32_t keep_from = (uint32_t)(new_base - old_base);
      co.stream_base = new_base;
      if (lz.is_zopfli()) {
        // qualities 10 / 11: the hasher travels as it is -- the H10 trees at the resume point (zopfli_device.h) -- instead of
        // as the set of stored positions
        lz.ExportZopfli(&co, partial);
      } else if (lz.is_quick()) {
        lz.ExportQuick(&co, partial);  // qualities 2 .. 4: likewise the BasicHasher table (quick_device.h)
      } else {
      std::vector<uint8_t> all(M);
      lz.DumpFlags(all.data(), M);
      // a hasher reset at or in front of the resume point wipes what lies before it (minus the three stitched positions)
      const uint64_t reset = LastHasherResetAtOrBelow(abs_resume);
      if (reset > old_base) {
        const uint32_t vis = (uint32_t)(reset - old_base) - 3;
        for (uint32_t q = 0; q < vis && q < resume; ++q) all[q] = 0;
        if (keep_from > vis) {
          lz.KeyCountsBetween(vis, keep_from, &co.key_counts);
        } else {
          co.key_counts.assign(65536, 0);
        }
      } else if (keep_from != 0 || (continuing && req.carry_in->key_counts.size() == 65536)) {
        lz.KeyCountsBefore(keep_from, &co.key_counts);
      }
      co.stored.assign(all.begin() + keep_from, all.begin() + resume);
      for (uint8_t& f : co.stored) f &= 5;  // stored, and "stored as a masked position" (kFlagMasked; never set unless modelled)
      }
      if (tail_nbits) {
        uint8_t last = 0;
        dev_d2h(&last, (const uint8_t*)B.out_words + total_bytes, 1);
        co.tail_bits = last & ((1u << tail_nbits) - 1u);
        co.tail_nbits = tail_nbits;
      }
      if (req.keep_from_out) *req.keep_from_out = keep_from;
      *req.carry_out = std::move(co);
    } else if (req.keep_from_out) {
      *req.keep_from_out = 0;
    }
    stats.metablocks = n_mb;
    stats.commands = K;
    stats.literals = L;
    for (uint32_t m = 0; m < n_mb; ++m) stats.uncompressed_metablocks += descs[m].uncompressed;
    break;
  }
  if (!wrote_direct) {
    if (req.direct_out) {
      if (result.size() > req.direct_capacity) throw std::runtime_error("brotli_mi355x: output buffer too small");
      if (req.direct_out_on_device) {
        dev_h2d(req.direct_out, result.data(), result.size());
        dev_sync();
      } else {
        memcpy(req.direct_out, result.data(), result.size());
      }
      *req.direct_size = result.size();
    } else if (out->empty()) {
      out->swap(result);
    } else {
      out->insert(out->end(), result.begin(), result.end());
    }
  }
  stats.ms_total = total_clock.lap(false) + stats.ms_phase[0];
  stats.ms_metablock = 0;
  for (int i = 1; i < 9; ++i) stats.ms_metablock += stats.ms_phase[i];
  if (stats_out) *stats_out = stats;
}

}  // namespace brotli_mi355x
