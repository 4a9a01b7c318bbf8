// fragment_stream.cpp -- see fragment_stream.h
#include "fragment_stream.h"

#include <stdio.h>
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
// is a single fragment), all fragments of a batch side by side on the device (fragment_api.h); `finish`: the last one carries
// is_last.  They go in batches of whole fragments (about 256 MiB of input, at least one fragment): device memory stays bounded
// however much one call hands over.  Whole bytes are appended to *out, the open byte stays in *fs.
void RunFragments(const EncoderParams& p, FragmentStream* fs, const uint8_t* input, size_t size, bool finish, std::vector<uint8_t>* out) {
  const size_t block_size_limit = (size_t)1 << p.lgwin;
  static const size_t batch_target = getenv("BROTLI_MI355X_FRAGMENT_BATCH") ? (size_t)strtoull(getenv("BROTLI_MI355X_FRAGMENT_BATCH"), nullptr, 10) : ((size_t)256 << 20);
  static const bool selftest = getenv("BROTLI_MI355X_SELFTEST") != nullptr;
  static const bool test_again = getenv("BROTLI_MI355X_TEST_FRAGMENT_AGAIN") != nullptr;  // every fragment off phase 0 takes the one-by-one path
  // (at most 4096 fragments side by side: slabs and grid dimensions stay bounded when the fragments are tiny -- and at most as many as
  // 512 MiB of per-fragment scratch hold (hash table + command + literal buffers: about 1.1 MiB per fragment of >= 128 KiB): a 256 MiB
  // call at lgwin 16 otherwise asked for 4096 of them = 2.5 GiB, times the concurrent callers of the library)
  const size_t table_bytes = ((size_t)1 << TableBits(p.quality, std::min(size, block_size_limit))) * 4;
  const size_t scratch_per_fragment = table_bytes + (p.quality == 0 ? 0 : 5 * (std::min<size_t>(std::min(size, block_size_limit), (size_t)1 << 17) + 64));
  static const size_t scratch_budget = getenv("BROTLI_MI355X_FRAGMENT_SCRATCH") ? (size_t)strtoull(getenv("BROTLI_MI355X_FRAGMENT_SCRATCH"), nullptr, 10) : ((size_t)512 << 20);
  const size_t by_scratch = std::max<size_t>(1, scratch_budget / std::max<size_t>(1, scratch_per_fragment));
  const size_t per_batch = std::min<size_t>(std::min<size_t>(4096, by_scratch), std::max<size_t>(1, batch_target / block_size_limit));
  const size_t batch_bytes = per_batch * block_size_limit;
  const size_t in_cap = std::min(size, batch_bytes);
  const size_t max_jobs = in_cap == 0 ? 1 : (in_cap + block_size_limit - 1) / block_size_limit;
  auto slot_bytes = [](size_t in_size) { return (2 * in_size + 520 + 63) & ~(size_t)63; };
  const size_t slots_cap = 2 * in_cap + (520 + 64) * max_jobs + 64;
  const bool q0 = p.quality == 0;
  const size_t largest = std::min(in_cap, block_size_limit);  // (the largest fragment of the call)
  FragmentBuffers B;
  B.table_stride = (size_t)1 << TableBits(p.quality, largest);
  B.cmd_stride = std::min<size_t>(largest, (size_t)1 << 17) + 16;
  B.lit_stride = std::min<size_t>(largest, (size_t)1 << 17) + 64;
  DevBuf in(in_cap + 64, true), slots(slots_cap + 64), table(max_jobs * B.table_stride * 4 + 64),
      commands(q0 ? 64 : max_jobs * B.cmd_stride * 4 + 64), literals(q0 ? 64 : max_jobs * B.lit_stride + 64),
      states_a((max_jobs + 1) * sizeof(FragmentState) + 64), states_b(max_jobs * sizeof(FragmentState) + 64),
      jobs_dev(max_jobs * sizeof(FragmentJob) + 64), results_dev(max_jobs * sizeof(FragmentResult) + 64),
      pieces_dev(2 * max_jobs * sizeof(FragmentPiece) + 64);
  B.table = (uint32_t*)table.p;
  B.commands = q0 ? nullptr : (uint32_t*)commands.p;
  B.literals = q0 ? nullptr : (uint8_t*)literals.p;
  FragmentState* const sa = (FragmentState*)states_a.p;  // [0] the code the batch comes in with, [j + 1] what fragment j leaves behind (pass A)
  FragmentState* const sb = (FragmentState*)states_b.p;  // what fragment j leaves behind (pass B)
  size_t done = 0;
  bool more = true;
  std::vector<FragmentJob> jobs;
  std::vector<FragmentResult> results;
  std::vector<FragmentPiece> pieces;
  std::vector<uint8_t> bytes;
  while (more) {
    const size_t here = std::min(size - done, batch_bytes);
    if (here) dev_h2d_bulk(in.p, input + done, here);
    // ---- the fragments of this batch
    jobs.clear();
    size_t at = 0, slot_at = 0;
    for (;;) {
      const size_t block_size = std::min(block_size_limit, here - at);
      const bool is_last = (size - done - at == block_size) && finish;
      if (block_size == 0 && !is_last) break;
      FragmentJob job;
      job.in_offset = (uint32_t)at;
      job.in_size = (uint32_t)block_size;
      job.is_last = is_last ? 1u : 0u;
      job.table_bits = TableBits(p.quality, block_size);
      job.out_offset = slot_at;
      job.start_bits = 0;
      job.state_in = 0;
      jobs.push_back(job);
      slot_at += slot_bytes(block_size);
      at += block_size;
      if (is_last || at == here) break;
    }
    done += here;
    more = done < size;
    const uint32_t n = (uint32_t)jobs.size();
    if (n == 0) break;
    if (slot_at > slots_cap) throw std::runtime_error("brotli_mi355x: fragment output ran over its bound");
    // ---- side by side, each into its own slot from bit 0 on
    if (q0) {
      // the command code a fragment leaves behind is built from the commands of its own last block, whatever code it came in
      // with (compress_fragment.rs:1033-1044): pass A runs every fragment but the last with the batch's incoming code to learn
      // what each leaves behind, pass B runs them all with the right incoming codes
      dev_h2d(sa, &fs->state, sizeof(FragmentState));
      if (n > 1) {
        dev_h2d(jobs_dev.p, jobs.data(), (size_t)(n - 1) * sizeof(FragmentJob));
        frag_compress_batch(0, (const uint8_t*)in.p, (const FragmentJob*)jobs_dev.p, n - 1, B, sa, sa + 1, (FragmentResult*)results_dev.p, (uint8_t*)slots.p);
      }
      for (uint32_t j = 0; j < n; ++j) jobs[j].state_in = j;
    }
    dev_h2d(jobs_dev.p, jobs.data(), (size_t)n * sizeof(FragmentJob));
    frag_compress_batch(p.quality, (const uint8_t*)in.p, (const FragmentJob*)jobs_dev.p, n, B, sa, q0 ? sb : nullptr, (FragmentResult*)results_dev.p, (uint8_t*)slots.p);
    results.resize(n);
    dev_d2h(results.data(), results_dev.p, (size_t)n * sizeof(FragmentResult));
    if (q0 && selftest && n > 1) {
      std::vector<FragmentState> a(n + 1), b(n);
      dev_d2h(a.data(), sa, (size_t)(n + 1) * sizeof(FragmentState));
      dev_d2h(b.data(), sb, (size_t)n * sizeof(FragmentState));
      for (uint32_t j = 0; j + 1 < n; ++j)
        if (memcmp(&a[j + 1], &b[j], sizeof(FragmentState)) != 0) throw std::runtime_error("brotli_mi355x selftest: the command code a quality 0 fragment leaves behind depends on the code it came in with");
    }
    // ---- where the slots go in the stream.  A fragment's bits move with the phase up to its first jump to a byte boundary; what
    // comes behind lands on whole bytes.  The padding of that jump is the one thing of a fragment that depends on the phase, and it
    // reaches one decision -- "larger than stored raw?", which counts output bits: where the true phase turns that decision around,
    // the fragment is compressed again by itself at its true phase.
    pieces.clear();
    uint64_t cur = fs->last_bytes_bits;
    for (uint32_t j = 0; j < n; ++j) {
      FragmentResult r = results[j];
      if (r.bad) throw std::runtime_error("brotli_mi355x: fragment compressor failed");
      const uint32_t phase = (uint32_t)(cur & 7u);
      const uint64_t slot_bit = jobs[j].out_offset * 8;
      bool again = false;
      if (phase != 0 && r.decision_align != ~0ull) {
        const uint64_t a = r.decision_align;
        const uint64_t pad0 = (8 - (a & 7)) & 7, padt = (8 - ((phase + a) & 7)) & 7;
        const uint64_t total = r.decision_bits - pad0 + padt;
        const bool fall_back = total > 31 + ((uint64_t)jobs[j].in_size << 3);
        again = fall_back != (r.fell_back != 0);
      }
      if (test_again && phase != 0) again = true;
      if (again) {
        if (getenv("BROTLI_MI355X_DEBUG")) fprintf(stderr, "fragment %u of %u: compressed again at phase %u (the raw fall-back hangs on the padding)\n", j, n, phase);
        // (as fragment 0 of a batch of one that sits where fragment j sat: its table slab, its incoming code, its result slot)
        FragmentJob one = jobs[j];
        one.start_bits = phase;
        one.state_in = 0;
        dev_h2d((FragmentJob*)jobs_dev.p + j, &one, sizeof(FragmentJob));
        FragmentBuffers Bj = B;
        Bj.table += (size_t)j * B.table_stride;
        if (Bj.commands) Bj.commands += (size_t)j * B.cmd_stride;
        if (Bj.literals) Bj.literals += (size_t)j * B.lit_stride;
        frag_compress_batch(p.quality, (const uint8_t*)in.p, (const FragmentJob*)jobs_dev.p + j, 1, Bj, sa + j, q0 ? sb + j : nullptr, (FragmentResult*)results_dev.p + j,
                            (uint8_t*)slots.p);
        dev_d2h(&r, (FragmentResult*)results_dev.p + j, sizeof(FragmentResult));
        if (r.bad) throw std::runtime_error("brotli_mi355x: fragment compressor failed");
        pieces.push_back({slot_bit + phase, cur, r.end_bits - phase});
        cur += r.end_bits - phase;
        continue;
      }
      if (r.first_align == ~0ull) {
        pieces.push_back({slot_bit, cur, r.end_bits});
        cur += r.end_bits;
      } else {
        const uint64_t a = r.first_align, a8 = (a + 7) & ~(uint64_t)7;
        if (a) pieces.push_back({slot_bit, cur, a});
        const uint64_t aligned = (cur + a + 7) & ~(uint64_t)7;
        if (r.end_bits > a8) pieces.push_back({slot_bit + a8, aligned, r.end_bits - a8});
        cur = aligned + (r.end_bits - a8);
      }
    }
    if (q0) dev_d2h(&fs->state, sb + (n - 1), sizeof(FragmentState));
    // ---- the join
    const uint64_t ix = cur;
    const size_t joined_bytes = (size_t)(ix >> 3) + 2;
    DevBuf joined(((joined_bytes + 7) & ~(size_t)7) + 64, true);
    uint8_t head[2] = {(uint8_t)fs->last_bytes, (uint8_t)(fs->last_bytes >> 8)};
    dev_h2d(joined.p, head, 2);
    if (!pieces.empty()) {
      dev_h2d(pieces_dev.p, pieces.data(), pieces.size() * sizeof(FragmentPiece));
      frag_join((const uint8_t*)slots.p, (const FragmentPiece*)pieces_dev.p, (uint32_t)pieces.size(), (uint8_t*)joined.p);
    }
    bytes.resize(joined_bytes);
    dev_d2h_bulk(bytes.data(), joined.p, bytes.size());
    out->insert(out->end(), bytes.begin(), bytes.begin() + (ptrdiff_t)(ix >> 3));
    fs->last_bytes = (uint16_t)(bytes[(size_t)(ix >> 3)] | (bytes[(size_t)(ix >> 3) + 1] << 8));
    fs->last_bytes_bits = (uint8_t)(ix & 7);
    if (fs->last_bytes_bits != 0) fs->last_bytes &= (uint16_t)((1u << fs->last_bytes_bits) - 1u); else fs->last_bytes = 0;
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
      fs->input_seen += n;
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
        fs->flushed_raw += n;
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

bool FragmentRingMetadataReturns(const FragmentStream& fs) {
  const uint64_t will_go_raw = fs.first_mb != 3 ? std::min<uint64_t>(2, fs.pending.size()) : 0;
  return fs.input_seen == fs.flushed_raw + will_go_raw;
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
