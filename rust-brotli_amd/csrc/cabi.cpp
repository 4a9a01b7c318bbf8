// cabi.cpp -- the C ABI of include/brotli_mi355x.h (drop-in for the encoder half of the reference's cdylib:
// src/ffi/compressor.rs, src/ffi/multicompress/mod.rs).
#include "../../include/brotli_mi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <stdexcept>
#include <string>
#include <vector>

#include "concat.h"
#include "device_api.h"
#include "encoder.h"
#include "encoder_params.h"
#include "fragment_stream.h"

using namespace brotli_mi355x;

namespace {

thread_local std::string g_last_error;

void SetError(const char* where, const char* what) {
  g_last_error = std::string(where) + ": " + what;
  fprintf(stderr, "brotli_mi355x: %s\n", g_last_error.c_str());
}

enum StreamState { kProcessing = 0, kFlushRequested = 1, kFinished = 2 };

}  // namespace

struct BrotliEncoderStateStruct {
  brotli_alloc_func alloc_func;
  brotli_free_func free_func;
  void* opaque;
  EncoderParams params;
  bool initialized;      // set_parameter is refused afterwards (encode.rs:289-295)
  bool first_encode_seen;
  bool failed;
  StreamState stream_state;
  std::vector<uint8_t> input;
  std::vector<uint8_t> dictionary;
  bool has_dictionary;
  bool dictionary_in_window;  // the custom dictionary has been put in front of `input` as the first piece's window
  size_t size_hint_at_dictionary;  // params.size_hint when BrotliEncoderSetCustomDictionary ran
  std::vector<uint8_t> output;
  size_t output_pos;
  uint64_t total_out;
  uint64_t total_in;
  // `input` = the window of the stream so far that the next piece needs as its LZ77 prefix ([0, encoded_upto), already in
  // the output) followed by the input that has not been encoded yet.  Older bytes are dropped (bounded memory).
  size_t encoded_upto;
  StreamCarry carry;
  FragmentStream fragments;  // qualities 0 and 1: the stream never enters the ring-buffer path (fragment_stream.h)
  bool metadata_draining;  // an EMIT_METADATA operation whose output has not been fully taken yet
  size_t next_batch_try;   // PROCESS: do not try another partial piece before this much input is pending
};

struct BrotliEncoderWorkPoolStruct {
  brotli_alloc_func alloc_func;
  brotli_free_func free_func;
  void* opaque;
};

namespace {

void EnsureInitialized(BrotliEncoderState* s) {
  if (s->initialized) return;
  s->initialized = true;
}

size_t BlockSize(const EncoderParams& user) {
  EncoderParams p = user;
  FinalizeParams(&p);
  return (size_t)1 << p.lgblock;
}

// How much input BROTLI_OPERATION_PROCESS lets pile up before it encodes a piece of the stream on its own (the device wants
// big pieces; the reference encodes whenever a 64 KiB block is full, encode.rs:2959-2964).
size_t StreamBatchBytes() {
  static const size_t v = getenv("BROTLI_MI355X_STREAM_BATCH") ? (size_t)strtoull(getenv("BROTLI_MI355X_STREAM_BATCH"), nullptr, 10) : ((size_t)64 << 20);
  return v;
}

// Runs the device encoder over the input buffered since the last piece.  FLUSH: finish = false, FINISH: finish = true;
// partial (PROCESS): only the meta-blocks that close by themselves within the whole input blocks buffered so far.
bool EncodeBuffered(BrotliEncoderState* s, bool finish, bool emit_metadata = false, size_t metadata_size = 0, bool partial = false,
                    bool last_block_processed_early = false) {
  try {
    EncodeRequest req;
    req.params = s->params;
    req.last_block_processed_early = last_block_processed_early;
    const bool flushed_stream = !finish || s->carry.valid;
    const bool first_piece = !s->carry.valid;
    size_t dict_use = 0;
    if (s->has_dictionary && first_piece) {
      // set_custom_dictionary (encode.rs:1196-1270): last (1 << lgwin) - 16 bytes
      EncoderParams p = s->params;
      FinalizeParams(&p);
      const size_t max_dict = ((size_t)1 << p.lgwin) - 16;
      dict_use = std::min(s->dictionary.size(), max_dict);
      if (flushed_stream && !s->dictionary_in_window) {
        // a stream that goes on after this piece: the dictionary is the first stretch of its window
        s->input.insert(s->input.begin(), s->dictionary.end() - (ptrdiff_t)dict_use, s->dictionary.end());
        s->encoded_upto += dict_use;
        s->dictionary_in_window = true;
      }
    }
    req.input = s->input.data() + s->encoded_upto;
    req.input_size = s->input.size() - s->encoded_upto;
    if (partial) {
      const size_t block = BlockSize(s->params);
      req.input_size = req.input_size / block * block;
      if (req.input_size == 0) return true;
    }
    req.finish = finish;
    req.partial = partial;
    req.emit_metadata = emit_metadata;
    req.metadata_size = metadata_size;
    size_t consumed = req.input_size, keep_from = 0;
    req.consumed_out = &consumed;
    req.keep_from_out = &keep_from;
    if (flushed_stream) {
      req.prefix = s->encoded_upto ? s->input.data() : nullptr;
      req.prefix_size = s->encoded_upto;
      req.prefix_is_file_continuation = !(s->has_dictionary && first_piece);
      req.carry_in = &s->carry;
      req.carry_out = &s->carry;
    }
    if (s->has_dictionary && first_piece) {
      // no static dictionary, hasher chosen before any size hint, prev bytes stay 0 (encode.rs:1196-1270)
      if (!flushed_stream) {
        req.prefix = s->dictionary.data() + (s->dictionary.size() - dict_use);
        req.prefix_size = dict_use;
        req.prefix_is_file_continuation = false;
      }
      req.hasher_chosen_before_size_hint = true;
      req.has_hasher_size_hint = true;
      req.hasher_size_hint = s->size_hint_at_dictionary;
      req.params.use_dictionary = false;
    }
    // output not yet taken by the caller stays in front
    if (s->output_pos != 0) {
      s->output.erase(s->output.begin(), s->output.begin() + (ptrdiff_t)s->output_pos);
      s->output_pos = 0;
    }
    EncodeStream(req, &s->output, nullptr);
    s->encoded_upto += consumed;
    if (keep_from != 0) {
      // the next piece only needs a window of what has been encoded: forget the rest
      s->input.erase(s->input.begin(), s->input.begin() + (ptrdiff_t)keep_from);
      s->encoded_upto -= keep_from;
    }
    if (partial) {
      // nothing closed: wait for (a good deal) more input before trying again
      const size_t pending = s->input.size() - s->encoded_upto;
      s->next_batch_try = consumed == 0 ? pending * 2 : 0;
    }
    return true;
  } catch (const std::exception& e) {
    SetError("BrotliEncoderCompressStream", e.what());
    s->failed = true;
    return false;
  }
}

size_t AvailableOut(const BrotliEncoderState* s) { return s->output.size() - s->output_pos; }

// One chunk of a compress_multi job: compress_part, src/enc/threading/mod.rs:337-411
void CompressChunk(const EncoderParams& user_params, const uint8_t* input, size_t input_size, bool input_on_device,
                   size_t thread_index, size_t num_threads, std::vector<uint8_t>* out, const std::vector<uint8_t>* host_copy) {
  const size_t start = (thread_index * input_size) / num_threads;
  const size_t end = ((thread_index + 1) * input_size) / num_threads;
  EncodeRequest req;
  req.params = user_params;
  if (thread_index != 0) {
    req.params.catable = true;
    req.params.magic_number = false;
  }
  req.params.appendable = true;
  req.input = input + start;
  req.input_size = end - start;
  req.input_on_device = input_on_device;
  std::vector<uint8_t> prefix_host;
  if (thread_index != 0) {
    // set_custom_dictionary_with_optional_precomputed_hasher(range.start, input[..range.start], _, true)
    req.params.use_dictionary = false;
    EncoderParams p = req.params;
    FinalizeParams(&p);
    const size_t max_dict = ((size_t)1 << p.lgwin) - 16;
    const size_t use = std::min(start, max_dict);
    if (input_on_device) {
      prefix_host.resize(use);
      dev_d2h(prefix_host.data(), input + (start - use), use);
      req.prefix = prefix_host.data();
    } else {
      req.prefix = input + (start - use);
    }
    req.prefix_size = use;
    if (start <= 1) {
      // too short to prime: catable + appendable without a dictionary (encode.rs:1237-1241); prefix pointer
      // stays non-null so that EncodeStream applies that rule
      static const uint8_t dummy[1] = {0};
      req.prefix = start == 0 ? dummy : req.prefix;
      req.prefix_size = start;
    }
    req.prefix_is_file_continuation = true;
    req.hasher_chosen_before_size_hint = start > 1;  // a too-short dictionary returns before hasher_setup
  }
  (void)host_copy;
  if (IsFragmentStream(req.params) || IsFragmentRing(req.params)) {
    // qualities 0 / 1: a shard is one FINISH call on a stream of its own -- the first one on the fragment path, the others catable
    // (set_custom_dictionary leaves them without a dictionary at these qualities, encode.rs:1237-1241) and so on the ring-buffer path
    if (input_on_device) throw std::runtime_error("brotli_mi355x: qualities 0 and 1 take their input from host memory");
    FragmentStream fs;
    if (IsFragmentRing(req.params)) FragmentRingCompress(req.params, &fs, req.input, req.input_size, true, false, out);
    else FragmentStreamCompress(req.params, &fs, req.input, req.input_size, true, false, out);
    // compress_part gives a shard BrotliEncoderMaxCompressedSize(its length) of room; its compress_stream loop returns once that
    // buffer is full without asking whether the stream is finished (threading/mod.rs:337-411), so the reference hands back a
    // TRUNCATED shard where the stream does not fit -- small fragments of these qualities on incompressible input do not.
    // Refused here instead of reproduced.
    if (out->size() > MaxCompressedSize(req.input_size))
      throw std::runtime_error("brotli_mi355x: refused: a shard of quality 0 / 1 outgrows BrotliEncoderMaxCompressedSize of its length (the reference hands back a truncated shard here: compress_part stops when its fixed buffer is full)");
    return;
  }
  EncodeStream(req, out, nullptr);
}

// Host threads that keep several chunks of one BrotliEncoderCompressMulti call in flight -- on every device of the node.
// A chunk's LZ77 rounds leave its device idle while the host resolves them, and its late rounds run a handful of wavefronts:
// two or three chunks side by side fill those gaps.  The reference's multi API fans the shards of ONE call over all its workers
// (src/ffi/multicompress/mod.rs:93, threading/mod.rs:413-453); here the workers are host threads bound to HIP devices: shard t
// goes to device slot t % (number of slots) -- by default one slot per visible device (hipGetDeviceCount), so a C / Rust / Go
// caller of BrotliEncoderCompressMulti on an 8-GPU node uses all eight with no collective (host buffers in, D2H per device out;
// SURVEY 8e).  BROTLI_MI355X_DEVICES: "current" = the calling thread's device only (one process per GPU under a launcher);
// "0,1,2" = these devices; an entry may repeat ("0,0": two slots on one device -- the multi-device code path on a 1-GPU box).
// The helpers live as long as the process, each bound to the device of its slot for good (their device memory pools, streams
// and events are per thread, hence per device, and are reused from call to call); BROTLI_MI355X_SHARD_WORKERS sets how many
// chunks run at once over all slots (default 8, at least one per slot; 1 with a single slot = one after the other on the calling thread).
class ShardWorkers {
 public:
  static ShardWorkers& Get() {
    // (never destroyed: the helpers wait on its condition variable until the process ends, and tearing down their
    // device memory pools from a static destructor would race the runtime's own shutdown)
    static ShardWorkers* w = new ShardWorkers;
    return *w;
  }
  // the device slots of this process (see above); empty = "current": whatever device the calling thread is on
  static const std::vector<int>& Slots() {
    static const std::vector<int> slots = [] {
      std::vector<int> v;
      const char* e = getenv("BROTLI_MI355X_DEVICES");
      const int visible = dev_device_count();
      if (e && !strcmp(e, "current")) return v;
      if (e && *e && strcmp(e, "all")) {
        for (const char* p = e; *p;) {
          char* end = nullptr;
          const long d = strtol(p, &end, 10);
          if (end == p) break;
          if (d >= 0 && d < visible) v.push_back((int)d);
          p = *end == ',' ? end + 1 : end;
        }
        return v;  // (nothing usable in the list: the calling thread's device)
      }
      if (visible > 1)
        for (int d = 0; d < visible; ++d) v.push_back(d);
      return v;
    }();
    return slots;
  }
  // runs job(0) ... job(count - 1), each exactly once, on the calling thread and the helpers; rethrows the first exception
  // light: every job is one wavefront on a table of its own (qualities 2 .. 4, quick_device.h) -- all of them side by side
  void Run(size_t count, const std::function<void(size_t)>& job, bool light = false) {
    // (BROTLI_MI355X_SHARD_WORKERS, when set, bounds the light jobs as well: it is the caller's handle on device memory)
    const size_t limit = std::min(count, light && WorkerCount() > 1 && !WorkerCountSetByUser() ? (size_t)16 : WorkerCount());
    const int here = dev_current_device();
    std::vector<int> slots = Slots();
    if (slots.empty()) slots.push_back(here);
    const size_t nslots = std::min(slots.size(), count);
    if (nslots == 1 && slots[0] == here && limit <= 1) {
      for (size_t i = 0; i < count; ++i) job(i);
      return;
    }
    std::lock_guard<std::mutex> one_call_at_a_time(run_mu_);
    Batch b;
    b.job = &job;
    b.q = std::vector<SlotQueue>(nslots);
    for (size_t t = 0; t < count; ++t) b.q[t % nslots].jobs.push_back(t);
    const size_t per_slot = std::max<size_t>(1, (limit + nslots - 1) / nslots);
    // the calling thread works in the first slot that sits on its own device (its pool and stream belong to that device)
    size_t my_slot = nslots;
    for (size_t s = 0; s < nslots && my_slot == nslots; ++s)
      if (slots[s] == here) my_slot = s;
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (threads_of_slot_.size() < slots.size()) threads_of_slot_.resize(slots.size(), 0);
      for (size_t s = 0; s < nslots; ++s) {
        const size_t want = std::min(per_slot, b.q[s].jobs.size()) - (s == my_slot ? 1 : 0);
        while (threads_of_slot_[s] < want) {
          const int device = slots[s];
          const size_t ordinal = threads_of_slot_[s];
          std::thread([this, s, device, ordinal] { Loop(s, device, ordinal); }).detach();
          ++threads_of_slot_[s];
        }
        b.q[s].helpers_wanted = want;
        b.q[s].helpers_seated = want;
      }
      batch_ = &b;
      ++generation_;
    }
    cv_.notify_all();
    if (my_slot < nslots) Work(&b, my_slot);
    {
      std::unique_lock<std::mutex> lock(mu_);
      done_cv_.wait(lock, [&] {
        if (b.helpers_in != 0) return false;
        for (auto& q : b.q)
          if (q.helpers_wanted != 0) return false;
        return true;
      });
      batch_ = nullptr;
    }
    if (b.error) std::rethrow_exception(b.error);
  }

 private:
  struct SlotQueue {
    std::vector<size_t> jobs;      // the shards dealt to this slot, in order
    std::atomic<size_t> next{0};
    size_t helpers_wanted = 0;     // (under mu_) seats not taken yet
    size_t helpers_seated = 0;     // seats of this batch: the helpers 0 .. helpers_seated - 1 of the slot take them
    SlotQueue() = default;
    SlotQueue(const SlotQueue& o) : jobs(o.jobs), next(o.next.load()), helpers_wanted(o.helpers_wanted), helpers_seated(o.helpers_seated) {}
  };
  struct Batch {
    const std::function<void(size_t)>* job = nullptr;
    std::vector<SlotQueue> q;
    size_t helpers_in = 0;       // (under mu_)
    std::exception_ptr error;    // (under mu_)
  };
  static bool WorkerCountSetByUser() {
    static const bool set = getenv("BROTLI_MI355X_SHARD_WORKERS") != nullptr;
    return set;
  }
  static size_t WorkerCount() {
    static const size_t n = [] {
      const char* e = getenv("BROTLI_MI355X_SHARD_WORKERS");
      const long v = e ? atol(e) : 8;
      return (size_t)std::min<long>(std::max<long>(v, 1), 8);
    }();
    return n;
  }
  void Work(Batch* b, size_t slot) {
    SlotQueue& q = b->q[slot];
    for (;;) {
      const size_t i = q.next.fetch_add(1);
      if (i >= q.jobs.size()) return;
      try {
        (*b->job)(q.jobs[i]);
      } catch (...) {
        std::lock_guard<std::mutex> lock(mu_);
        if (!b->error) b->error = std::current_exception();
        for (auto& other : b->q) other.next.store(other.jobs.size());  // nobody starts another chunk
      }
    }
  }
  // A call that needs fewer helpers than the slot has (sixteen light shards one call, eight heavy ones the next) is served by the
  // SAME helpers every time, the lowest-numbered ones: a helper's device memory pool is its own, and with whichever threads woke
  // first taking the seats every call of bench.py's 1 GiB / 8 shards found two or three cold pools (40-odd hipMalloc calls, 0.7-1.1 s
  // of a 1.3 s call, call after call).
  void Loop(size_t slot, int device, size_t ordinal) {
    uint64_t seen = 0;
    bool bound = false;
    for (;;) {
      Batch* b = nullptr;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] {
          return generation_ != seen && batch_ != nullptr && slot < batch_->q.size() && ordinal < batch_->q[slot].helpers_seated && batch_->q[slot].helpers_wanted > 0;
        });
        b = batch_;
        b->q[slot].helpers_wanted--;
        seen = generation_;  // (a helper that has taken its place in this batch does not take a second one)
        b->helpers_in++;
      }
      try {
        if (!bound) {
          dev_use_device(device);  // for good: this thread's pool, stream and events live on this device
          bound = true;
        }
        Work(b, slot);
      } catch (...) {
        std::lock_guard<std::mutex> lock(mu_);
        if (!b->error) b->error = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> lock(mu_);
        b->helpers_in--;
      }
      done_cv_.notify_all();
    }
  }
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<size_t> threads_of_slot_;
  Batch* batch_ = nullptr;
  uint64_t generation_ = 0;
};

// A small one-shot call is a chain of about 150 launches and copies whatever its size, and the host side of the runtime queues
// them one thread at a time: measured on alice29 at quality 5 (tools/thread_trace.py, 40 calls per thread) 63 / 108 / 153 MB/s with
// 1 / 2 / 4 threads -- and 139 / 112 / 92 / 60 MB/s with 8 / 16 / 32 / 64, the threads queueing behind one another inside the runtime.
// Calls of up to 1 MiB therefore take one of four seats (BROTLI_MI355X_SMALL_CALLS_IN_FLIGHT, 0 = no gate); larger calls are bound
// by the device and are not gated.
class SmallCallGate {
 public:
  explicit SmallCallGate(size_t input_size) : held_(input_size <= ((size_t)1 << 20) && Seats() != 0) {
    if (!held_) return;
    std::unique_lock<std::mutex> lock(Mu());
    Cv().wait(lock, [] { return InFlight() < Seats(); });
    ++InFlight();
  }
  ~SmallCallGate() {
    if (!held_) return;
    {
      std::lock_guard<std::mutex> lock(Mu());
      --InFlight();
    }
    Cv().notify_one();
  }
  SmallCallGate(const SmallCallGate&) = delete;
  SmallCallGate& operator=(const SmallCallGate&) = delete;

 private:
  static size_t Seats() {
    static const size_t n = getenv("BROTLI_MI355X_SMALL_CALLS_IN_FLIGHT") ? (size_t)atol(getenv("BROTLI_MI355X_SMALL_CALLS_IN_FLIGHT")) : 4;
    return n;
  }
  // (never destroyed: calls may still be in flight on other threads when the process ends)
  static std::mutex& Mu() { static std::mutex* m = new std::mutex; return *m; }
  static std::condition_variable& Cv() { static std::condition_variable* c = new std::condition_variable; return *c; }
  static size_t& InFlight() { static size_t n = 0; return n; }
  const bool held_;
};

bool ParamsFromLists(size_t num_params, const BrotliEncoderParameter* keys, const uint32_t* values, EncoderParams* p) {
  for (size_t i = 0; i < num_params; ++i)
    if (!SetParameter(p, (int)keys[i], values[i])) return false;
  return true;
}

int32_t CompressMultiImpl(size_t num_params, const BrotliEncoderParameter* keys, const uint32_t* values, size_t input_size,
                          const uint8_t* input, size_t* encoded_size, uint8_t* encoded, size_t desired_num_threads) {
  if (desired_num_threads == 0) return 0;
  const size_t num_threads = std::min<size_t>(desired_num_threads, 16);  // MAX_THREADS, multicompress/mod.rs:26
  try {
    std::vector<uint8_t> out;
    if (num_threads == 1) {
      // help_brotli_encoder_compress_single, multicompress/mod.rs:57-91: invalid parameters are ignored
      EncodeRequest req;
      for (size_t i = 0; i < num_params; ++i) SetParameter(&req.params, (int)keys[i], values[i]);
      req.input = input;
      req.input_size = input_size;
      if (IsFragmentStream(req.params) || IsFragmentRing(req.params)) {
        FragmentStream fs;
        if (IsFragmentRing(req.params)) FragmentRingCompress(req.params, &fs, input, input_size, true, false, &out);
        else FragmentStreamCompress(req.params, &fs, input, input_size, true, false, &out);
      } else {
        EncodeStream(req, &out, nullptr);
      }
    } else {
      EncoderParams params;
      if (!ParamsFromLists(num_params, keys, values, &params)) return 0;
      // The helper threads allocate from pools of their own: room for them while the device is idle -- about 100 bytes of scratch per
      // input byte of every shard in flight.  What the calling thread keeps pooled from earlier one-shot calls
      // goes back to the driver when it is clearly more than its own share of this call (it works on one shard itself).  Round 6: with
      // 128 GiB pooled by the calling thread the helpers' hipMalloc calls took 70 ms each and ended in trims of every pool: 5-9 s for
      // a call of 1.4 s, call after call.
      {
        const size_t in_flight = std::min<size_t>(num_threads, 8);
        const size_t shard_bytes = input_size / num_threads + 1;
        const size_t per_shard = shard_bytes * 110 + ((size_t)256 << 20);
        dev_make_room_for(per_shard * in_flight, per_shard);
      }
      if (params.favor_cpu_efficiency) {
        // threading/mod.rs:456-542: one hasher is filled with the whole input in front of every shard and handed to its
        // encoder instead of priming it from the shard's prefix.  While a shard starts within the window that is the same
        // table (the reference asserts it in debug builds, encode.rs:1249-1268) and the same stream.  Further in, the shared
        // table holds FILE offsets where the shard encoder works with ring-buffer positions (its prefix is cut to the
        // window, encode.rs:1243-1246): the reference's own assertion fails there and its release build searches junk
        // candidates.  That state is not reproduced; such a call is refused rather than answered with other bytes.
        const int lgwin = std::min(std::max(params.lgwin, 10), params.large_window ? 30 : 24);
        const size_t last_start = (num_threads - 1) * input_size / num_threads;
        if (last_start > ((size_t)1 << lgwin) - 16)
          throw std::runtime_error("favor_cpu_efficiency with shards that start beyond the window is not supported (the reference's shared hasher is inconsistent there)");
      }
      // the chunks are independent streams (compress_multi hands them to a worker pool, threading/mod.rs:333-453):
      // a few of them are in flight on the device at a time, each on the stream of its host thread
      std::vector<std::vector<uint8_t>> chunks(num_threads);
      EncoderParams fin = params;
      FinalizeParams(&fin);
      ShardWorkers::Get().Run(num_threads, [&](size_t t) { CompressChunk(params, input, input_size, false, t, num_threads, &chunks[t], nullptr); },
                              fin.quality >= 2 && fin.quality < 5);
      // stitched straight into the caller's buffer
      ChunkStitcher stitcher;
      ByteSink sink(encoded, *encoded_size);
      for (size_t t = 0; t < num_threads; ++t) {
        if (!stitcher.Append(chunks[t].data(), chunks[t].size(), &sink)) throw std::runtime_error("chunk cannot be concatenated");
        std::vector<uint8_t>().swap(chunks[t]);
      }
      stitcher.Finish(&sink);
      if (sink.overflow()) {
        SetError("BrotliEncoderCompressMulti", "insufficient output space");
        return 0;
      }
      *encoded_size = sink.size();
      return 1;
    }
    if (out.size() > *encoded_size) {
      SetError("BrotliEncoderCompressMulti", "insufficient output space");
      return 0;
    }
    memcpy(encoded, out.data(), out.size());
    *encoded_size = out.size();
    return 1;
  } catch (const std::exception& e) {
    SetError("BrotliEncoderCompressMulti", e.what());
    return 0;
  }
}

// MakeUncompressedStream, encode.rs:1388-1433
size_t MakeUncompressedStream(const uint8_t* input, size_t input_size, uint8_t* output) {
  size_t size = input_size, result = 0, offset = 0;
  if (input_size == 0) {
    output[0] = 6;
    return 1;
  }
  output[result++] = 0x21;
  output[result++] = 0x03;
  while (size > 0) {
    uint32_t nibbles = 0;
    const uint32_t chunk_size = size > (1u << 24) ? (1u << 24) : (uint32_t)size;
    if (chunk_size > (1u << 16)) nibbles = chunk_size > (1u << 20) ? 2 : 1;
    const uint32_t bits = (nibbles << 1) | ((chunk_size - 1) << 3) | (1u << (19 + 4 * nibbles));
    output[result++] = (uint8_t)bits;
    output[result++] = (uint8_t)(bits >> 8);
    output[result++] = (uint8_t)(bits >> 16);
    if (nibbles == 2) output[result++] = (uint8_t)(bits >> 24);
    memcpy(&output[result], &input[offset], chunk_size);
    result += chunk_size;
    offset += chunk_size;
    size -= chunk_size;
  }
  output[result++] = 3;
  return result;
}

}  // namespace

extern "C" {

BrotliEncoderState* BrotliEncoderCreateInstance(brotli_alloc_func alloc_func, brotli_free_func free_func, void* opaque) {
  if ((alloc_func == nullptr) != (free_func == nullptr)) {  // both or neither (compressor.rs:89-98)
    SetError("BrotliEncoderCreateInstance", "alloc_func and free_func must be given together");
    return nullptr;
  }
  void* mem = alloc_func ? alloc_func(opaque, sizeof(BrotliEncoderStateStruct)) : malloc(sizeof(BrotliEncoderStateStruct));
  if (!mem) return nullptr;
  BrotliEncoderState* s = new (mem) BrotliEncoderStateStruct();
  s->alloc_func = alloc_func;
  s->free_func = free_func;
  s->opaque = opaque;
  s->initialized = false;
  s->first_encode_seen = false;
  s->failed = false;
  s->stream_state = kProcessing;
  s->has_dictionary = false;
  s->dictionary_in_window = false;
  s->size_hint_at_dictionary = 0;
  s->output_pos = 0;
  s->total_out = 0;
  s->total_in = 0;
  s->encoded_upto = 0;
  s->next_batch_try = 0;
  s->metadata_draining = false;
  return s;
}

BROTLI_BOOL BrotliEncoderSetParameter(BrotliEncoderState* state, BrotliEncoderParameter p, uint32_t value) {
  if (!state || state->initialized) return BROTLI_FALSE;
  return SetParameter(&state->params, (int)p, value) ? BROTLI_TRUE : BROTLI_FALSE;
}

void BrotliEncoderDestroyInstance(BrotliEncoderState* state) {
  if (!state) return;
  brotli_free_func free_func = state->free_func;
  void* opaque = state->opaque;
  state->~BrotliEncoderStateStruct();
  if (free_func) {
    free_func(opaque, state);
  } else {
    free(state);
  }
}

size_t BrotliMi355xTrimPool(void) {
  try {
    return dev_trim_pool();
  } catch (const std::exception& e) {
    SetError("BrotliMi355xTrimPool", e.what());
    return 0;
  }
}

size_t BrotliEncoderMaxCompressedSize(size_t input_size) { return MaxCompressedSize(input_size); }
size_t BrotliEncoderMaxCompressedSizeMulti(size_t input_size, size_t num_threads) { return MaxCompressedSizeMulti(input_size, num_threads); }
uint32_t BrotliEncoderVersion(void) { return 0x01000f01u; }

BROTLI_BOOL BrotliEncoderCompressStream(BrotliEncoderState* s, BrotliEncoderOperation op, size_t* available_in,
                                        const uint8_t** next_in, size_t* available_out, uint8_t** next_out,
                                        size_t* total_out) {
  if (!s || s->failed) return BROTLI_FALSE;
  EnsureInitialized(s);
  if (op == BROTLI_OPERATION_EMIT_METADATA) {
    // process_metadata, encode.rs:2579-2685: the bytes handed over are not input but the payload of a metadata block;
    // pending input is flushed in front of it (without the byte-alignment block: the metadata header follows directly)
    if (s->stream_state != kProcessing || *available_in > ((size_t)1 << 24)) return BROTLI_FALSE;
    if (s->metadata_draining) {
      if (*available_in != 0) return BROTLI_FALSE;  // (the reference insists on the rest of the same payload)
    } else {
    // (update_size_hint(0), encode.rs:2904: a hint that comes out 0 -- nothing received yet -- stays "unset" and is tried again
    // at the next encode_data)
    s->first_encode_seen = true;
    if (s->params.size_hint == 0) s->params.size_hint = (size_t)std::min<uint64_t>(s->total_in, (uint64_t)1 << 30);
    const size_t n_meta = *available_in;
    if (IsFragmentStream(s->params) || IsFragmentRing(s->params)) {
      // (qualities 0 / 1: nothing is ever pending in a ring buffer; the header goes behind the open byte, encode.rs:2630-2640)
      try {
        // A catable stream at these qualities that has seen input: the reference's process_metadata asks encode_data to flush
        // "pending" input for ever -- its quality 0 / 1 branch never moves last_flush_pos_ (encode.rs:2335-2389, 2621-2629) -- and
        // does not return.  Nothing to reproduce.
        // (It does return while everything received so far goes out as the raw first bytes of the stream -- at most the first
        // two or three bytes: those move last_flush_pos_.)  The refusal leaves the state as it was: nothing has been done.
        if (IsFragmentRing(s->params) && !FragmentRingMetadataReturns(s->fragments)) {
          SetError("BrotliEncoderCompressStream", "a metadata block behind input on a catable stream of quality 0 / 1: the reference encoder does not return from this call");
          return BROTLI_FALSE;
        }
        if (s->output_pos != 0) {
          s->output.erase(s->output.begin(), s->output.begin() + (ptrdiff_t)s->output_pos);
          s->output_pos = 0;
        }
        // encode_data(is_last = false, force_flush = true) on what is pending: the raw first bytes (nothing else is, see above)
        if (IsFragmentRing(s->params) && !s->fragments.pending.empty()) FragmentRingCompress(s->params, &s->fragments, nullptr, 0, false, true, &s->output);
        FragmentStreamMetadataHeader(s->params, &s->fragments, n_meta, &s->output);
      } catch (const std::exception& e) {
        SetError("BrotliEncoderCompressStream", e.what());
        s->failed = true;
        return BROTLI_FALSE;
      }
    } else if (!EncodeBuffered(s, false, true, n_meta)) return BROTLI_FALSE;
    s->output.insert(s->output.end(), *next_in, *next_in + n_meta);
    *next_in += n_meta;
    *available_in = 0;
    s->metadata_draining = true;
    }
  }
  if (s->stream_state != kProcessing && *available_in != 0) return BROTLI_FALSE;  // encode.rs:2918-2922
  if (s->stream_state == kProcessing && op != BROTLI_OPERATION_EMIT_METADATA && (IsFragmentStream(s->params) || IsFragmentRing(s->params))) {
    // BrotliEncoderCompressStreamFast, encode.rs:2706-2861: the input of THIS call, cut into fragments of at most 1 << lgwin bytes,
    // is compressed right away (fragment_stream.h); nothing is buffered
    try {
      if (s->output_pos != 0) {
        s->output.erase(s->output.begin(), s->output.begin() + (ptrdiff_t)s->output_pos);
        s->output_pos = 0;
      }
      const size_t n = *available_in;
      if (IsFragmentRing(s->params)) {
        // catable (also: after BrotliEncoderSetCustomDictionary): the ring-buffer path, block by block (encode.rs:2335-2389)
        FragmentRingCompress(s->params, &s->fragments, *next_in, n, op == BROTLI_OPERATION_FINISH, op == BROTLI_OPERATION_FLUSH, &s->output);
      } else {
        FragmentStreamCompress(s->params, &s->fragments, *next_in, n, op == BROTLI_OPERATION_FINISH, op == BROTLI_OPERATION_FLUSH, &s->output);
      }
      s->total_in += n;
      *next_in += n;
      *available_in = 0;
      if (op == BROTLI_OPERATION_FINISH) s->stream_state = kFinished;
    } catch (const std::exception& e) {
      SetError("BrotliEncoderCompressStream", e.what());
      s->failed = true;
      return BROTLI_FALSE;
    }
  } else if (s->stream_state == kProcessing && op != BROTLI_OPERATION_EMIT_METADATA) {
    // FINISH without input right behind a full input block: the reference has run that block through encode_data already
    // (it does so as soon as a block is full, encode.rs:2959-2964), not knowing it was the last one
    const bool early_last = op == BROTLI_OPERATION_FINISH && *available_in == 0 && s->total_in != 0 && (s->total_in % BlockSize(s->params)) == 0;
    if (*available_in != 0) {
      s->input.insert(s->input.end(), *next_in, *next_in + *available_in);
      s->total_in += *available_in;
      *next_in += *available_in;
      *available_in = 0;
    }
    // size_hint as update_size_hint would see it at the first encode_data (encode.rs:1604-1620, 2970):
    // everything received up to the end of the call in which the first input block fills up
    if ((!s->first_encode_seen || s->params.size_hint == 0) && (s->total_in >= BlockSize(s->params) || op != BROTLI_OPERATION_PROCESS)) {
      s->first_encode_seen = true;
      if (s->params.size_hint == 0) s->params.size_hint = (size_t)std::min<uint64_t>(s->total_in, (uint64_t)1 << 30);
    }
    if (op == BROTLI_OPERATION_FINISH) {
      if (!EncodeBuffered(s, true, false, 0, false, early_last)) return BROTLI_FALSE;
      s->stream_state = kFinished;
      std::vector<uint8_t>().swap(s->input);
    } else if (op == BROTLI_OPERATION_FLUSH) {
      // everything handed over so far becomes decodable output (encode.rs:2940-2975 with force_flush, then the
      // injected byte-alignment block :1541-1566); the encoder keeps the stream so far as the window of what follows
      // (repeated FLUSH calls that only drain output must not encode again)
      // (a partial piece leaves the output on a bit boundary: the flush padding is still owed then)
      if (!(s->carry.valid && s->encoded_upto == s->input.size() && s->carry.tail_nbits == 0) && !EncodeBuffered(s, false)) return BROTLI_FALSE;
    } else if (s->first_encode_seen) {
      // PROCESS: once a batch worth of input has piled up, the meta-blocks that are complete within it are encoded and
      // their output becomes available (BrotliEncoderHasMoreOutput); the bytes of the still open meta-block stay
      // buffered, together with the window the next piece needs -- memory stays bounded however long the stream is
      const size_t pending = s->input.size() - s->encoded_upto;
      if (pending >= StreamBatchBytes() && pending >= s->next_batch_try && !EncodeBuffered(s, false, false, 0, true)) return BROTLI_FALSE;
    }
  }
  // push output (inject_flush_or_push_output, encode.rs:1568-1598)
  if (AvailableOut(s) != 0 && *available_out != 0) {
    const size_t n = std::min(AvailableOut(s), *available_out);
    memcpy(*next_out, s->output.data() + s->output_pos, n);
    *next_out += n;
    *available_out -= n;
    s->output_pos += n;
    s->total_out += n;
  }
  if (AvailableOut(s) == 0) s->metadata_draining = false;
  if (total_out) *total_out = (size_t)s->total_out;
  return BROTLI_TRUE;
}

BROTLI_BOOL BrotliEncoderCompressStreaming(BrotliEncoderState* state, BrotliEncoderOperation op, size_t* available_in,
                                           const uint8_t* next_in, size_t* available_out, uint8_t* next_out) {
  const uint8_t* in = next_in;
  uint8_t* out = next_out;
  return BrotliEncoderCompressStream(state, op, available_in, &in, available_out, &out, nullptr);
}

BROTLI_BOOL BrotliEncoderIsFinished(BrotliEncoderState* s) {
  return (s && s->stream_state == kFinished && AvailableOut(s) == 0) ? BROTLI_TRUE : BROTLI_FALSE;
}
BROTLI_BOOL BrotliEncoderHasMoreOutput(BrotliEncoderState* s) { return (s && AvailableOut(s) != 0) ? BROTLI_TRUE : BROTLI_FALSE; }

const uint8_t* BrotliEncoderTakeOutput(BrotliEncoderState* s, size_t* size) {
  // take_output, encode.rs:3004-3030
  size_t consumed = AvailableOut(s);
  const uint8_t* result = s->output.data() + s->output_pos;
  if (*size != 0) consumed = std::min(*size, consumed);
  if (consumed != 0) {
    s->output_pos += consumed;
    s->total_out += consumed;
    *size = consumed;
    if (AvailableOut(s) == 0) s->metadata_draining = false;  // check_flush_complete, encode.rs:3018-3022
    return result;
  }
  *size = 0;
  return nullptr;
}

void BrotliEncoderSetCustomDictionary(BrotliEncoderState* s, size_t size, const uint8_t* dict) {
  if (!s) return;
  EnsureInitialized(s);
  s->params.use_dictionary = false;
  EncoderParams fin = s->params;
  FinalizeParams(&fin);
  if (size == 0 || size <= 1 || fin.quality == 0 || fin.quality == 1) {  // (qualities 0 / 1 take no dictionary, encode.rs:1237-1241)
    // too short: the reference only turns on catable + appendable (encode.rs:1237-1241)
    s->params.catable = true;
    s->params.appendable = true;
    s->params.use_dictionary = false;
    return;
  }
  s->dictionary.assign(dict, dict + size);
  s->has_dictionary = true;
  s->size_hint_at_dictionary = s->params.size_hint;  // ensure_initialized() fixes the hasher parameters here
}

uint8_t* BrotliEncoderMallocU8(BrotliEncoderState* s, size_t size) {
  return (uint8_t*)(s && s->alloc_func ? s->alloc_func(s->opaque, size) : malloc(size));
}
void BrotliEncoderFreeU8(BrotliEncoderState* s, uint8_t* data, size_t) {
  if (s && s->free_func) {
    s->free_func(s->opaque, data);
  } else {
    free(data);
  }
}
size_t* BrotliEncoderMallocUsize(BrotliEncoderState* s, size_t size) {
  return (size_t*)(s && s->alloc_func ? s->alloc_func(s->opaque, size * sizeof(size_t)) : malloc(size * sizeof(size_t)));
}
void BrotliEncoderFreeUsize(BrotliEncoderState* s, size_t* data, size_t) {
  if (s && s->free_func) {
    s->free_func(s->opaque, data);
  } else {
    free(data);
  }
}

// Inputs above this size take the one-shot call through the stream state machine in batches (bounded device memory, no 2 GiB
// limit): encoder_compress IS a loop over compress_stream with BROTLI_OPERATION_FINISH (encode.rs:1484-1520), and the stream
// does not depend on how the input is cut into calls as long as nothing is flushed -- input blocks fill up to 64 KiB whatever
// the writes, and the last block is known to be the last because the FINISH call itself hands over the final bytes.
static size_t OneShotStreamThreshold() {
  static const size_t v = getenv("BROTLI_MI355X_ONESHOT_STREAM_ABOVE") ? (size_t)strtoull(getenv("BROTLI_MI355X_ONESHOT_STREAM_ABOVE"), nullptr, 10) : ((size_t)1 << 30);
  return v;
}

// 1 = done, 0 = failed (message set), -1 = the stream does not fit the output buffer
static int CompressOneShotStreamed(int quality, int lgwin, BrotliEncoderMode mode, size_t input_size, const uint8_t* input, size_t* encoded_size,
                                   uint8_t* encoded) {
  BrotliEncoderState* s = BrotliEncoderCreateInstance(nullptr, nullptr, nullptr);
  if (!s) return 0;
  BrotliEncoderSetParameter(s, BROTLI_PARAM_QUALITY, (uint32_t)quality);
  BrotliEncoderSetParameter(s, BROTLI_PARAM_LGWIN, (uint32_t)lgwin);
  BrotliEncoderSetParameter(s, BROTLI_PARAM_MODE, (uint32_t)mode);
  BrotliEncoderSetParameter(s, BROTLI_PARAM_SIZE_HINT, (uint32_t)input_size);  // (a u32 in the reference as well, encode.rs:1474)
  if (lgwin > 24) BrotliEncoderSetParameter(s, BROTLI_PARAM_LARGE_WINDOW, 1);
  const size_t batch = StreamBatchBytes();
  size_t available_out = *encoded_size, done = 0;
  uint8_t* next_out = encoded;
  int result = 1;
  while (result == 1) {
    const bool last = input_size - done <= batch;
    size_t available_in = last ? input_size - done : batch;
    const uint8_t* next_in = input + done;
    const size_t fed = available_in;
    if (!BrotliEncoderCompressStream(s, last ? BROTLI_OPERATION_FINISH : BROTLI_OPERATION_PROCESS, &available_in, &next_in, &available_out, &next_out, nullptr)) {
      result = 0;
      break;
    }
    done += fed - available_in;
    if (last && available_in == 0) {
      // drain what the last call could not hand over
      while (BrotliEncoderHasMoreOutput(s) && available_out != 0) {
        size_t none = 0;
        const uint8_t* nothing = nullptr;
        if (!BrotliEncoderCompressStream(s, BROTLI_OPERATION_FINISH, &none, &nothing, &available_out, &next_out, nullptr)) {
          result = 0;
          break;
        }
      }
      if (result == 1 && !BrotliEncoderIsFinished(s)) result = -1;
      break;
    }
    if (BrotliEncoderHasMoreOutput(s) && available_out == 0) result = -1;
  }
  if (result == 1) *encoded_size = (size_t)(next_out - encoded);
  BrotliEncoderDestroyInstance(s);
  return result;
}

static BROTLI_BOOL CompressOneShot(int quality, int lgwin, BrotliEncoderMode mode, size_t input_size, const uint8_t* input,
                                   bool input_on_device, size_t* encoded_size, uint8_t* encoded, double* stats_out) {
  // encoder_compress, encode.rs:1436-1538
  const size_t out_size = *encoded_size;
  const size_t max_out_size = MaxCompressedSize(input_size);
  if (out_size == 0) return BROTLI_FALSE;
  if (input_size == 0) {
    *encoded_size = 1;
    encoded[0] = 6;
    return BROTLI_TRUE;
  }
  // encode.rs:1468-1481: the one-shot entry runs quality 10 ("9.5") at quality 9, with an H9 hasher made ahead of time from
  // {q9_5, quality 10} -- the hasher quality 9 selects anyway.  (Quality 10 / 11 through the stream API are Zopfli.)
  if (quality == 10) quality = 9;
  // Qualities 2 .. 4 and 11 (the BasicHasher table / the H10 trees travel through the stream as they are) do not reproduce the
  // hasher reset of the reference's 3 GiB position wrap (encode.rs:1623-1631, 1705-1710): a one-shot input that would reach it is
  // refused here, before any work is done, rather than after gigabytes (include/brotli_mi355x.h, "Limits").
  if (((quality >= 2 && quality <= 4) || quality == 11) && (uint64_t)input_size > FirstPositionWrap()) {
    SetError("BrotliEncoderCompress", "qualities 2..4 and 10/11: inputs that cross the reference's position wrap at 3 GiB are not supported (split the input, e.g. BrotliEncoderCompressMulti)");
    *encoded_size = 0;
    return BROTLI_FALSE;
  }
  // (qualities 2 .. 4 walk their blocks one launch at a time with the host resolver replaying the stream so far in between: the
  // pieces of the stream machine keep that replay short -- 16 KiB blocks, quadratic otherwise)
  const size_t stream_above = (quality >= 2 && quality <= 4) ? std::min(OneShotStreamThreshold(), (size_t)64 << 20) : OneShotStreamThreshold();
  if (!input_on_device && quality > 1 && input_size > stream_above) {
    size_t n = out_size;
    const int r = CompressOneShotStreamed(quality, lgwin, mode, input_size, input, &n, encoded);
    if (r == 1 && !(max_out_size != 0 && n > max_out_size)) {
      *encoded_size = n;
      return BROTLI_TRUE;
    }
    *encoded_size = 0;
    if (r == 0 || max_out_size == 0) return BROTLI_FALSE;
    if (out_size >= max_out_size) {  // (compression expanded the input, or the caller's buffer is too small for it: as below)
      *encoded_size = MakeUncompressedStream(input, input_size, encoded);
      return BROTLI_TRUE;
    }
    return BROTLI_FALSE;
  }
  bool ok = false;
  std::vector<uint8_t> out;
  try {
    EncodeRequest req;
    SetParameter(&req.params, kParamQuality, (uint32_t)quality);
    SetParameter(&req.params, kParamLgwin, (uint32_t)lgwin);
    SetParameter(&req.params, kParamMode, (uint32_t)mode);
    SetParameter(&req.params, kParamSizeHint, (uint32_t)input_size);
    if (lgwin > 24) SetParameter(&req.params, kParamLargeWindow, 1);
    req.input = input;
    req.input_size = input_size;
    req.input_on_device = input_on_device;
    EncodeStats st;
    SmallCallGate gate(input_size);
    if (IsFragmentStream(req.params)) {
      // qualities 0 and 1: compress_stream(FINISH) takes the fragment path (encode.rs:2929-2937)
      if (input_on_device) throw std::runtime_error("qualities 0 and 1 take their input from host memory");
      FragmentStream fs;
      FragmentStreamCompress(req.params, &fs, input, input_size, true, false, &out);
    } else {
      EncodeStream(req, &out, &st);
    }
    if (stats_out) {
      stats_out[0] = st.lz77_rounds;
      stats_out[1] = (double)st.searches;
      stats_out[2] = (double)st.commands;
      stats_out[3] = (double)st.literals;
      stats_out[4] = st.metablocks;
      stats_out[5] = st.uncompressed_metablocks;
      stats_out[6] = st.fallback_retries;
      stats_out[7] = st.ms_lz77;
      stats_out[8] = st.ms_metablock;
      stats_out[9] = st.ms_total;
      for (int i = 0; i < 16; ++i) stats_out[10 + i] = st.ms_phase[i];
      stats_out[26] = st.parse_kernel_ms;
      stats_out[27] = st.parse_launches;
      stats_out[28] = (double)st.parse_segments;
      stats_out[29] = st.num_segments;
      stats_out[30] = st.segment_bytes;
    }
    ok = out.size() <= out_size;
  } catch (const std::exception& e) {
    SetError("BrotliEncoderCompress", e.what());
    *encoded_size = 0;
    return BROTLI_FALSE;  // no silent fallback for unsupported parameters / missing device
  }
  if (ok && !(max_out_size != 0 && out.size() > max_out_size)) {
    memcpy(encoded, out.data(), out.size());
    *encoded_size = out.size();
    return BROTLI_TRUE;
  }
  *encoded_size = 0;
  if (max_out_size == 0) return BROTLI_FALSE;
  if (out_size >= max_out_size && !input_on_device) {
    *encoded_size = MakeUncompressedStream(input, input_size, encoded);
    return BROTLI_TRUE;
  }
  return BROTLI_FALSE;
}

BROTLI_BOOL BrotliEncoderCompress(int quality, int lgwin, BrotliEncoderMode mode, size_t input_size, const uint8_t* input_buffer,
                                  size_t* encoded_size, uint8_t* encoded_buffer) {
  return CompressOneShot(quality, lgwin, mode, input_size, input_buffer, false, encoded_size, encoded_buffer, nullptr);
}

BROTLI_BOOL BrotliMi355xCompressDevice(int quality, int lgwin, BrotliEncoderMode mode, size_t input_size,
                                       const uint8_t* input_device, size_t* encoded_size, uint8_t* encoded_host, double* stats) {
  return CompressOneShot(quality, lgwin, mode, input_size, input_device, true, encoded_size, encoded_host, stats);
}

int32_t BrotliEncoderCompressMulti(size_t num_params, const BrotliEncoderParameter* param_keys, const uint32_t* param_values,
                                   size_t input_size, const uint8_t* input_buffer, size_t* encoded_size, uint8_t* encoded,
                                   size_t desired_num_threads, brotli_alloc_func, brotli_free_func, void**) {
  return CompressMultiImpl(num_params, param_keys, param_values, input_size, input_buffer, encoded_size, encoded, desired_num_threads);
}

BrotliEncoderWorkPool* BrotliEncoderCreateWorkPool(size_t, brotli_alloc_func alloc_func, brotli_free_func free_func, void** opaque) {
  if ((alloc_func == nullptr) != (free_func == nullptr)) return nullptr;
  void* o = opaque ? *opaque : nullptr;
  void* mem = alloc_func ? alloc_func(o, sizeof(BrotliEncoderWorkPoolStruct)) : malloc(sizeof(BrotliEncoderWorkPoolStruct));
  if (!mem) return nullptr;
  BrotliEncoderWorkPool* w = (BrotliEncoderWorkPool*)mem;
  w->alloc_func = alloc_func;
  w->free_func = free_func;
  w->opaque = o;
  return w;
}
void BrotliEncoderDestroyWorkPool(BrotliEncoderWorkPool* w) {
  if (!w) return;
  if (w->free_func) {
    w->free_func(w->opaque, w);
  } else {
    free(w);
  }
}
int32_t BrotliEncoderCompressWorkPool(BrotliEncoderWorkPool*, size_t num_params, const BrotliEncoderParameter* param_keys,
                                      const uint32_t* param_values, size_t input_size, const uint8_t* input_buffer,
                                      size_t* encoded_size, uint8_t* encoded, size_t desired_num_threads, brotli_alloc_func,
                                      brotli_free_func, void**) {
  return CompressMultiImpl(num_params, param_keys, param_values, input_size, input_buffer, encoded_size, encoded, desired_num_threads);
}

int32_t BrotliMi355xCompressChunk(size_t num_params, const BrotliEncoderParameter* param_keys, const uint32_t* param_values,
                                  size_t input_size, const uint8_t* input_buffer, int input_on_device, size_t thread_index,
                                  size_t num_threads, size_t* encoded_size, uint8_t* encoded) {
  try {
    EncoderParams params;
    if (!ParamsFromLists(num_params, param_keys, param_values, &params)) return 0;
    if (num_threads == 0 || thread_index >= num_threads) return 0;
    std::vector<uint8_t> out;
    CompressChunk(params, input_buffer, input_size, input_on_device != 0, thread_index, num_threads, &out, nullptr);
    if (out.size() > *encoded_size) {
      SetError("BrotliMi355xCompressChunk", "insufficient output space");
      return 0;
    }
    memcpy(encoded, out.data(), out.size());
    *encoded_size = out.size();
    return 1;
  } catch (const std::exception& e) {
    SetError("BrotliMi355xCompressChunk", e.what());
    return 0;
  }
}

int32_t BrotliMi355xConcatChunks(size_t num_chunks, const uint8_t* const* chunks, const size_t* chunk_sizes, size_t* encoded_size,
                                 uint8_t* encoded) {
  // stitched straight into the caller's buffer: every chunk byte is copied once
  ByteSink out(encoded, *encoded_size);
  ChunkStitcher stitcher;
  for (size_t i = 0; i < num_chunks; ++i) {
    if (!stitcher.Append(chunks[i], chunk_sizes[i], &out)) {
      SetError("BrotliMi355xConcatChunks", "chunk cannot be concatenated");
      return 0;
    }
  }
  stitcher.Finish(&out);
  if (out.overflow()) {
    SetError("BrotliMi355xConcatChunks", "insufficient output space");
    return 0;
  }
  *encoded_size = out.size();
  return 1;
}

int32_t BrotliMi355xConcatChunkEnds(size_t num_chunks, const uint8_t* heads, const uint8_t* tails, const size_t* chunk_sizes,
                                    size_t* encoded_size, uint8_t* encoded, size_t* body_copies) {
  ByteSink out(encoded, *encoded_size);
  ChunkStitcher stitcher;
  for (size_t i = 0; i < num_chunks; ++i) {
    ChunkView view;
    view.size = chunk_sizes[i];
    view.head = heads + 8 * i;
    view.tail = tails + 8 * i;
    view.head_len = view.tail_len = view.size < 8 ? view.size : 8;
    BodyCopy body;
    if (!stitcher.Append(view, &out, &body)) {
      SetError("BrotliMi355xConcatChunkEnds", "chunk cannot be concatenated");
      return 0;
    }
    body_copies[3 * i] = body.dst_offset;
    body_copies[3 * i + 1] = body.src_offset;
    body_copies[3 * i + 2] = body.size;
  }
  stitcher.Finish(&out);
  if (out.overflow()) {
    SetError("BrotliMi355xConcatChunkEnds", "insufficient output space");
    return 0;
  }
  *encoded_size = out.size();
  return 1;
}

const char* BrotliMi355xDeviceName(void) {
  try {
    return dev_name();
  } catch (...) {
    return "unavailable";
  }
}
const char* BrotliMi355xLastError(void) { return g_last_error.c_str(); }

}  // extern "C"
