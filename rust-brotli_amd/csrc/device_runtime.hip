// device_runtime.hip -- device memory helpers and the read-only tables of the encoder hot path.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <stdexcept>
#include <string>

#include "device_api.h"
#include "../../tables/brotli_tables.h"
#include "../../tables/brotli_static_dict_lut.h"

// Hardware queues: every host thread of the library works on its own stream, and a live chain (lz77_live.h) is one kernel
// that runs for seconds; with the runtime's default of four hardware queues the streams of eight shard workers share
// queues and the short kernels of one shard wait behind the long kernel of another.  The library does NOT touch the
// process environment: the host application asks for more queues itself (GPU_MAX_HW_QUEUES=16 before the HIP runtime
// starts) -- the Python binding and bench.py do.

namespace brotli_mi355x {

void hip_check(hipError_t e, const char* what);
#define HIP_CHECK(x) hip_check((x), #x)

// Device memory pool: a compression call needs a few dozen large scratch buffers whose sizes repeat from call
// to call.  hipMalloc / hipFree cost milliseconds each (and hipFree synchronises the device), so freed blocks
// are kept per size class and handed out again.  Blocks are zero-filled on allocation (the encoder relies on
// it, like the reference relies on zeroed hash tables, encode.rs:1147).
namespace {
struct Pool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;  // capacity -> block
  std::unordered_map<void*, size_t> capacity;  // every live block handed out or pooled
  std::unordered_map<void*, uint64_t> pooled_seq;  // pooled blocks: when they came back (the oldest go first when the pool is over its cap)
  uint64_t seq = 0;
  size_t pooled_bytes = 0;
  int device = -1;  // the device the owning thread allocates on (a host thread of the library stays on one device)
  // (a 1 GiB piece alone needs ~95 GiB of the 288; with 192 GiB kept by the calling thread, the eight helper threads of a
  // multi-shard call that followed found 7 GB free and trimmed in the middle of the call)
  static constexpr size_t kMaxPooled = (size_t)128 << 30;
  Pool();
  ~Pool();
  // hands every pooled block back to the driver; returns the bytes freed (any thread may call it: hipFree waits for the
  // device, so nothing queued on the owner's stream can still be using a block that sits in its free list)
  size_t Trim() {
    std::vector<void*> drop;
    size_t bytes;
    {
      std::lock_guard<std::mutex> lock(mu);
      for (auto& kv : free_blocks) {
        drop.push_back(kv.second);
        capacity.erase(kv.second);
      }
      free_blocks.clear();
      pooled_seq.clear();
      bytes = pooled_bytes;
      pooled_bytes = 0;
    }
    for (void* d : drop) (void)hipFree(d);
    return bytes;
  }
};
// Every host thread has a pool of its own, but the memory they sit on is one device's: when an allocation fails, the
// pools of the OTHER threads are emptied as well (the helper threads of BrotliEncoderCompressMulti start with nothing while
// the calling thread may be sitting on 190 GiB from earlier one-shot calls).
std::mutex g_pools_mu;
std::vector<Pool*>& all_pools() {
  static std::vector<Pool*>* v = new std::vector<Pool*>;  // (never destroyed: threads may outlive static destruction)
  return *v;
}
Pool::Pool() {
  std::lock_guard<std::mutex> lock(g_pools_mu);
  all_pools().push_back(this);
}
// What a host thread leaves behind when it ends.  A pool is thread-local, and its destructor runs while the thread is being torn
// down -- next to the HIP runtime's own thread-local state (hipStreamPerThread), in no defined order: calling into the runtime from
// there (hipFree, hipHostFree, hipEventDestroy) corrupted the heap now and then (64 Python threads of six calls each:
// "malloc_consolidate(): unaligned fastbin chunk detected" at exit in two runs of three).  So a dying thread calls nothing: its idle
// blocks go to this list, from where the next thread that needs one of that size adopts it, and TrimAllPools() hands the rest back
// to the driver from a live thread.  (Never destroyed, like the list of pools: threads may outlive static destruction.)
struct Orphans {
  std::mutex mu;
  std::map<int, std::multimap<size_t, void*>> device_blocks;  // device -> (capacity -> block): a block is only adopted on its own device
  std::multimap<size_t, void*> host_blocks;                  // capacity -> block
  size_t device_bytes = 0;                                    // what device_blocks holds
};
Orphans& orphans() {
  static Orphans* o = new Orphans;
  return *o;
}
Pool::~Pool() {
  {
    std::lock_guard<std::mutex> lock(g_pools_mu);
    auto& v = all_pools();
    v.erase(std::remove(v.begin(), v.end(), this), v.end());
  }
  Orphans& o = orphans();
  std::lock_guard<std::mutex> lock(o.mu);
  for (auto& kv : free_blocks) {
    o.device_blocks[device].emplace(kv.first, kv.second);
    o.device_bytes += kv.first;
  }
}
// (called from a live thread: device blocks AND the page-locked host blocks that ended threads left behind go back to the driver;
// hipFree / hipHostFree wait for the device, so nothing queued by the dead thread can still be using them)
static size_t TrimOrphans() {
  std::vector<void*> drop, drop_host;
  size_t bytes = 0;
  {
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lock(o.mu);
    for (auto& dv : o.device_blocks)
      for (auto& kv : dv.second) {
        drop.push_back(kv.second);
        bytes += kv.first;
      }
    o.device_blocks.clear();
    o.device_bytes = 0;
    for (auto& kv : o.host_blocks) drop_host.push_back(kv.second);
    o.host_blocks.clear();
  }
  for (void* d : drop) (void)hipFree(d);
  for (void* h : drop_host) (void)hipHostFree(h);
  return bytes;
}
size_t TrimAllPools() {
  size_t bytes = TrimOrphans();
  std::lock_guard<std::mutex> lock(g_pools_mu);
  for (Pool* q : all_pools()) bytes += q->Trim();
  return bytes;
}
// one pool per host thread = per stream: a block is only ever reused by work queued behind its previous use
Pool& pool() {
  static thread_local Pool p;
  return p;
}
size_t RoundUp(size_t bytes) {
  size_t g = 1 << 16;
  while (g < bytes / 8) g <<= 1;  // granularity ~ 1/8 of the size: at most 12.5 % slack
  return (bytes + g - 1) / g * g;
}
}  // namespace

// what the driver was asked for (BROTLI_MI355X_TIMELINE): [0] hipMalloc calls, [1] ms in them, [2] hipFree calls, [3] ms
static thread_local double g_pool_counters[4] = {0, 0, 0, 0};
void dev_pool_counters(double* out, bool reset) {
  for (int i = 0; i < 4; ++i) {
    out[i] = g_pool_counters[i];
    if (reset) g_pool_counters[i] = 0;
  }
}
namespace {
struct DriverCallClock {
  int slot;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit DriverCallClock(int s) : slot(s) {}
  ~DriverCallClock() {
    g_pool_counters[slot] += 1;
    g_pool_counters[slot + 1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
};
}  // namespace

// ---- zero fills, noted and written in batches (BR_STREAM, device_api.h)
namespace {
constexpr uint32_t kZeroBatch = 32;               // ranges per launch
constexpr size_t kZeroDeferMax = (size_t)4 << 20; // longer fills go to the runtime's own fill kernel at once
constexpr size_t kZeroChunk = (size_t)64 << 10;   // bytes per workgroup
struct ZeroRanges {
  unsigned long long ptr[kZeroBatch];
  unsigned long long bytes[kZeroBatch];
};
__global__ __launch_bounds__(256) void k_zero_ranges(ZeroRanges z) {
  const uint32_t r = blockIdx.y;
  const size_t n = (size_t)z.bytes[r];
  const size_t lo = (size_t)blockIdx.x * kZeroChunk;
  if (lo >= n) return;
  const size_t hi = lo + kZeroChunk < n ? lo + kZeroChunk : n;
  uint8_t* q = (uint8_t*)z.ptr[r] + lo;
  uint8_t* e = (uint8_t*)z.ptr[r] + hi;
  uint8_t* qa = (uint8_t*)(((uintptr_t)q + 15) & ~(uintptr_t)15);
  if (qa > e) qa = e;
  uint8_t* ea = (uint8_t*)((uintptr_t)e & ~(uintptr_t)15);
  if (ea < qa) ea = qa;
  for (uint8_t* x = q + threadIdx.x; x < qa; x += 256) *x = 0;
  for (uint4* x = (uint4*)qa + threadIdx.x; x < (uint4*)ea; x += 256) *x = make_uint4(0u, 0u, 0u, 0u);
  for (uint8_t* x = ea + threadIdx.x; x < e; x += 256) *x = 0;
}
struct PendingZero {
  ZeroRanges z;
  uint32_t n = 0;
  size_t longest = 0;
};
PendingZero& pending_zero() {
  static thread_local PendingZero p;
  return p;
}
void flush_zero() {
  PendingZero& pz = pending_zero();
  if (pz.n == 0) return;
  const uint32_t n = pz.n;
  const size_t longest = pz.longest;
  pz.n = 0;
  pz.longest = 0;
  if (n == 1) {
    HIP_CHECK(hipMemsetAsync((void*)pz.z.ptr[0], 0, (size_t)pz.z.bytes[0], hipStreamPerThread));
    return;
  }
  hipLaunchKernelGGL(k_zero_ranges, dim3((uint32_t)((longest + kZeroChunk - 1) / kZeroChunk), n), dim3(256), 0, hipStreamPerThread, pz.z);
  HIP_CHECK(hipGetLastError());
}
void note_zero(void* p, size_t bytes) {
  if (bytes == 0) return;
  static const bool direct = getenv("BROTLI_MI355X_DIRECT_FILLS") != nullptr;  // (A/B: one runtime fill per request, as before)
  static const bool log_big = getenv("BROTLI_MI355X_DEBUG_FILLS") != nullptr;  // (which zero fills of a call are megabytes)
  if (log_big && bytes >= ((size_t)1 << 20)) fprintf(stderr, "zero fill of %.1f MiB\n", bytes / 1048576.0);
  if (bytes > kZeroDeferMax || direct) {
    flush_zero();
    HIP_CHECK(hipMemsetAsync(p, 0, bytes, hipStreamPerThread));
    return;
  }
  PendingZero& pz = pending_zero();
  if (pz.n == kZeroBatch) flush_zero();
  pz.z.ptr[pz.n] = (unsigned long long)(uintptr_t)p;
  pz.z.bytes[pz.n] = bytes;
  pz.n++;
  if (bytes > pz.longest) pz.longest = bytes;
}
}  // namespace
ihipStream_t* dev_stream_flushed() {
  flush_zero();
  return hipStreamPerThread;
}

static void* AllocBlock(size_t bytes);
void* dev_alloc(size_t bytes) {
  if (bytes == 0) bytes = 16;
  void* p = AllocBlock(bytes);
  note_zero(p, bytes);
  return p;
}
// for the big work arrays that are written in full before anything reads them (zero-filling them costs a pass over
// gigabytes per call); BROTLI_MI355X_POISON=1 fills them with a pattern instead, to flush out hidden dependencies
void* dev_alloc_uninit(size_t bytes) {
  if (bytes == 0) bytes = 16;
  void* p = AllocBlock(bytes);
  static const bool poison = getenv("BROTLI_MI355X_POISON") != nullptr;
  if (poison) HIP_CHECK(hipMemsetAsync(p, 0xA5, bytes, BR_STREAM));
  return p;
}
static void* AllocBlock(size_t bytes) {
  const size_t cap = RoundUp(bytes);
  Pool& P = pool();
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.free_blocks.lower_bound(cap);
    if (it != P.free_blocks.end() && it->first <= cap + cap / 4) {
      p = it->second;
      P.pooled_bytes -= it->first;
      P.pooled_seq.erase(p);
      P.free_blocks.erase(it);
    }
  }
  if (P.device < 0) HIP_CHECK(hipGetDevice(&P.device));
  bool adopted = false;
  if (!p) {
    // a block that a finished thread left behind on this device?
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lock(o.mu);
    auto dv = o.device_blocks.find(P.device);
    if (dv != o.device_blocks.end()) {
      auto it = dv->second.lower_bound(cap);
      if (it != dv->second.end() && it->first <= cap + cap / 4) {
        p = it->second;
        const size_t got = it->first;
        dv->second.erase(it);
        o.device_bytes -= got;
        std::lock_guard<std::mutex> mine(P.mu);
        P.capacity[p] = got;
        adopted = true;
      }
    }
  }
  if (adopted) {
    // the block was last used on the DEAD thread's stream, and the rule of the pool -- a block is only reused by work queued behind
    // its previous use -- does not hold across streams: wait once for the device (adoptions are rare: a thread ended with idle blocks)
    HIP_CHECK(hipDeviceSynchronize());
  }
  if (!p) {
    DriverCallClock clock(0);
    // what ended threads left behind does not count against any pool's cap: before asking the driver for more, a live thread hands
    // back orphaned blocks beyond 32 GiB (nothing of the right size was among them, or it would have been adopted above)
    {
      bool trim;
      {
        Orphans& o = orphans();
        std::lock_guard<std::mutex> lock(o.mu);
        trim = o.device_bytes > ((size_t)32 << 30);
      }
      if (trim) TrimOrphans();
    }
    // Leave the runtime room of its own: kernel scratch, the queues of helper threads, signals.  With every last byte in
    // the pools, a later launch died inside the runtime (HSA_STATUS_ERROR_OUT_OF_RESOURCES, "Available Free mem : 0 MB")
    // where nothing can be caught.  Idle pooled blocks go back first -- this thread's, then every thread's.
    {
      // (1 GiB is plenty for the runtime, and trimming is expensive: hipFree waits for the whole device and handing back
      // ~150 GiB of pooled blocks takes seconds)
      static constexpr size_t kHeadroom = (size_t)1 << 30;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < cap + kHeadroom) {
        P.Trim();
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < cap + kHeadroom) TrimAllPools();
      }
    }
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
      // give pooled memory back to the driver and retry: this thread's pool first, then every thread's
      (void)hipGetLastError();
      P.Trim();
      e = hipMalloc(&p, cap);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        TrimAllPools();
        HIP_CHECK(hipMalloc(&p, cap));
      }
    }
    std::lock_guard<std::mutex> lock(P.mu);
    P.capacity[p] = cap;
  }
  return p;
}
void dev_free(void* p) {
  if (!p) return;
  flush_zero();  // (a noted fill of this block must not outlive it: the pool may hand the block back to the driver)
  Pool& P = pool();
  std::lock_guard<std::mutex> lock(P.mu);
  auto it = P.capacity.find(p);
  if (it == P.capacity.end()) {
    DriverCallClock clock(2);
    (void)hipFree(p);
    return;
  }
  if (it->second > Pool::kMaxPooled) {
    DriverCallClock clock(2);
    P.capacity.erase(it);
    (void)hipFree(p);
    return;
  }
  // over the cap: the blocks that have sat in the pool longest go (the sizes of the call before this one), not the block
  // coming back now -- the next call of the same kind wants exactly that one
  while (P.pooled_bytes + it->second > Pool::kMaxPooled && !P.free_blocks.empty()) {
    auto oldest = P.free_blocks.begin();
    for (auto b = P.free_blocks.begin(); b != P.free_blocks.end(); ++b)
      if (P.pooled_seq[b->second] < P.pooled_seq[oldest->second]) oldest = b;
    DriverCallClock clock(2);
    P.pooled_bytes -= oldest->first;
    P.capacity.erase(oldest->second);
    P.pooled_seq.erase(oldest->second);
    (void)hipFree(oldest->second);
    P.free_blocks.erase(oldest);
  }
  P.free_blocks.emplace(it->second, p);
  P.pooled_seq[p] = ++P.seq;
  P.pooled_bytes += it->second;
}
// page-locked host blocks are pooled like the device blocks (hipHostMalloc costs a fraction of a millisecond)
namespace {
struct HostPool {
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, size_t> capacity;
  ~HostPool() {  // (no call into the runtime from a thread that is being torn down: see Orphans)
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lock(o.mu);
    for (auto& kv : free_blocks) o.host_blocks.emplace(kv.first, kv.second);
  }
};
HostPool& host_pool() {
  static thread_local HostPool p;
  return p;
}
}  // namespace
void* dev_host_alloc(size_t bytes) {
  const size_t cap = RoundUp(bytes ? bytes : 16);
  HostPool& P = host_pool();
  auto it = P.free_blocks.lower_bound(cap);
  if (it != P.free_blocks.end() && it->first <= cap + cap / 4) {
    void* p = it->second;
    P.free_blocks.erase(it);
    return p;
  }
  void* p = nullptr;
  {
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lock(o.mu);
    auto oit = o.host_blocks.lower_bound(cap);
    if (oit != o.host_blocks.end() && oit->first <= cap + cap / 4) {
      p = oit->second;
      P.capacity[p] = oit->first;
      o.host_blocks.erase(oit);
      return p;
    }
  }
  HIP_CHECK(hipHostMalloc(&p, cap, hipHostMallocDefault));
  P.capacity[p] = cap;
  return p;
}
void dev_host_free(void* p) {
  if (!p) return;
  HostPool& P = host_pool();
  auto it = P.capacity.find(p);
  if (it == P.capacity.end()) {
    (void)hipHostFree(p);
    return;
  }
  P.free_blocks.emplace(it->second, p);
}
static void wait_stream();
void dev_memset(void* p, int value, size_t bytes) {
  if (bytes == 0) return;
  if (value == 0) {
    note_zero(p, bytes);
    return;
  }
  HIP_CHECK(hipMemsetAsync(p, value, bytes, BR_STREAM));
}
void dev_h2d(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, BR_STREAM));
}
void dev_d2h_async(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, BR_STREAM));
}
void dev_d2h(void* dst, const void* src, size_t bytes) {
  if (bytes) {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, BR_STREAM));
    wait_stream();
  }
}
namespace {
constexpr size_t kBounceBytes = (size_t)4 << 20;
struct Bounce {
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool busy[2] = {false, false};
  void init() {
    if (buf[0]) return;
    for (int i = 0; i < 2; ++i) {
      {
        // (the buffers of a thread that has ended are taken over: see Orphans)
        Orphans& o = orphans();
        std::lock_guard<std::mutex> lock(o.mu);
        auto it = o.host_blocks.find(kBounceBytes);
        if (it != o.host_blocks.end()) {
          buf[i] = it->second;
          o.host_blocks.erase(it);
        }
      }
      if (!buf[i]) HIP_CHECK(hipHostMalloc(&buf[i], kBounceBytes, hipHostMallocDefault));
      HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
  }
  ~Bounce() {
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lock(o.mu);
    for (int i = 0; i < 2; ++i)
      if (buf[i]) o.host_blocks.emplace(kBounceBytes, buf[i]);
  }
  void wait(int i) {
    if (!busy[i]) return;
    HIP_CHECK(hipEventSynchronize(ev[i]));
    busy[i] = false;
  }
};
Bounce& bounce() {
  static thread_local Bounce b;
  return b;
}
bool is_page_locked(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (ordinary memory: not an error for us)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
}  // namespace
// BROTLI_MI355X_BULK: bit 0 = bounce uploads, bit 1 = bounce downloads.  Measured on 64 MiB in / 16 MB out per call
// (tools/host_path.py, bench.py's c_abi_pageable): with neither 48 ms per call; with both 30 ms.  Bouncing only one
// direction gave 28-29 ms in one harness and 48 ms in another -- the runtime's own path for pageable memory is the
// part that is not dependable, so both directions are bounced unless the caller's buffer is page-locked.
static int bulk_mode() {
  static const int m = getenv("BROTLI_MI355X_BULK") ? atoi(getenv("BROTLI_MI355X_BULK")) : 3;
  return m;
}
void dev_h2d_bulk(void* dst, const void* src, size_t bytes) {
  if (!(bulk_mode() & 1) || bytes < kBounceBytes / 4 || is_page_locked(src)) {
    dev_h2d(dst, src, bytes);
    return;
  }
  Bounce& b = bounce();
  b.init();
  int i = 0;
  for (size_t off = 0; off < bytes; off += kBounceBytes, i ^= 1) {
    const size_t len = bytes - off < kBounceBytes ? bytes - off : kBounceBytes;
    b.wait(i);
    memcpy(b.buf[i], (const uint8_t*)src + off, len);
    HIP_CHECK(hipMemcpyAsync((uint8_t*)dst + off, b.buf[i], len, hipMemcpyHostToDevice, BR_STREAM));
    HIP_CHECK(hipEventRecord(b.ev[i], BR_STREAM));
    b.busy[i] = true;
  }
}
void dev_d2h_bulk(void* dst, const void* src, size_t bytes) {
  if (!(bulk_mode() & 2) || bytes < kBounceBytes / 4 || is_page_locked(dst)) {
    dev_d2h(dst, src, bytes);
    return;
  }
  Bounce& b = bounce();
  b.init();
  b.wait(0);
  b.wait(1);
  // chunk c is on the bus while chunk c - 1 is copied out of its buffer
  size_t prev_off = 0, prev_len = 0;
  int i = 0;
  for (size_t off = 0; off < bytes; off += kBounceBytes, i ^= 1) {
    const size_t len = bytes - off < kBounceBytes ? bytes - off : kBounceBytes;
    HIP_CHECK(hipMemcpyAsync(b.buf[i], (const uint8_t*)src + off, len, hipMemcpyDeviceToHost, BR_STREAM));
    HIP_CHECK(hipEventRecord(b.ev[i], BR_STREAM));
    b.busy[i] = true;
    if (prev_len) {
      b.wait(i ^ 1);
      memcpy((uint8_t*)dst + prev_off, b.buf[i ^ 1], prev_len);
    }
    prev_off = off;
    prev_len = len;
  }
  if (prev_len) {
    b.wait(i ^ 1);
    memcpy((uint8_t*)dst + prev_off, b.buf[i ^ 1], prev_len);
  }
}
void dev_d2d(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, BR_STREAM));
}
// BROTLI_MI355X_POLLING_WAIT=1: poll (hipStreamQuery / hipEventQuery) instead of sleeping in the runtime's wait.  Off by
// default: a caller that polls in a tight loop keeps the runtime from feeding large copies (1 GiB host to host went from
// 2.1 to 4.9 s), and the short waits between LZ77 rounds gained nothing measurable from it.
static bool polling_waits() {
  static const bool on = getenv("BROTLI_MI355X_POLLING_WAIT") != nullptr;
  return on;
}
static void wait_stream() {
  if (!polling_waits()) {
    HIP_CHECK(hipStreamSynchronize(BR_STREAM));
    return;
  }
  for (;;) {
    const hipError_t e = hipStreamQuery(BR_STREAM);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    __builtin_ia32_pause();
  }
}
void dev_sync() { wait_stream(); }

namespace {
struct Mark {
  hipEvent_t ev = nullptr;  // (left to the runtime when the thread ends: no call into it from a dying thread, see Orphans)
};
Mark& mark(int i = 4) {
  static thread_local Mark m[5];
  return m[i];
}
void record_mark(Mark& m) {
  if (!m.ev) HIP_CHECK(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(m.ev, BR_STREAM));
}
void wait_mark(Mark& m) {
  if (!m.ev) return;
  if (!polling_waits()) {
    HIP_CHECK(hipEventSynchronize(m.ev));
    return;
  }
  for (;;) {
    const hipError_t e = hipEventQuery(m.ev);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    __builtin_ia32_pause();
  }
}
}  // namespace
void dev_mark_n(int i) { record_mark(mark(i & 3)); }
void dev_wait_mark_n(int i) { wait_mark(mark(i & 3)); }
void dev_mark() {
  Mark& m = mark();
  if (!m.ev) HIP_CHECK(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(m.ev, BR_STREAM));
}
void dev_wait_mark() {
  Mark& m = mark();
  if (!m.ev) return;
  if (!polling_waits()) {
    HIP_CHECK(hipEventSynchronize(m.ev));
    return;
  }
  for (;;) {
    const hipError_t e = hipEventQuery(m.ev);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    __builtin_ia32_pause();
  }
}

// Before a call that spreads its work over helper threads: if less than `min_free_share` percent of the device memory is
// free, every thread's idle pooled blocks go back to the driver NOW, while nothing runs.  hipFree waits for the whole
// device: the same trim forced by allocations in the middle of the call -- kernels of seconds when the shards are live
// chains -- stalls every worker (8 H5 shards: 42 s instead of 14 with 7 GB free after the 1 GiB cases).  It is not free
// either -- handing back ~150 GiB takes seconds -- hence only when memory is really short.
size_t dev_trim_pool();
void dev_make_room(unsigned min_free_share) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return;
  if (free_b * 100 >= total_b * (size_t)min_free_share) return;
  // The caller is about to hand its work to helper threads and wait: what sits idle in ITS pool goes first (round 6: a process that had
  // run 1 GiB one-shot calls -- 128 GiB pooled by the calling thread -- and then multi-shard calls trimmed EVERY pool at the start of
  // every such call, the helpers' blocks of the call before included: 5 s of hipMalloc per shard, 9 s for a 1.4 s call).
  dev_trim_pool();
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b * 100 >= total_b * (size_t)min_free_share) return;
  TrimAllPools();
}

void dev_make_room_for(size_t bytes, size_t own_share) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return;
  // twice the estimate: the shards of a call differ in what they need, and a pool only hands out blocks of about the size asked for
  if (free_b >= std::min(2 * bytes, total_b - total_b / 8)) return;
  // (what the helpers keep pooled from the call before is not "free" but is theirs to reuse: only the caller's left-overs are at stake,
  // and only when they are clearly more than its own share of this call)
  size_t mine;
  {
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    mine = P.pooled_bytes;
  }
  if (mine > 2 * own_share) dev_trim_pool();
}

// gives the pooled (currently unused) device memory of the calling thread back to the driver
size_t dev_trim_pool() {
  Pool& P = pool();
  std::vector<void*> drop;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lock(P.mu);
    for (auto& kv : P.free_blocks) {
      drop.push_back(kv.second);
      P.capacity.erase(kv.second);
      bytes += kv.first;
    }
    P.free_blocks.clear();
    P.pooled_seq.clear();
    P.pooled_bytes = 0;
  }
  if (!drop.empty()) HIP_CHECK(hipStreamSynchronize(BR_STREAM));
  for (void* q : drop) (void)hipFree(q);
  return bytes;
}

int dev_current_device() {
  int d = 0;
  HIP_CHECK(hipGetDevice(&d));
  return d;
}
void dev_use_device(int device) { HIP_CHECK(hipSetDevice(device)); }
int dev_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* dev_name() {
  static std::string name;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      name = std::string("hip:") + prop.gcnArchName + " (" + prop.name + ")";
    } else {
      name = "hip:unavailable";
    }
  });
  return name.c_str();
}

// One copy of the tables per device (multi-GPU: one process per GPU, so in practice one).
struct TablesHolder {
  int device = -1;
  DeviceTables t{};
};

const DeviceTables& dev_tables() {
  static std::mutex mu;
  static TablesHolder holders[16];
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  TablesHolder& h = holders[dev & 15];
  if (h.device == dev) return h.t;
  auto upload = [](const void* src, size_t bytes) -> void* {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes + 64));  // padded: the match-length code reads 32 bytes at a time
    HIP_CHECK(hipMemset(p, 0, bytes + 64));
    HIP_CHECK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return p;
  };
  h.t.dict_hash = (const uint16_t*)upload(kBrotliStaticDictionaryHash, sizeof(kBrotliStaticDictionaryHash));
  h.t.dict_data = (const uint8_t*)upload(kBrotliDictionaryData, sizeof(kBrotliDictionaryData));
  h.t.dict_offsets_by_length = (const uint32_t*)upload(kBrotliDictionaryOffsetsByLength, sizeof(kBrotliDictionaryOffsetsByLength));
  h.t.dict_size_bits_by_length = (const uint8_t*)upload(kBrotliDictionarySizeBitsByLength, sizeof(kBrotliDictionarySizeBitsByLength));
  h.t.logs_16 = (const float*)upload(kBrotliLog2Table16_bits, sizeof(kBrotliLog2Table16_bits));
  h.t.logs_8 = (const float*)upload(kBrotliLog2Table8_bits, sizeof(kBrotliLog2Table8_bits));
  h.t.utf8_context_lookup = (const uint8_t*)upload(kBrotliUTF8ContextLookup, sizeof(kBrotliUTF8ContextLookup));
  h.t.signed_context_lookup = (const uint8_t*)upload(kBrotliSigned3BitContextLookup, sizeof(kBrotliSigned3BitContextLookup));
  h.t.dict_lut_buckets = (const uint16_t*)upload(kStaticDictionaryBuckets, sizeof(kStaticDictionaryBuckets));
  h.t.dict_lut_words = (const uint32_t*)upload(kStaticDictionaryWords, sizeof(kStaticDictionaryWords));
  h.device = dev;
  return h.t;
}

}  // namespace brotli_mi355x
