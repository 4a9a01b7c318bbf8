// device_runtime.hip -- device memory helpers and the read-only tables of the encoder hot path.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdexcept>
#include <string>

#include "device_api.h"
#include "../../tables/brotli_tables.h"

namespace brotli_mi355x {

void hip_check(hipError_t e, const char* what);
#define HIP_CHECK(x) hip_check((x), #x)

void* dev_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  HIP_CHECK(hipMalloc(&p, bytes));
  HIP_CHECK(hipMemsetAsync(p, 0, bytes, 0));
  return p;
}
void dev_free(void* p) {
  if (p) (void)hipFree(p);
}
void dev_memset(void* p, int value, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemsetAsync(p, value, bytes, 0));
}
void dev_h2d(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, 0));
}
void dev_d2h(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
}
void dev_d2d(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
}
void dev_sync() { HIP_CHECK(hipStreamSynchronize(0)); }

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
    HIP_CHECK(hipMalloc(&p, bytes));
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
  h.device = dev;
  return h.t;
}

}  // namespace brotli_mi355x
