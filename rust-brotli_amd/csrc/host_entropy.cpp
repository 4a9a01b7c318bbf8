// host_entropy.cpp -- see host_entropy.h
#include "host_entropy.h"

#include <math.h>
#include <string.h>

#include "../../tables/brotli_tables.h"

namespace brotli_mi355x {

static inline float Bits2Float(uint32_t b) {
  float f;
  memcpy(&f, &b, 4);
  return f;
}

float HostFastLog2(uint64_t v) {
  if (v < 256) return Bits2Float(kBrotliLog2Table8_bits[v]);
  return log2f((float)v);  // util.rs:23: (v as f32).log2(), i.e. the platform libm
}

float HostShannonEntropy(const uint32_t* population, size_t size, size_t* total) {
  size_t sum = 0;
  float retval = 0.0f;
  for (size_t i = 0; i < size; ++i) {
    const size_t p = population[i];
    sum += p;
    retval -= (float)p * Bits2Float(kBrotliLog2Table16_bits[(uint16_t)p]);
  }
  if (sum != 0) retval += (float)sum * HostFastLog2(sum);
  if (total) *total = sum;
  return retval;
}

float HostBitsEntropy(const uint32_t* population, size_t size) {
  size_t sum;
  float retval = HostShannonEntropy(population, size, &sum);
  if (retval < (float)sum) retval = (float)sum;
  return retval;
}

}  // namespace brotli_mi355x
