// host_entropy.h -- f32 entropy helpers used by the host driver for the few decisions it takes itself
// (should_compress sampling, context-map choice).  Reference: src/enc/bit_cost.rs:13-42,
// src/enc/util.rs:17-25.  Sums are evaluated left to right in f32 (compile with -ffp-contract=off).
#ifndef BROTLI_MI355X_HOST_ENTROPY_H_
#define BROTLI_MI355X_HOST_ENTROPY_H_
#include <stddef.h>
#include <stdint.h>
namespace brotli_mi355x {
float HostFastLog2(uint64_t v);
float HostShannonEntropy(const uint32_t* population, size_t size, size_t* total);
float HostBitsEntropy(const uint32_t* population, size_t size);
}  // namespace brotli_mi355x
#endif
