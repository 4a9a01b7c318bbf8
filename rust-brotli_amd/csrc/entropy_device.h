// entropy_device.h -- the reference's f32 entropy arithmetic for device threads (and, compiled for the host, the emulation):
// src/enc/bit_cost.rs:13-42, util.rs:17-25.  Sequential sums in the reference's order; include after the BR_DEV macros
// (lz77_chain.h does).
#ifndef BROTLI_MI355X_ENTROPY_DEVICE_H_
#define BROTLI_MI355X_ENTROPY_DEVICE_H_

#include <stdint.h>

namespace brotli_mi355x {

struct EntropyTables {
  const float* logs_16;  // 65536 entries
  const float* logs_8;   // 256 entries
};

// glibc log2f (sysdeps/ieee754/flt-32/e_log2f.c), restated in exact double arithmetic so that the device
// result equals the libm result the reference relies on (util.rs:23).  Valid for finite x >= 1.
BR_DEV float br_log2f(float x) {
  const double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
      {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
      {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
      {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
      {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
      {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
  const double A0 = -0x1.712b6f70a7e4dp-2, A1 = 0x1.ecabf496832ep-2, A2 = -0x1.715479ffae3dep-1, A3 = 0x1.715475f35c8b8p0;
  uint32_t ix;
  __builtin_memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = (int32_t)tmp >> 23;
  float zf;
  __builtin_memcpy(&zf, &iz, 4);
  const double z = (double)zf;
  const double r = z * T[i][0] - 1.0;
  const double y0 = T[i][1] + (double)k;
  const double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  const double p = A3 * r + y0;
  y = y * r2 + p;
  return (float)y;
}

BR_DEV float br_fast_log2(const EntropyTables& t, uint32_t v) { return v < 256 ? t.logs_8[v] : br_log2f((float)v); }

// BitsEntropy, bit_cost.rs:13-42: strictly left-to-right f32 accumulation over `size` bins
BR_DEV float br_bits_entropy(const EntropyTables& t, const uint32_t* population, uint32_t size) {
  uint32_t sum = 0;
  float retval = 0.0f;
  for (uint32_t i = 0; i < size; ++i) {
    const uint32_t p = population[i];
    sum += p;
    const float term = (float)p * t.logs_16[p & 0xffffu];
    retval = retval - term;
  }
  if (sum != 0) {
    const float term = (float)sum * br_fast_log2(t, sum);
    retval = retval + term;
  }
  const float fsum = (float)sum;
  return retval < fsum ? fsum : retval;
}

}  // namespace brotli_mi355x
#endif
