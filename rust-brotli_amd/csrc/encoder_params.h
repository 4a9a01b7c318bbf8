// encoder_params.h -- encoder parameters of the accelerated path (host side).
//
// Mirrors BrotliEncoderParams (reference src/enc/backward_references/mod.rs:71-125), its defaults
// (src/enc/encode.rs:318-357), the setter (encode.rs:196-286), SanitizeParams (:546-568),
// ComputeLgBlock (:570-585), ChooseDistanceParams (:2169-2190) and ChooseHasher (:834-893).
#ifndef BROTLI_MI355X_ENCODER_PARAMS_H_
#define BROTLI_MI355X_ENCODER_PARAMS_H_

#include <stddef.h>
#include <stdint.h>

namespace brotli_mi355x {

// BrotliEncoderParameter ids, reference src/enc/parameters.rs:3-33
enum ParamId {
  kParamMode = 0,
  kParamQuality = 1,
  kParamLgwin = 2,
  kParamLgblock = 3,
  kParamDisableLiteralContextModeling = 4,
  kParamSizeHint = 5,
  kParamLargeWindow = 6,
  kParamQ9_5 = 150,
  kParamMetablockCallback = 151,
  kParamStrideDetectionQuality = 152,
  kParamHighEntropyDetectionQuality = 153,
  kParamLiteralByteScore = 154,
  kParamCdfAdaptationDetection = 155,
  kParamPriorBitmaskDetection = 156,
  kParamSpeed = 157,
  kParamSpeedMax = 158,
  kParamCmSpeed = 159,
  kParamCmSpeedMax = 160,
  kParamSpeedLow = 161,
  kParamSpeedLowMax = 162,
  kParamCmSpeedLow = 164,
  kParamCmSpeedLowMax = 165,
  kParamAvoidDistancePrefixSearch = 166,
  kParamCatable = 167,
  kParamAppendable = 168,
  kParamMagicNumber = 169,
  kParamNoDictionary = 170,
  kParamFavorEfficiency = 171,
  kParamByteAlign = 172,
  kParamBareStream = 173,
};

struct HasherParams {
  int type = 6;
  int bucket_bits = 15;
  int block_bits = 8;
  int hash_len = 5;
  int num_last_distances_to_check = 16;
  int literal_byte_score = 0;
};

struct DistanceParams {
  uint32_t distance_postfix_bits = 0;
  uint32_t num_direct_distance_codes = 0;
  uint32_t alphabet_size = 16 + (24u << 1);
  size_t max_distance = 0x03fffffc;
};

struct EncoderParams {
  DistanceParams dist;
  int mode = 0;
  int quality = 11;
  bool q9_5 = false;
  int lgwin = 22;
  int lgblock = 0;
  size_t size_hint = 0;
  int disable_literal_context_modeling = 0;
  HasherParams hasher;
  bool large_window = false;
  bool byte_align = false;
  bool bare_stream = false;
  bool catable = false;
  bool use_dictionary = true;
  bool appendable = false;
  bool magic_number = false;
  bool favor_cpu_efficiency = false;
  // research knobs of the reference that never change the default output; stored, not acted upon
  uint32_t ignored_research_knobs = 0;
};

// encode.rs:196-286.  Returns false for unknown ids / invalid values.
bool SetParameter(EncoderParams* params, int id, uint32_t value);
// encode.rs:546-568, :570-585, :2169-2190 -- what ensure_initialized() does to the parameters
void FinalizeParams(EncoderParams* params);
// encode.rs:834-893
void ChooseHasher(EncoderParams* params);
// Hasher types 40 / 41 / 42 (lgwin <= 16 at quality 5..8, encode.rs:855-862) have no implementation of their own in the
// reference: BrotliMakeHasher falls through to InitializeH6 (encode.rs:1096-1114) with the hasher parameters nobody has
// touched -- bucket_bits 15, block_bits 8 (256-deep rings), hash_len 5, 16 last distances (encode.rs:348-355).
inline bool IsH6Family(int type) { return type == 6 || type == 40 || type == 41 || type == 42; }
// true when (quality, hasher) is covered by the gfx950 kernels of this build
bool IsAccelerated(const EncoderParams& params, const char** why_not);

inline int ComputeRbBits(const EncoderParams& p) { return 1 + (p.lgwin > p.lgblock ? p.lgwin : p.lgblock); }
inline size_t MaxMetablockSize(const EncoderParams& p) {
  int b = ComputeRbBits(p);
  return (size_t)1 << (b < 24 ? b : 24);
}
size_t MaxCompressedSize(size_t input_size);                          // encode.rs:1276-1299
size_t MaxCompressedSizeMulti(size_t input_size, size_t num_threads);  // encode.rs:1272-1274

}  // namespace brotli_mi355x
#endif
