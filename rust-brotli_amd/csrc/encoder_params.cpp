// encoder_params.cpp -- see encoder_params.h
#include "encoder_params.h"

#include <stdlib.h>

#include <algorithm>

namespace brotli_mi355x {

bool SetParameter(EncoderParams* params, int id, uint32_t value) {
  switch (id) {
    case kParamMode: params->mode = value <= 6 ? (int)value : 0; return true;
    case kParamQuality: params->quality = (int)value; return true;
    case kParamQ9_5: params->q9_5 = value != 0; return true;
    case kParamLiteralByteScore: params->hasher.literal_byte_score = (int)value; return true;
    case kParamLgwin: params->lgwin = (int)value; return true;
    case kParamLgblock: params->lgblock = (int)value; return true;
    case kParamDisableLiteralContextModeling:
      if (value != 0 && value != 1) return false;
      params->disable_literal_context_modeling = value != 0;
      return true;
    case kParamSizeHint: params->size_hint = value; return true;
    case kParamLargeWindow: params->large_window = value != 0; return true;
    case kParamCatable:
      params->catable = value != 0;
      if (!params->appendable) params->appendable = value != 0;
      params->use_dictionary = (value == 0);
      return true;
    case kParamAppendable: params->appendable = value != 0; return true;
    case kParamMagicNumber: params->magic_number = value != 0; return true;
    case kParamFavorEfficiency: params->favor_cpu_efficiency = value != 0; return true;
    case kParamByteAlign: params->byte_align = value != 0; return true;
    case kParamBareStream:
      params->bare_stream = value != 0;
      if (!params->byte_align) params->byte_align = value != 0;
      return true;
    // accepted and remembered, but these only drive the reference's optional IR/prior research code
    case kParamMetablockCallback:
    case kParamStrideDetectionQuality:
    case kParamHighEntropyDetectionQuality:
    case kParamCdfAdaptationDetection:
    case kParamPriorBitmaskDetection:
    case kParamSpeed:
    case kParamSpeedMax:
    case kParamCmSpeed:
    case kParamCmSpeedMax:
    case kParamSpeedLow:
    case kParamSpeedLowMax:
    case kParamCmSpeedLow:
    case kParamCmSpeedLowMax:
    case kParamAvoidDistancePrefixSearch:
      if (value != 0) params->ignored_research_knobs |= 1u;
      return true;
    default: return false;
  }
}

void FinalizeParams(EncoderParams* p) {
  // SanitizeParams
  p->quality = std::min(11, std::max(0, p->quality));
  if (p->lgwin < 10) {
    p->lgwin = 10;
  } else if (p->lgwin > 24) {
    if (p->large_window) {
      if (p->lgwin > 30) p->lgwin = 30;
    } else {
      p->lgwin = 24;
    }
  }
  if (p->catable) {
    p->appendable = true;
    p->use_dictionary = false;
  }
  if (p->bare_stream) {
    p->byte_align = true;
  } else if (!p->appendable) {
    p->byte_align = false;
  }
  // ComputeLgBlock
  int lgblock = p->lgblock;
  if (p->quality == 0 || p->quality == 1) {
    lgblock = p->lgwin;
  } else if (p->quality < 4) {
    lgblock = 14;
  } else if (lgblock == 0) {
    lgblock = 16;
    if (p->quality >= 9 && p->lgwin > lgblock) lgblock = std::min(18, p->lgwin);
  } else {
    lgblock = std::min(24, std::max(16, lgblock));
  }
  p->lgblock = lgblock;
  // ChooseDistanceParams + BrotliInitDistanceParams (metablock.rs:28-60)
  uint32_t ndirect = 0, npostfix = 0;
  if (p->quality >= 4) {
    if (p->mode == 2) {
      npostfix = 1;
      ndirect = 12;
    } else {
      npostfix = p->dist.distance_postfix_bits;
      ndirect = p->dist.num_direct_distance_codes;
    }
    const uint32_t ndirect_msb = (ndirect >> npostfix) & 0x0f;
    if (npostfix > 3 || ndirect > 120 || (ndirect_msb << npostfix) != ndirect) {
      npostfix = 0;
      ndirect = 0;
    }
  }
  p->dist.distance_postfix_bits = npostfix;
  p->dist.num_direct_distance_codes = ndirect;
  uint32_t alphabet_size = 16 + ndirect + (24u << (npostfix + 1));
  uint32_t max_distance = ndirect + (1u << (24 + npostfix + 2)) - (1u << (npostfix + 2));
  if (p->large_window) {
    static const uint32_t bound[4] = {0, 4, 12, 28};
    const uint32_t postfix = 1u << npostfix;
    alphabet_size = 16 + ndirect + (62u << (npostfix + 1));
    if (ndirect < bound[npostfix]) {
      max_distance = 0x07fffffcu - (bound[npostfix] - ndirect);
    } else if (ndirect >= bound[npostfix] + postfix) {
      max_distance = (3u << 29) - 4 + (ndirect - bound[npostfix]);
    } else {
      max_distance = 0x07fffffcu;
    }
  }
  p->dist.alphabet_size = alphabet_size;
  p->dist.max_distance = max_distance;
}

void ChooseHasher(EncoderParams* params) {
  HasherParams* hp = &params->hasher;
  if (params->quality >= 10 && !params->q9_5) {
    hp->type = 10;
  } else if (params->quality == 10 || params->quality == 9) {
    hp->type = 9;
    hp->num_last_distances_to_check = 16;
    hp->block_bits = 8;
    hp->bucket_bits = 15;
    hp->hash_len = 4;
  } else if (params->quality == 4 && params->size_hint >= (1u << 20)) {
    hp->type = 54;
  } else if (params->quality < 5) {
    hp->type = params->quality;
  } else if (params->lgwin <= 16) {
    hp->type = params->quality < 7 ? 40 : (params->quality < 9 ? 41 : 42);
  } else if (((params->q9_5 && params->size_hint > (1u << 20)) || params->size_hint > (1u << 22)) && params->lgwin >= 19) {
    hp->type = 6;
    hp->block_bits = std::min(params->quality - 1, 9);
    hp->bucket_bits = 15;
    hp->hash_len = 5;
    hp->num_last_distances_to_check = params->quality < 7 ? 4 : (params->quality < 9 ? 10 : 16);
  } else {
    hp->type = 5;
    hp->block_bits = std::min(params->quality - 1, 9);
    hp->bucket_bits = (params->quality < 7 && params->size_hint <= (1u << 20)) ? 14 : 15;
    hp->num_last_distances_to_check = params->quality < 7 ? 4 : (params->quality < 9 ? 10 : 16);
  }
}

// The 512-deep rings of quality 11 + Q9_5 run on kernels of their own (ChainScratchT<.., kDeep>).  On the hardware: the
// reference's 129 715-byte known answer and the wide identity set (H5 and H6 at depth 512, lgwin 18 / 20 / 22, streamed pieces,
// the API sweep), part of the default -m gpu suite since round 4 (profiles/r04_deep_rings_wide_gpu.log).  The switch is a
// diagnosis aid: with it set such parameters are refused, not taken another way.
static bool DeepRingsAllowed() {
  static const bool off = getenv("BROTLI_MI355X_NO_DEEP_RINGS") != nullptr;
  return !off;
}

bool IsAccelerated(const EncoderParams& p, const char** why_not) {
  const char* why = nullptr;
  const bool basic = p.hasher.type == 2 || p.hasher.type == 3 || p.hasher.type == 4 || p.hasher.type == 54;
  if (p.quality < 2 || p.quality > 11) {
    why = "qualities 0 and 1 run through fragment_stream.h (BrotliEncoderCompress, BrotliEncoderCompressStream, BrotliEncoderCompressMulti), not through this entry";
  } else if (basic != (p.quality < 5)) {
    why = "hasher type and quality do not go together";
  } else if (basic) {
    // BasicHasher family under the greedy / lazy parse (quick_device.h)
  } else if (p.hasher.type != 5 && !IsH6Family(p.hasher.type) && p.hasher.type != 9 && p.hasher.type != 10) {
    why = "hasher type not implemented on the device";
  } else if (p.hasher.block_bits > 9) {
    why = "ring depth above 512 does not occur in the reference";
  } else if (p.hasher.block_bits > 8 && !DeepRingsAllowed()) {
    why = "the 512-deep rings of quality 11 + Q9_5 are switched off (BROTLI_MI355X_NO_DEEP_RINGS)";
  }
  if (why_not) *why_not = why;
  return why == nullptr;
}

size_t MaxCompressedSize(size_t input_size) {
  const size_t magic_size = 16;
  const size_t num_large_blocks = input_size >> 14;
  const size_t tail = input_size - (num_large_blocks << 24);
  const size_t tail_overhead = tail > (1u << 20) ? 4 : 3;
  const size_t overhead = 2 + 4 * num_large_blocks + tail_overhead + 1;
  const size_t result = input_size + overhead;
  if (input_size == 0) return 1 + magic_size;
  if (result < input_size) return 0;
  return result + magic_size;
}
size_t MaxCompressedSizeMulti(size_t input_size, size_t num_threads) { return MaxCompressedSize(input_size) + num_threads * 8; }

}  // namespace brotli_mi355x
