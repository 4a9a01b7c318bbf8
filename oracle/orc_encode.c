/* oracle/orc_encode.c -- CPU restatement of rust-brotli's encoder state machine (src/enc/encode.rs)
 * for qualities 4..9.  TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows src/enc/encode.rs: params :196-357, SanitizeParams :546-568, ComputeLgBlock :570-585,
 * ring buffer :587-832, custom dictionary :1196-1270, should_compress :1325-1354, context-map
 * heuristics :1717-1927, WriteMetaBlockInternal :1941-2167, encode_data :2214-2543,
 * compress_stream :2873-2995, encoder_compress :1436-1538.
 */
#include <stdio.h>

#include "orc_internal.h"

enum { STREAM_PROCESSING = 0, STREAM_FLUSH_REQUESTED = 1, STREAM_FINISHED = 2, STREAM_METADATA_HEAD = 3, STREAM_METADATA_BODY = 4 };
enum { FIRST_NOTHING = 0, FIRST_HEADER, FIRST_ONE_CATABLE_BYTE, FIRST_BOTH_CATABLE_BYTES };
enum { NEXT_OUT_NONE = 0, NEXT_OUT_STORAGE, NEXT_OUT_TINY };

typedef struct {
  uint32_t size_, mask_, tail_size_, total_size_, cur_size_, pos_;
  uint8_t* data_mo; /* allocation; logical buffer starts at buffer_index (=2) */
  size_t buffer_index;
  size_t alloc_len;
} RingBuffer;

struct OrcEncoder {
  EncoderParams params;
  Hasher hasher_;
  uint64_t input_pos_;
  RingBuffer ringbuffer_;
  size_t cmd_alloc_size_;
  Command* commands_;
  size_t num_commands_;
  size_t num_literals_;
  size_t last_insert_len_;
  uint64_t last_flush_pos_;
  uint64_t last_processed_pos_;
  int32_t dist_cache_[16];
  int32_t saved_dist_cache_[4];
  uint16_t last_bytes_;
  uint8_t last_bytes_bits_;
  uint8_t prev_byte_, prev_byte2_;
  size_t storage_size_;
  uint8_t* storage_;
  int next_out_kind;
  size_t next_out_off;
  size_t available_out_;
  uint64_t total_out_;
  uint8_t tiny_buf_[16];
  /* quality 0: the command code carried from one fragment to the next (encode.rs:172-176) */
  uint8_t cmd_depths_[128];
  uint16_t cmd_bits_[128];
  uint8_t cmd_code_[512];
  size_t cmd_code_numbits_;
  uint32_t remaining_metadata_bytes_;
  int stream_state_;
  int is_last_block_emitted_;
  int is_initialized_;
  int is_first_mb;
  size_t custom_dictionary_size; /* 0 = None */
  int custom_dictionary;
  OrcStats stats;
  OrcMetablockTrace trace;
  void* trace_opaque;
};

/* encode.rs:318-357 */
static void init_params(EncoderParams* p) {
  memset(p, 0, sizeof(*p));
  p->dist.distance_postfix_bits = 0;
  p->dist.num_direct_distance_codes = 0;
  p->dist.alphabet_size = 16 + 0 + (24u << 1);
  p->dist.max_distance = 0x03fffffc;
  p->mode = 0;
  p->quality = 11;
  p->lgwin = 22;
  p->use_dictionary = 1;
  p->hasher.type_ = 6;
  p->hasher.block_bits = 9 - 1;
  p->hasher.bucket_bits = 15;
  p->hasher.hash_len = 5;
  p->hasher.num_last_distances_to_check = 16;
  p->hasher.literal_byte_score = 0;
}

OrcEncoder* orc_encoder_create(void) {
  OrcEncoder* s = (OrcEncoder*)calloc(1, sizeof(OrcEncoder));
  static const int32_t cache[4] = {4, 11, 15, 16};
  init_params(&s->params);
  memcpy(s->dist_cache_, cache, sizeof(cache));
  memcpy(s->saved_dist_cache_, cache, sizeof(cache));
  s->stream_state_ = STREAM_PROCESSING;
  return s;
}

void orc_encoder_destroy(OrcEncoder* s) {
  if (!s) return;
  orc_hasher_free(&s->hasher_);
  free(s->ringbuffer_.data_mo);
  free(s->commands_);
  free(s->storage_);
  free(s);
}

void orc_encoder_set_trace(OrcEncoder* s, OrcMetablockTrace cb, void* opaque) {
  s->trace = cb;
  s->trace_opaque = opaque;
}
const OrcStats* orc_encoder_stats(const OrcEncoder* s) { return &s->stats; }

/* encode.rs:196-286 */
int orc_encoder_set_parameter(OrcEncoder* s, int p, uint32_t value) {
  EncoderParams* params = &s->params;
  if (s->is_initialized_) return 0;
  switch (p) {
    case ORC_PARAM_MODE: params->mode = value <= 6 ? (int)value : 0; return 1;
    case ORC_PARAM_QUALITY: params->quality = (int)value; return 1;
    case ORC_PARAM_Q9_5: params->q9_5 = value != 0; return 1;
    case ORC_PARAM_LITERAL_BYTE_SCORE: params->hasher.literal_byte_score = (int)value; return 1;
    case ORC_PARAM_LGWIN: params->lgwin = (int)value; return 1;
    case ORC_PARAM_LGBLOCK: params->lgblock = (int)value; return 1;
    case ORC_PARAM_DISABLE_LITERAL_CONTEXT_MODELING:
      if (value != 0 && value != 1) return 0;
      params->disable_literal_context_modeling = value != 0;
      return 1;
    case ORC_PARAM_SIZE_HINT: params->size_hint = value; return 1;
    case ORC_PARAM_LARGE_WINDOW: params->large_window = value != 0; return 1;
    case ORC_PARAM_CATABLE:
      params->catable = value != 0;
      if (!params->appendable) params->appendable = value != 0;
      params->use_dictionary = (value == 0);
      return 1;
    case ORC_PARAM_APPENDABLE: params->appendable = value != 0; return 1;
    case ORC_PARAM_MAGIC_NUMBER: params->magic_number = value != 0; return 1;
    case ORC_PARAM_FAVOR_EFFICIENCY: params->favor_cpu_efficiency = value != 0; return 1;
    case ORC_PARAM_BYTE_ALIGN: params->byte_align = value != 0; return 1;
    case ORC_PARAM_BARE_STREAM:
      params->bare_stream = value != 0;
      if (!params->byte_align) params->byte_align = value != 0;
      return 1;
    default: return 0;
  }
}

/* encode.rs:546-568 */
static void sanitize_params(EncoderParams* params) {
  params->quality = ORC_MIN(11, ORC_MAX(0, params->quality));
  if (params->lgwin < 10) {
    params->lgwin = 10;
  } else if (params->lgwin > 24) {
    if (params->large_window) {
      if (params->lgwin > 30) params->lgwin = 30;
    } else {
      params->lgwin = 24;
    }
  }
  if (params->catable) {
    params->appendable = 1;
    params->use_dictionary = 0;
  }
  if (params->bare_stream) {
    params->byte_align = 1;
  } else if (!params->appendable) {
    params->byte_align = 0;
  }
}

/* encode.rs:570-585 */
static int compute_lg_block(const EncoderParams* params) {
  int lgblock = params->lgblock;
  if (params->quality == 0 || params->quality == 1) {
    lgblock = params->lgwin;
  } else if (params->quality < 4) {
    lgblock = 14;
  } else if (lgblock == 0) {
    lgblock = 16;
    if (params->quality >= 9 && params->lgwin > lgblock) lgblock = ORC_MIN(18, params->lgwin);
  } else {
    lgblock = ORC_MIN(24, ORC_MAX(16, lgblock));
  }
  return lgblock;
}

static int compute_rb_bits(const EncoderParams* params) { return 1 + ORC_MAX(params->lgwin, params->lgblock); }

/* metablock.rs:28-60 */
void orc_init_distance_params(EncoderParams* params, uint32_t npostfix, uint32_t ndirect) {
  params->dist.distance_postfix_bits = npostfix;
  params->dist.num_direct_distance_codes = ndirect;
  uint32_t alphabet_size = 16 + ndirect + (24u << (npostfix + 1));
  uint32_t max_distance = ndirect + (1u << (24 + npostfix + 2)) - (1u << (npostfix + 2));
  if (params->large_window) {
    static const uint32_t bound[4] = {0, 4, 12, 28};
    uint32_t postfix = 1u << npostfix;
    alphabet_size = 16 + ndirect + (62u << (npostfix + 1));
    if (ndirect < bound[npostfix]) {
      max_distance = 0x07fffffcu - (bound[npostfix] - ndirect);
    } else if (ndirect >= bound[npostfix] + postfix) {
      max_distance = (3u << 29) - 4 + (ndirect - bound[npostfix]);
    } else {
      max_distance = 0x07fffffcu;
    }
  }
  params->dist.alphabet_size = alphabet_size;
  params->dist.max_distance = max_distance;
}

/* encode.rs:2169-2190 */
static void choose_distance_params(EncoderParams* params) {
  uint32_t ndirect = 0, npostfix = 0;
  if (params->quality >= 4) {
    if (params->mode == 2 /* FONT */) {
      npostfix = 1;
      ndirect = 12;
    } else {
      npostfix = params->dist.distance_postfix_bits;
      ndirect = params->dist.num_direct_distance_codes;
    }
    uint32_t ndirect_msb = (ndirect >> npostfix) & 0x0f;
    if (npostfix > 3 || ndirect > 120 || (ndirect_msb << npostfix) != ndirect) {
      npostfix = 0;
      ndirect = 0;
    }
  }
  orc_init_distance_params(params, npostfix, ndirect);
}

/* encode.rs:603-625 */
static void encode_window_bits(int lgwin, int large_window, uint16_t* last_bytes, uint8_t* last_bytes_bits) {
  if (large_window) {
    *last_bytes = (uint16_t)(((lgwin & 0x3F) << 8) | 0x11);
    *last_bytes_bits = 14;
  } else if (lgwin == 16) {
    *last_bytes = 0;
    *last_bytes_bits = 1;
  } else if (lgwin == 17) {
    *last_bytes = 1;
    *last_bytes_bits = 7;
  } else if (lgwin > 17) {
    *last_bytes = (uint16_t)(((lgwin - 17) << 1) | 1);
    *last_bytes_bits = 4;
  } else {
    *last_bytes = (uint16_t)(((lgwin - 8) << 4) | 1);
    *last_bytes_bits = 7;
  }
}

/* encode.rs:657-707 */
static int ensure_initialized(OrcEncoder* s) {
  if (s->is_initialized_) return 1;
  sanitize_params(&s->params);
  s->params.lgblock = compute_lg_block(&s->params);
  choose_distance_params(&s->params);
  s->remaining_metadata_bytes_ = 0xffffffffu;
  {
    RingBuffer* rb = &s->ringbuffer_;
    int window_bits = compute_rb_bits(&s->params);
    int tail_bits = s->params.lgblock;
    rb->size_ = 1u << window_bits;
    rb->mask_ = (1u << window_bits) - 1;
    rb->tail_size_ = 1u << tail_bits;
    rb->total_size_ = rb->size_ + rb->tail_size_;
  }
  {
    int lgwin = s->params.lgwin;
    if (s->params.quality == 0 || s->params.quality == 1) lgwin = ORC_MAX(lgwin, 18);
    if (!(s->params.catable && s->params.bare_stream))
      encode_window_bits(lgwin, s->params.large_window, &s->last_bytes_, &s->last_bytes_bits_);
  }
  if (s->params.quality == 0) orc_init_command_prefix_codes(s->cmd_depths_, s->cmd_bits_, s->cmd_code_, &s->cmd_code_numbits_);
  if (s->params.catable) {
    for (int i = 0; i < 16; ++i) s->dist_cache_[i] = 0x7ffffff0;
    for (int i = 0; i < 4; ++i) s->saved_dist_cache_[i] = 0x7ffffff0;
  }
  s->is_initialized_ = 1;
  return 1;
}

/* encode.rs:709-739 */
static void ring_buffer_init_buffer(uint32_t buflen, RingBuffer* rb) {
  const size_t kSlack = 7;
  size_t n = 2 + (size_t)buflen + kSlack;
  uint8_t* new_data = (uint8_t*)calloc(n, 1);
  if (rb->data_mo) {
    size_t lim = 2 + (size_t)rb->cur_size_ + kSlack;
    memcpy(new_data, rb->data_mo, lim);
    free(rb->data_mo);
  }
  rb->data_mo = new_data;
  rb->alloc_len = n;
  rb->cur_size_ = buflen;
  rb->buffer_index = 2;
  rb->data_mo[0] = 0;
  rb->data_mo[1] = 0;
  for (size_t i = 0; i < kSlack; ++i) rb->data_mo[rb->buffer_index + rb->cur_size_ + i] = 0;
}

/* encode.rs:741-810 */
static void ring_buffer_write(const uint8_t* bytes, size_t n, RingBuffer* rb) {
  if (rb->pos_ == 0 && n < rb->tail_size_) {
    rb->pos_ = (uint32_t)n;
    ring_buffer_init_buffer(rb->pos_, rb);
    memcpy(rb->data_mo + rb->buffer_index, bytes, n);
    return;
  }
  if (rb->cur_size_ < rb->total_size_) {
    ring_buffer_init_buffer(rb->total_size_, rb);
    rb->data_mo[rb->buffer_index + rb->size_ - 2] = 0;
    rb->data_mo[rb->buffer_index + rb->size_ - 1] = 0;
  }
  {
    size_t masked_pos = rb->pos_ & rb->mask_;
    if (masked_pos < rb->tail_size_) { /* RingBufferWriteTail */
      size_t p = (size_t)rb->size_ + masked_pos;
      size_t lim = ORC_MIN(n, (size_t)rb->tail_size_ - masked_pos);
      memcpy(rb->data_mo + rb->buffer_index + p, bytes, lim);
    }
    if (masked_pos + n <= rb->size_) {
      memcpy(rb->data_mo + rb->buffer_index + masked_pos, bytes, n);
    } else {
      size_t mid = ORC_MIN(n, (size_t)rb->total_size_ - masked_pos);
      memcpy(rb->data_mo + rb->buffer_index + masked_pos, bytes, mid);
      size_t size = n - ((size_t)rb->size_ - masked_pos);
      size_t bytes_start = (size_t)rb->size_ - masked_pos;
      memcpy(rb->data_mo + rb->buffer_index, bytes + bytes_start, size);
    }
  }
  rb->data_mo[rb->buffer_index - 2] = rb->data_mo[rb->buffer_index + rb->size_ - 2];
  rb->data_mo[rb->buffer_index - 1] = rb->data_mo[rb->buffer_index + rb->size_ - 1];
  rb->pos_ += (uint32_t)n;
  if (rb->pos_ > (1u << 30)) rb->pos_ = (rb->pos_ & ((1u << 30) - 1)) | (1u << 30);
}

/* encode.rs:812-831 */
static void copy_input_to_ring_buffer(OrcEncoder* s, size_t input_size, const uint8_t* input_buffer) {
  if (!ensure_initialized(s)) return;
  ring_buffer_write(input_buffer, input_size, &s->ringbuffer_);
  s->input_pos_ += input_size;
  if (s->ringbuffer_.pos_ <= s->ringbuffer_.mask_) {
    memset(s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index + s->ringbuffer_.pos_, 0, 7);
  }
}

/* encode.rs:1196-1270 */
void orc_encoder_set_custom_dictionary(OrcEncoder* s, size_t size, const uint8_t* dict,
                                       int is_multithreading_file_continue) {
  s->params.use_dictionary = 0;
  s->prev_byte_ = 0;
  s->prev_byte2_ = 0;
  if (is_multithreading_file_continue) {
    if (size > 0) s->prev_byte_ = dict[size - 1];
    if (size > 1) s->prev_byte2_ = dict[size - 2];
  }
  if (!ensure_initialized(s)) return;
  size_t max_dict_size = ((size_t)1 << s->params.lgwin) - 16;
  size_t dict_size = size;
  if (dict_size == 0 || s->params.quality == 0 || s->params.quality == 1 || size <= 1) {
    s->params.catable = 1;
    s->params.appendable = 1;
    return;
  }
  s->custom_dictionary = 1;
  if (size > max_dict_size) {
    dict += size - max_dict_size;
    dict_size = max_dict_size;
  }
  s->custom_dictionary_size = dict_size;
  copy_input_to_ring_buffer(s, dict_size, dict);
  s->last_flush_pos_ = dict_size;
  s->last_processed_pos_ = dict_size;
  orc_hasher_prepend_dictionary(&s->hasher_, &s->params, s->custom_dictionary_size, dict_size, dict, &s->stats);
}

/* encode.rs:1272-1299 */
size_t orc_max_compressed_size(size_t input_size) {
  size_t magic_size = 16;
  size_t num_large_blocks = input_size >> 14;
  size_t tail = input_size - (num_large_blocks << 24);
  size_t tail_overhead = tail > (1u << 20) ? 4 : 3;
  size_t overhead = 2 + 4 * num_large_blocks + tail_overhead + 1;
  size_t result = input_size + overhead;
  if (input_size == 0) return 1 + magic_size;
  if (result < input_size) return 0;
  return result + magic_size;
}
size_t orc_max_compressed_size_multi(size_t input_size, size_t num_threads) {
  return orc_max_compressed_size(input_size) + num_threads * 8;
}

/* encode.rs:1325-1354 */
static int should_compress(const uint8_t* data, size_t mask, uint64_t last_flush_pos, size_t bytes,
                           size_t num_literals, size_t num_commands) {
  const uint32_t kSampleRate = 13;
  const float kMinEntropy = 7.92f;
  if (num_commands < (bytes >> 8) + 2 && (float)num_literals > 0.99f * (float)bytes) {
    uint32_t literal_histo[256] = {0};
    float bit_cost_threshold = (float)bytes * kMinEntropy / (float)kSampleRate;
    size_t t = (bytes + kSampleRate - 1) / kSampleRate;
    uint32_t pos = (uint32_t)last_flush_pos;
    for (size_t i = 0; i < t; ++i) {
      literal_histo[data[pos & mask]]++;
      pos += kSampleRate;
    }
    if (orc_bits_entropy_impl(literal_histo, 256) > bit_cost_threshold) return 0;
  }
  return 1;
}

/* TEST SWITCH (tests/test_streaming.py): log2 of the unit of the position wrap, 30 (GiB) in the reference.  The
   reference folds stream positions from 3 units on back into [1, 3) units and empties its hash table whenever the folded
   position jumps backwards (at 3, 5, 7 ... units).  Scaling the unit down lets the tests drive a stream through several
   wraps with megabytes instead of gigabytes of input (window and ring buffer must then be smaller than one unit). */
int orc_test_wrap_shift = 30;

/* encode.rs:1623-1631 */
static uint32_t wrap_position(uint64_t position) {
  const int sh = orc_test_wrap_shift;
  uint32_t result = (uint32_t)position;
  uint64_t gb = position >> sh;
  if (gb > 2) result = (result & ((1u << sh) - 1)) | ((uint32_t)((gb - 1) & 1) + 1) << sh;
  return result;
}

static float shannon0(const uint32_t* p, size_t n) {
  size_t total;
  return orc_shannon_entropy(p, n, &total);
}

static const uint32_t kStaticContextMapContinuation[64] = {1, 1, 2, 2};
static const uint32_t kStaticContextMapSimpleUTF8[64] = {0, 0, 1, 1};

/* encode.rs:1717-1780 */
static void choose_context_map(int quality, uint32_t* bigram_histo, size_t* num_literal_contexts,
                               const uint32_t** literal_context_map) {
  uint32_t monogram_histo[3] = {0, 0, 0};
  uint32_t two_prefix_histo[6] = {0, 0, 0, 0, 0, 0};
  float entropy[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < 9; ++i) {
    monogram_histo[i % 3] += bigram_histo[i];
    two_prefix_histo[i % 6] += bigram_histo[i];
  }
  entropy[1] = shannon0(monogram_histo, 3);
  entropy[2] = shannon0(two_prefix_histo, 3) + shannon0(two_prefix_histo + 3, 3);
  entropy[3] = 0.0f;
  for (size_t i = 0; i < 3; ++i) entropy[3] += shannon0(bigram_histo + 3 * i, 3);
  size_t total = (size_t)(monogram_histo[0] + monogram_histo[1] + monogram_histo[2]);
  entropy[0] = 1.0f / (float)total;
  entropy[1] *= entropy[0];
  entropy[2] *= entropy[0];
  entropy[3] *= entropy[0];
  if (quality < 7) entropy[3] = entropy[1] * 10.0f;
  if (entropy[1] - entropy[2] < 0.2f && entropy[1] - entropy[3] < 0.2f) {
    *num_literal_contexts = 1;
  } else if (entropy[2] - entropy[3] < 0.02f) {
    *num_literal_contexts = 2;
    *literal_context_map = kStaticContextMapSimpleUTF8;
  } else {
    *num_literal_contexts = 3;
    *literal_context_map = kStaticContextMapContinuation;
  }
}

/* encode.rs:1782-1798 */
static const uint32_t kStaticContextMapComplexUTF8[64] = {
    11, 11, 12, 12, 0, 0, 0, 0, 1, 1, 9, 9, 2, 2, 2, 2, 1, 1, 1, 1, 8, 3, 3, 3, 1, 1, 1, 1, 2, 2, 2, 2,
    8,  4,  4,  4,  8, 7, 4, 4, 8, 0, 0, 0, 3, 3, 3, 3, 5, 5, 10, 5, 5, 5, 10, 5, 6, 6, 6, 6, 6, 6, 6, 6};

/* encode.rs:1802-1871 */
static int should_use_complex_static_context_map(const uint8_t* input, size_t start_pos, size_t length, size_t mask,
                                                 size_t size_hint, size_t* num_literal_contexts,
                                                 const uint32_t** literal_context_map) {
  if (size_hint < (1u << 20)) return 0;
  size_t end_pos = start_pos + length;
  uint32_t combined_histo[32] = {0};
  uint32_t context_histo[13][32];
  uint32_t total = 0;
  float entropy[3];
  memset(context_histo, 0, sizeof(context_histo));
  while (start_pos + 64 <= end_pos) {
    size_t stride_end_pos = start_pos + 64;
    uint8_t prev2 = input[start_pos & mask];
    uint8_t prev1 = input[(start_pos + 1) & mask];
    for (size_t pos = start_pos + 2; pos < stride_end_pos; ++pos) {
      uint8_t literal = input[pos & mask];
      uint8_t context = (uint8_t)kStaticContextMapComplexUTF8[orc_context(prev1, prev2, ORC_CONTEXT_UTF8)];
      total++;
      combined_histo[literal >> 3]++;
      context_histo[context][literal >> 3]++;
      prev2 = prev1;
      prev1 = literal;
    }
    start_pos += 4096;
  }
  entropy[1] = shannon0(combined_histo, 32);
  entropy[2] = 0.0f;
  for (size_t i = 0; i < 13; ++i) entropy[2] += shannon0(context_histo[i], 32);
  entropy[0] = 1.0f / (float)total;
  entropy[1] *= entropy[0];
  entropy[2] *= entropy[0];
  if (entropy[2] > 3.0f || entropy[1] - entropy[2] < 0.2f) return 0;
  *num_literal_contexts = 13;
  *literal_context_map = kStaticContextMapComplexUTF8;
  return 1;
}

/* encode.rs:1873-1927 */
static void decide_over_literal_context_modeling(const uint8_t* input, size_t start_pos, size_t length, size_t mask,
                                                 int quality, size_t size_hint, size_t* num_literal_contexts,
                                                 const uint32_t** literal_context_map) {
  if (quality < 5 || length < 64) {
  } else if (should_use_complex_static_context_map(input, start_pos, length, mask, size_hint, num_literal_contexts,
                                                   literal_context_map)) {
  } else {
    size_t end_pos = start_pos + length;
    uint32_t bigram_prefix_histo[9] = {0};
    static const int lut[4] = {0, 0, 1, 2};
    for (; start_pos + 64 <= end_pos; start_pos += 4096) {
      size_t stride_end_pos = start_pos + 64;
      int prev = lut[input[start_pos & mask] >> 6] * 3;
      for (size_t pos = start_pos + 1; pos < stride_end_pos; ++pos) {
        uint8_t literal = input[pos & mask];
        bigram_prefix_histo[prev + lut[literal >> 6]]++;
        prev = lut[literal >> 6] * 3;
      }
    }
    choose_context_map(quality, bigram_prefix_histo, num_literal_contexts, literal_context_map);
  }
}

/* encode.rs:1928-1940 */
static void write_empty_last_blocks_internal(const EncoderParams* params, size_t* storage_ix, uint8_t* storage) {
  if (params->byte_align) orc_write_padding_meta_block(storage_ix, storage);
  if (!params->bare_stream) orc_write_empty_last_meta_block(storage_ix, storage);
}

/* encode.rs:1941-2167 */
static void write_meta_block_internal(OrcEncoder* s, const uint8_t* data, size_t mask, uint64_t last_flush_pos,
                                      size_t bytes, int is_last, int literal_context_mode, size_t* storage_ix,
                                      uint8_t* storage) {
  const EncoderParams* params = &s->params;
  int actual_is_last = is_last;
  if (params->appendable || params->byte_align) is_last = 0;
  uint32_t wrapped_last_flush_pos = wrap_position(last_flush_pos);
  if (bytes == 0) {
    write_empty_last_blocks_internal(params, storage_ix, storage);
    return;
  }
  if (!should_compress(data, mask, last_flush_pos, bytes, s->num_literals_, s->num_commands_)) {
    memcpy(s->dist_cache_, s->saved_dist_cache_, 4 * sizeof(int32_t));
    orc_store_uncompressed_meta_block(is_last, data, wrapped_last_flush_pos, mask, bytes, storage_ix, storage);
    if (actual_is_last != is_last) write_empty_last_blocks_internal(params, storage_ix, storage);
    s->stats.uncompressed_metablocks++;
    if (s->trace) s->trace(s->trace_opaque, 1, last_flush_pos, bytes, s->commands_, s->num_commands_, s->dist_cache_);
    return;
  }
  size_t saved_byte_location = *storage_ix >> 3;
  uint16_t last_bytes = (uint16_t)((storage[saved_byte_location + 1] << 8) | storage[saved_byte_location]);
  uint8_t last_bytes_bits = (uint8_t)*storage_ix;
  int kind = 0;
  if (params->quality <= 2) {
    orc_store_meta_block_fast(data, wrapped_last_flush_pos, bytes, mask, is_last, params, s->commands_, s->num_commands_,
                              storage_ix, storage);
  } else if (params->quality < 4) {
    orc_store_meta_block_trivial(data, wrapped_last_flush_pos, bytes, mask, is_last, params, s->commands_,
                                 s->num_commands_, storage_ix, storage);
  } else {
    MetaBlockSplit mb;
    EncoderParams block_params = *params;
    if (params->quality < 10) {
      size_t num_literal_contexts = 1;
      const uint32_t* literal_context_map = NULL;
      if (params->disable_literal_context_modeling == 0) {
        decide_over_literal_context_modeling(data, wrapped_last_flush_pos, bytes, mask, params->quality,
                                             params->size_hint, &num_literal_contexts, &literal_context_map);
      }
      orc_build_meta_block_greedy(data, wrapped_last_flush_pos, mask, s->prev_byte_, s->prev_byte2_,
                                  literal_context_mode, num_literal_contexts, literal_context_map, s->commands_,
                                  s->num_commands_, &mb);
    } else {
      /* the distance-parameter search of BrotliBuildMetaBlock works on a per-meta-block copy (encode.rs:1961) */
      orc_build_meta_block(data, wrapped_last_flush_pos, mask, &block_params, s->prev_byte_, s->prev_byte2_, s->commands_,
                           s->num_commands_, literal_context_mode, &mb);
    }
    {
      size_t num_effective_dist_codes = block_params.dist.alphabet_size;
      if (num_effective_dist_codes > ORC_NUM_DISTANCE_HISTO_SYMBOLS) num_effective_dist_codes = ORC_NUM_DISTANCE_HISTO_SYMBOLS;
      orc_optimize_histograms(num_effective_dist_codes, &mb);
    }
    orc_store_meta_block(data, wrapped_last_flush_pos, bytes, mask, s->prev_byte_, s->prev_byte2_, is_last, &block_params,
                         literal_context_mode, s->commands_, s->num_commands_, &mb, storage_ix, storage);
    orc_metablock_destroy(&mb);
  }
  if (bytes + 4 + saved_byte_location < (*storage_ix >> 3)) {
    memcpy(s->dist_cache_, s->saved_dist_cache_, 4 * sizeof(int32_t));
    storage[saved_byte_location] = (uint8_t)last_bytes;
    storage[saved_byte_location + 1] = (uint8_t)(last_bytes >> 8);
    *storage_ix = last_bytes_bits;
    orc_store_uncompressed_meta_block(is_last, data, wrapped_last_flush_pos, mask, bytes, storage_ix, storage);
    s->stats.uncompressed_metablocks++;
    kind = 2;
  }
  if (actual_is_last != is_last) write_empty_last_blocks_internal(params, storage_ix, storage);
  if (s->trace) s->trace(s->trace_opaque, kind, last_flush_pos, bytes, s->commands_, s->num_commands_, s->dist_cache_);
}

static int32_t* get_hash_table(int quality, size_t input_size, size_t* table_size);
static size_t input_block_size(OrcEncoder* s) {
  if (!ensure_initialized(s)) return 0;
  return (size_t)1 << s->params.lgblock;
}
static uint64_t unprocessed_input_size(const OrcEncoder* s) { return s->input_pos_ - s->last_processed_pos_; }

/* encode.rs:1713-1715 */
static size_t max_metablock_size(const EncoderParams* params) { return (size_t)1 << ORC_MIN(compute_rb_bits(params), 24); }

static int update_last_processed_pos(OrcEncoder* s) {
  uint32_t wrapped_last_processed_pos = wrap_position(s->last_processed_pos_);
  uint32_t wrapped_input_pos = wrap_position(s->input_pos_);
  s->last_processed_pos_ = s->input_pos_;
  return wrapped_input_pos < wrapped_last_processed_pos;
}

static void get_brotli_storage(OrcEncoder* s, size_t size) {
  if (s->storage_size_ < size) {
    free(s->storage_);
    s->storage_ = (uint8_t*)calloc(size, 1);
    s->storage_size_ = size;
  }
}

/* encode.rs:360-400 */
static void extend_last_command(OrcEncoder* s, uint32_t* bytes, uint32_t* wrapped_last_processed_pos) {
  Command* last_command = &s->commands_[s->num_commands_ - 1];
  const uint8_t* data = s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index;
  uint32_t mask = s->ringbuffer_.mask_;
  uint64_t max_backward_distance = (1ull << s->params.lgwin) - 16;
  uint64_t last_copy_len = last_command->copy_len_ & 0x01ffffffu;
  uint64_t last_processed_pos = s->last_processed_pos_ - last_copy_len;
  uint64_t max_distance = last_processed_pos < max_backward_distance ? last_processed_pos : max_backward_distance;
  uint64_t cmd_dist = (uint64_t)(int64_t)s->dist_cache_[0];
  uint32_t distance_code = orc_command_restore_distance_code(last_command, &s->params.dist);
  if (distance_code < 16 || (uint64_t)distance_code - 15 == cmd_dist) {
    if (cmd_dist <= max_distance) {
      while (*bytes != 0 && data[*wrapped_last_processed_pos & mask] ==
                                data[(size_t)((size_t)*wrapped_last_processed_pos - (size_t)cmd_dist) & mask]) {
        last_command->copy_len_++;
        (*bytes)--;
        (*wrapped_last_processed_pos)++;
      }
    }
    orc_get_length_code(last_command->insert_len_,
                        (size_t)((int32_t)(last_command->copy_len_ & 0x01ffffffu) + (int32_t)(last_command->copy_len_ >> 25)),
                        (last_command->dist_prefix_ & 0x3ff) == 0, &last_command->cmd_prefix_);
  }
}

/* utf8_util.rs:3-43 */
static size_t parse_as_utf8(const uint8_t* input, size_t size, int32_t* symbol) {
  if ((input[0] & 0x80) == 0) {
    if (input[0] > 0) {
      *symbol = input[0];
      return 1;
    }
  }
  if (size > 1 && (input[0] & 0xe0) == 0xc0 && (input[1] & 0xc0) == 0x80) {
    *symbol = ((input[0] & 0x1f) << 6) | (input[1] & 0x3f);
    if (*symbol > 0x7f) return 2;
  }
  if (size > 2 && (input[0] & 0xf0) == 0xe0 && (input[1] & 0xc0) == 0x80 && (input[2] & 0xc0) == 0x80) {
    *symbol = ((input[0] & 0x0f) << 12) | ((input[1] & 0x3f) << 6) | (input[2] & 0x3f);
    if (*symbol > 0x7ff) return 3;
  }
  if (size > 3 && (input[0] & 0xf8) == 0xf0 && (input[1] & 0xc0) == 0x80 && (input[2] & 0xc0) == 0x80 &&
      (input[3] & 0xc0) == 0x80) {
    *symbol = ((input[0] & 0x07) << 18) | ((input[1] & 0x3f) << 12) | ((input[2] & 0x3f) << 6) | (input[3] & 0x3f);
    if (*symbol > 0xffff && *symbol <= 0x10ffff) return 4;
  }
  *symbol = 0x110000 | input[0];
  return 1;
}

/* utf8_util.rs:45-62 */
int orc_is_mostly_utf8(const uint8_t* data, size_t pos, size_t mask, size_t length, float min_fraction) {
  size_t size_utf8 = 0;
  size_t i = 0;
  while (i < length) {
    int32_t symbol;
    size_t bytes_read = parse_as_utf8(&data[(pos + i) & mask], length - i, &symbol);
    i += bytes_read;
    if (symbol < 0x110000) size_utf8 += bytes_read;
  }
  return (float)size_utf8 > min_fraction * (float)length;
}
#define is_mostly_utf8 orc_is_mostly_utf8

/* encode.rs:2214-2543 */
static int encode_data(OrcEncoder* s, int is_last, int force_flush, size_t* out_size) {
  uint64_t delta = unprocessed_input_size(s);
  uint32_t bytes = (uint32_t)delta;
  uint32_t mask = s->ringbuffer_.mask_;
  if (!ensure_initialized(s)) return 0;
  if (s->is_last_block_emitted_) return 0;
  if (is_last) s->is_last_block_emitted_ = 1;
  if (delta > input_block_size(s)) return 0;
  size_t storage_ix = s->last_bytes_bits_;
  {
    size_t meta_size = ORC_MAX((size_t)bytes, (size_t)(s->input_pos_ - s->last_flush_pos_));
    get_brotli_storage(s, 2 * meta_size + 503 + 24);
  }
  s->storage_[0] = (uint8_t)s->last_bytes_;
  s->storage_[1] = (uint8_t)(s->last_bytes_ >> 8);
  size_t catable_header_size = 0;
  if (s->is_first_mb == FIRST_NOTHING) {
    if (s->params.magic_number) {
      orc_write_metadata_meta_block(&s->params, &storage_ix, s->storage_);
      s->last_bytes_ = (uint16_t)(s->storage_[storage_ix >> 3] | (s->storage_[1 + (storage_ix >> 3)] << 8));
      s->last_bytes_bits_ = (uint8_t)(storage_ix & 7);
      s->next_out_kind = NEXT_OUT_STORAGE;
      s->next_out_off = 0;
      catable_header_size = storage_ix >> 3;
      *out_size = catable_header_size;
      s->is_first_mb = FIRST_HEADER;
    }
    if (bytes == 0 && s->params.byte_align && s->params.appendable && !s->params.catable)
      orc_write_padding_meta_block(&storage_ix, s->storage_);
  }
  if (s->is_first_mb == FIRST_BOTH_CATABLE_BYTES) {
  } else if (!s->params.catable) {
    s->is_first_mb = FIRST_BOTH_CATABLE_BYTES;
  } else if (bytes != 0) {
    size_t n = ORC_MIN((size_t)2, (size_t)bytes);
    const uint8_t* data = s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index;
    orc_store_uncompressed_meta_block(0, data, (size_t)s->last_flush_pos_, mask, n, &storage_ix, s->storage_);
    s->last_bytes_ = (uint16_t)(s->storage_[storage_ix >> 3] | (s->storage_[1 + (storage_ix >> 3)] << 8));
    s->last_bytes_bits_ = (uint8_t)(storage_ix & 7);
    s->prev_byte2_ = s->prev_byte_;
    s->prev_byte_ = data[(size_t)s->last_flush_pos_ & mask];
    if (n == 2) {
      s->prev_byte2_ = s->prev_byte_;
      s->prev_byte_ = data[(size_t)(s->last_flush_pos_ + 1) & mask];
    }
    s->last_flush_pos_ += n;
    bytes -= (uint32_t)n;
    s->last_processed_pos_ += n;
    if (n >= 2) {
      s->is_first_mb = FIRST_BOTH_CATABLE_BYTES;
    } else if (n == 1) {
      s->is_first_mb = (s->is_first_mb == FIRST_ONE_CATABLE_BYTE) ? FIRST_BOTH_CATABLE_BYTES : FIRST_ONE_CATABLE_BYTE;
    }
    catable_header_size = storage_ix >> 3;
    s->next_out_kind = NEXT_OUT_STORAGE;
    s->next_out_off = 0;
    *out_size = catable_header_size;
    delta = unprocessed_input_size(s);
  }
  uint32_t wrapped_last_processed_pos = wrap_position(s->last_processed_pos_);
  if (s->params.quality < 2) {
    /* encode.rs:2335-2389: the fragment compressors on what the ring buffer holds of this block -- the way a CATABLE stream goes
       at qualities 0 / 1 (and with it one that was given a custom dictionary, :1237-1241, and the shards of compress_multi);
       everything else takes compress_stream_fast and never gets here */
    if (delta == 0 && !is_last) {
      *out_size = catable_header_size;
      return 1;
    }
    {
      const uint8_t* data = s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index;
      size_t table_size = 0;
      int32_t* table = get_hash_table(s->params.quality, bytes, &table_size);
      if (s->params.quality == 0) {
        orc_compress_fragment_fast(data + (wrapped_last_processed_pos & mask), bytes, is_last, table, table_size, s->cmd_depths_,
                                   s->cmd_bits_, &s->cmd_code_numbits_, s->cmd_code_, &storage_ix, s->storage_);
      } else {
        uint32_t* command_buf = (uint32_t*)calloc((size_t)1 << 17, sizeof(uint32_t)); /* kCompressFragmentTwoPassBlockSize */
        uint8_t* literal_buf = (uint8_t*)calloc((size_t)1 << 17, 1);
        orc_compress_fragment_two_pass(data + (wrapped_last_processed_pos & mask), bytes, is_last, command_buf, literal_buf, table,
                                       table_size, &storage_ix, s->storage_);
        free(command_buf);
        free(literal_buf);
      }
      free(table);
      s->last_bytes_ = (uint16_t)(s->storage_[storage_ix >> 3] | (s->storage_[1 + (storage_ix >> 3)] << 8));
      s->last_bytes_bits_ = (uint8_t)(storage_ix & 7);
    }
    update_last_processed_pos(s);
    s->next_out_kind = NEXT_OUT_STORAGE;
    s->next_out_off = 0;
    *out_size = storage_ix >> 3;
    return 1;
  }
  {
    size_t newsize = s->num_commands_ + bytes / 2 + 1;
    if (newsize > s->cmd_alloc_size_) {
      newsize += bytes / 4 + 16;
      s->cmd_alloc_size_ = newsize;
      s->commands_ = (Command*)realloc(s->commands_, newsize * sizeof(Command));
    }
  }
  {
    uint8_t* data = s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index;
    /* InitOrStitchToPreviousBlock, encode.rs:1301-1323 */
    orc_hasher_setup(&s->hasher_, &s->params, s->custom_dictionary_size, data, wrapped_last_processed_pos, bytes, is_last);
    orc_hasher_stitch(&s->hasher_, bytes, wrapped_last_processed_pos, data, mask, &s->stats);
  }
  /* ChooseContextMode, encode.rs:1357-1377: UTF8 unless forced (q<10) */
  int literal_context_mode = ORC_CONTEXT_UTF8;
  switch (s->params.mode) {
    case 3: literal_context_mode = ORC_CONTEXT_LSB6; break;
    case 4: literal_context_mode = ORC_CONTEXT_MSB6; break;
    case 5: literal_context_mode = ORC_CONTEXT_UTF8; break;
    case 6: literal_context_mode = ORC_CONTEXT_SIGNED; break;
    default:
      /* the reference hands ChooseContextMode the ring buffer allocation itself (data_mo, encode.rs:2427-2433), not the
         slice behind buffer_index that every other consumer gets: the UTF-8 census runs over bytes shifted by two */
      if (s->params.quality >= 10 &&
          !is_mostly_utf8(s->ringbuffer_.data_mo, wrap_position(s->last_flush_pos_), mask,
                          (size_t)(s->input_pos_ - s->last_flush_pos_), 0.75f))
        literal_context_mode = ORC_CONTEXT_SIGNED;
      break;
  }
  if (s->num_commands_ != 0 && s->last_insert_len_ == 0) extend_last_command(s, &bytes, &wrapped_last_processed_pos);
  orc_create_backward_references(bytes, wrapped_last_processed_pos, s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index,
                                 mask, s->custom_dictionary_size, &s->params, &s->hasher_, s->dist_cache_,
                                 &s->last_insert_len_, &s->commands_[s->num_commands_], &s->num_commands_,
                                 &s->num_literals_, &s->stats);
  {
    size_t max_length = max_metablock_size(&s->params);
    size_t max_literals = max_length / 8;
    size_t max_commands = max_length / 8;
    size_t processed_bytes = (size_t)(s->input_pos_ - s->last_flush_pos_);
    int next_input_fits_metablock = processed_bytes + input_block_size(s) <= max_length;
    int should_flush = s->params.quality < 4 && s->num_literals_ + s->num_commands_ >= 0x2fff;
    if (!is_last && !force_flush && !should_flush && next_input_fits_metablock && s->num_literals_ < max_literals &&
        s->num_commands_ < max_commands) {
      if (update_last_processed_pos(s)) orc_hasher_reset(&s->hasher_);
      *out_size = catable_header_size;
      return 1;
    }
  }
  if (s->last_insert_len_ > 0) {
    orc_command_init_insert(&s->commands_[s->num_commands_++], s->last_insert_len_);
    s->num_literals_ += s->last_insert_len_;
    s->last_insert_len_ = 0;
  }
  if (!is_last && s->input_pos_ == s->last_flush_pos_) {
    *out_size = catable_header_size;
    return 1;
  }
  {
    uint32_t metablock_size = (uint32_t)(s->input_pos_ - s->last_flush_pos_);
    const uint8_t* data = s->ringbuffer_.data_mo + s->ringbuffer_.buffer_index;
    s->stats.metablocks++;
    s->stats.commands += s->num_commands_;
    s->stats.literals += s->num_literals_;
    write_meta_block_internal(s, data, mask, s->last_flush_pos_, metablock_size, is_last, literal_context_mode,
                              &storage_ix, s->storage_);
    s->last_bytes_ = (uint16_t)(s->storage_[storage_ix >> 3] | (s->storage_[1 + (storage_ix >> 3)] << 8));
    s->last_bytes_bits_ = (uint8_t)(storage_ix & 7);
    s->last_flush_pos_ = s->input_pos_;
    if (update_last_processed_pos(s)) orc_hasher_reset(&s->hasher_);
    if (s->last_flush_pos_ > 0) s->prev_byte_ = data[((uint32_t)s->last_flush_pos_ - 1) & mask];
    if (s->last_flush_pos_ > 1) s->prev_byte2_ = data[(uint32_t)(s->last_flush_pos_ - 2) & mask];
    s->num_commands_ = 0;
    s->num_literals_ = 0;
    memcpy(s->saved_dist_cache_, s->dist_cache_, 4 * sizeof(int32_t));
    s->next_out_kind = NEXT_OUT_STORAGE;
    s->next_out_off = 0;
    *out_size = storage_ix >> 3;
    return 1;
  }
}

static uint8_t* get_next_out(OrcEncoder* s) {
  if (s->next_out_kind == NEXT_OUT_STORAGE) return s->storage_ + s->next_out_off;
  if (s->next_out_kind == NEXT_OUT_TINY) return s->tiny_buf_ + s->next_out_off;
  return NULL;
}

/* encode.rs:1541-1566 */
static void inject_byte_padding_block(OrcEncoder* s) {
  uint32_t seal = s->last_bytes_;
  size_t seal_bits = s->last_bytes_bits_;
  uint8_t* destination;
  s->last_bytes_ = 0;
  s->last_bytes_bits_ = 0;
  seal |= 0x6u << seal_bits;
  seal_bits += 6;
  if (s->next_out_kind != NEXT_OUT_NONE) {
    destination = get_next_out(s) + s->available_out_;
  } else {
    destination = s->tiny_buf_;
    s->next_out_kind = NEXT_OUT_TINY;
    s->next_out_off = 0;
  }
  destination[0] = (uint8_t)seal;
  if (seal_bits > 8) destination[1] = (uint8_t)(seal >> 8);
  if (seal_bits > 16) destination[2] = (uint8_t)(seal >> 16);
  s->available_out_ += (seal_bits + 7) >> 3;
}

/* encode.rs:1568-1598 */
static int inject_flush_or_push_output(OrcEncoder* s, size_t* available_out, uint8_t** next_out, size_t* total_out) {
  if (s->stream_state_ == STREAM_FLUSH_REQUESTED && s->last_bytes_bits_ != 0) {
    inject_byte_padding_block(s);
    return 1;
  }
  if (s->available_out_ != 0 && *available_out != 0) {
    size_t copy_output_size = ORC_MIN(s->available_out_, *available_out);
    memcpy(*next_out, get_next_out(s), copy_output_size);
    *next_out += copy_output_size;
    *available_out -= copy_output_size;
    s->next_out_off += copy_output_size;
    s->available_out_ -= copy_output_size;
    s->total_out_ += copy_output_size;
    if (total_out) *total_out = (size_t)s->total_out_;
    return 1;
  }
  return 0;
}

/* encode.rs:1604-1620 */
static void update_size_hint(OrcEncoder* s, size_t available_in) {
  if (s->params.size_hint == 0) {
    uint64_t delta = unprocessed_input_size(s);
    uint64_t tail = available_in;
    uint32_t limit = 1u << 30;
    uint32_t total;
    if (delta >= limit || tail >= limit || delta + tail >= limit) {
      total = limit;
    } else {
      total = (uint32_t)(delta + tail);
    }
    s->params.size_hint = total;
  }
}

static void check_flush_complete(OrcEncoder* s) {
  if (s->stream_state_ == STREAM_FLUSH_REQUESTED && s->available_out_ == 0) {
    s->stream_state_ = STREAM_PROCESSING;
    s->next_out_kind = NEXT_OUT_NONE;
  }
}

/* encode.rs:2545-2575 */
static size_t write_metadata_header(OrcEncoder* s) {
  size_t block_size = s->remaining_metadata_bytes_;
  uint8_t* header = s->tiny_buf_;
  size_t storage_ix = s->last_bytes_bits_;
  memset(header, 0, sizeof(s->tiny_buf_));
  header[0] = (uint8_t)s->last_bytes_;
  header[1] = (uint8_t)(s->last_bytes_ >> 8);
  s->last_bytes_ = 0;
  s->last_bytes_bits_ = 0;
  orc_write_bits(1, 0, &storage_ix, header);
  orc_write_bits(2, 3, &storage_ix, header);
  orc_write_bits(1, 0, &storage_ix, header);
  if (block_size == 0) {
    orc_write_bits(2, 0, &storage_ix, header);
  } else {
    uint32_t nbits = block_size == 1 ? 0 : orc_log2_floor_nonzero((uint32_t)block_size - 1) + 1;
    uint32_t nbytes = (nbits + 7) / 8;
    orc_write_bits(2, nbytes, &storage_ix, header);
    orc_write_bits(8 * nbytes, block_size - 1, &storage_ix, header);
  }
  return (storage_ix + 7) >> 3;
}

/* encode.rs:2579-2685 */
static int process_metadata(OrcEncoder* s, size_t* available_in, const uint8_t** next_in, size_t* available_out,
                            uint8_t** next_out, size_t* total_out) {
  if (*available_in > (1u << 24)) return 0;
  if (s->stream_state_ == STREAM_PROCESSING) {
    s->remaining_metadata_bytes_ = (uint32_t)*available_in;
    s->stream_state_ = STREAM_METADATA_HEAD;
  }
  if (s->stream_state_ != STREAM_METADATA_HEAD && s->stream_state_ != STREAM_METADATA_BODY) return 0;
  for (;;) {
    if (inject_flush_or_push_output(s, available_out, next_out, total_out)) continue;
    if (s->available_out_ != 0) break;
    if (s->input_pos_ != s->last_flush_pos_) {
      size_t avail_out = s->available_out_;
      int result = encode_data(s, 0, 1, &avail_out);
      s->available_out_ = avail_out;
      if (!result) return 0;
      continue;
    }
    if (s->stream_state_ == STREAM_METADATA_HEAD) {
      s->next_out_kind = NEXT_OUT_TINY;
      s->next_out_off = 0;
      s->available_out_ = write_metadata_header(s);
      s->stream_state_ = STREAM_METADATA_BODY;
      continue;
    } else {
      if (s->remaining_metadata_bytes_ == 0) {
        s->remaining_metadata_bytes_ = 0xffffffffu;
        s->stream_state_ = STREAM_PROCESSING;
        break;
      }
      if (*available_out != 0) {
        uint32_t copy = (uint32_t)ORC_MIN((size_t)s->remaining_metadata_bytes_, *available_out);
        memcpy(*next_out, *next_in, copy);
        *next_in += copy;
        *available_in -= copy;
        s->remaining_metadata_bytes_ -= copy;
        *next_out += copy;
        *available_out -= copy;
      } else {
        uint32_t copy = ORC_MIN(s->remaining_metadata_bytes_, 16u);
        s->next_out_kind = NEXT_OUT_TINY;
        s->next_out_off = 0;
        memcpy(s->tiny_buf_, *next_in, copy);
        *next_in += copy;
        *available_in -= copy;
        s->remaining_metadata_bytes_ -= copy;
        s->available_out_ = copy;
      }
      continue;
    }
  }
  return 1;
}

/* encode.rs:1643-1700 (MaxHashTableSize, HashTableSize, GetHashTableInternal): a zeroed table per call */
static int32_t* get_hash_table(int quality, size_t input_size, size_t* table_size) {
  size_t max_table_size = quality == 0 ? ((size_t)1 << 15) : ((size_t)1 << 17);
  size_t htsize = 256;
  while (htsize < max_table_size && htsize < input_size) htsize <<= 1;
  if (quality == 0 && (htsize & 0xaaaaa) == 0) htsize <<= 1;
  *table_size = htsize;
  return (int32_t*)calloc(htsize, sizeof(int32_t));
}

/* encode.rs:2706-2861 (qualities 0 and 1: no ring buffer, the caller's input is compressed in place) */
static int compress_stream_fast(OrcEncoder* s, int op, size_t* available_in, const uint8_t** next_in,
                                size_t* available_out, uint8_t** next_out, size_t* total_out) {
  const size_t block_size_limit = (size_t)1 << s->params.lgwin;
  const size_t buf_size = ORC_MIN((size_t)1 << 17, ORC_MIN(*available_in, block_size_limit));
  uint32_t* command_buf = NULL;
  uint8_t* literal_buf = NULL;
  if (s->params.quality != 0 && s->params.quality != 1) return 0;
  command_buf = (uint32_t*)calloc(buf_size ? buf_size : 1, sizeof(uint32_t));
  literal_buf = (uint8_t*)calloc(buf_size ? buf_size : 1, 1);
  for (;;) {
    if (inject_flush_or_push_output(s, available_out, next_out, total_out)) continue;
    if (s->available_out_ == 0 && s->stream_state_ == STREAM_PROCESSING && (*available_in != 0 || op != ORC_OP_PROCESS)) {
      const size_t block_size = ORC_MIN(block_size_limit, *available_in);
      const int is_last = *available_in == block_size && op == ORC_OP_FINISH;
      const int force_flush = *available_in == block_size && op == ORC_OP_FLUSH;
      const size_t max_out_size = 2 * block_size + 503;
      int inplace = 1;
      uint8_t* storage;
      size_t storage_ix = s->last_bytes_bits_;
      size_t table_size = 0;
      if (force_flush && block_size == 0) {
        s->stream_state_ = STREAM_FLUSH_REQUESTED;
        continue;
      }
      if (max_out_size <= *available_out) {
        storage = *next_out;
      } else {
        inplace = 0;
        get_brotli_storage(s, max_out_size);
        storage = s->storage_;
      }
      storage[0] = (uint8_t)s->last_bytes_;
      storage[1] = (uint8_t)(s->last_bytes_ >> 8);
      int32_t* table = get_hash_table(s->params.quality, block_size, &table_size);
      if (s->params.quality == 0) {
        orc_compress_fragment_fast(*next_in, block_size, is_last, table, table_size, s->cmd_depths_, s->cmd_bits_,
                                   &s->cmd_code_numbits_, s->cmd_code_, &storage_ix, storage);
      } else {
        orc_compress_fragment_two_pass(*next_in, block_size, is_last, command_buf, literal_buf, table, table_size,
                                       &storage_ix, storage);
      }
      free(table);
      *next_in += block_size;
      *available_in -= block_size;
      if (inplace) {
        size_t out_bytes = storage_ix >> 3;
        *next_out += out_bytes;
        *available_out -= out_bytes;
        s->total_out_ += out_bytes;
        if (total_out) *total_out = (size_t)s->total_out_;
      } else {
        s->next_out_kind = NEXT_OUT_STORAGE;
        s->next_out_off = 0;
        s->available_out_ = storage_ix >> 3;
      }
      s->last_bytes_ = (uint16_t)(storage[storage_ix >> 3] | (storage[1 + (storage_ix >> 3)] << 8));
      s->last_bytes_bits_ = (uint8_t)(storage_ix & 7);
      if (force_flush) s->stream_state_ = STREAM_FLUSH_REQUESTED;
      if (is_last) s->stream_state_ = STREAM_FINISHED;
      continue;
    }
    break;
  }
  free(command_buf);
  free(literal_buf);
  check_flush_complete(s);
  return 1;
}

/* encode.rs:2873-2995 */
int orc_encoder_compress_stream(OrcEncoder* s, int op, size_t* available_in, const uint8_t** next_in,
                                size_t* available_out, uint8_t** next_out, size_t* total_out) {
  if (!ensure_initialized(s)) return 0;
  if (s->remaining_metadata_bytes_ != 0xffffffffu) {
    if (*available_in != (size_t)s->remaining_metadata_bytes_) return 0;
    if (op != ORC_OP_EMIT_METADATA) return 0;
  }
  if (op == ORC_OP_EMIT_METADATA) {
    update_size_hint(s, 0);
    return process_metadata(s, available_in, next_in, available_out, next_out, total_out);
  }
  if (s->stream_state_ == STREAM_METADATA_HEAD || s->stream_state_ == STREAM_METADATA_BODY) return 0;
  if (s->stream_state_ != STREAM_PROCESSING && *available_in != 0) return 0;
  if ((s->params.quality == 0 || s->params.quality == 1) && !s->params.catable)
    return compress_stream_fast(s, op, available_in, next_in, available_out, next_out, total_out);
  for (;;) {
    size_t remaining_block_size;
    {
      uint64_t delta = unprocessed_input_size(s);
      size_t block_size = input_block_size(s);
      remaining_block_size = delta >= block_size ? 0 : (size_t)(block_size - delta);
    }
    if (remaining_block_size != 0 && *available_in != 0) {
      size_t copy_input_size = ORC_MIN(remaining_block_size, *available_in);
      copy_input_to_ring_buffer(s, copy_input_size, *next_in);
      *next_in += copy_input_size;
      *available_in -= copy_input_size;
      continue;
    }
    if (inject_flush_or_push_output(s, available_out, next_out, total_out)) continue;
    if (s->available_out_ == 0 && s->stream_state_ == STREAM_PROCESSING &&
        (remaining_block_size == 0 || op != ORC_OP_PROCESS)) {
      int is_last = (*available_in == 0) && op == ORC_OP_FINISH;
      int force_flush = (*available_in == 0) && op == ORC_OP_FLUSH;
      update_size_hint(s, *available_in);
      size_t avail_out = s->available_out_;
      int result = encode_data(s, is_last, force_flush, &avail_out);
      s->available_out_ = avail_out;
      if (!result) return 0;
      if (force_flush) s->stream_state_ = STREAM_FLUSH_REQUESTED;
      if (is_last) s->stream_state_ = STREAM_FINISHED;
      continue;
    }
    break;
  }
  check_flush_complete(s);
  return 1;
}

int orc_encoder_is_finished(const OrcEncoder* s) { return s->stream_state_ == STREAM_FINISHED && s->available_out_ == 0; }
int orc_encoder_has_more_output(const OrcEncoder* s) { return s->available_out_ != 0; }

/* encode.rs:3004-3030 */
const uint8_t* orc_encoder_take_output(OrcEncoder* s, size_t* size) {
  size_t consumed_size = s->available_out_;
  uint8_t* result = get_next_out(s);
  if (*size != 0) consumed_size = ORC_MIN(*size, s->available_out_);
  if (consumed_size != 0) {
    s->next_out_off += consumed_size;
    s->available_out_ -= consumed_size;
    s->total_out_ += consumed_size;
    check_flush_complete(s);
    *size = consumed_size;
  } else {
    *size = 0;
    result = NULL;
  }
  return result;
}

/* encode.rs:1388-1433 */
static size_t make_uncompressed_stream(const uint8_t* input, size_t input_size, uint8_t* output) {
  size_t size = input_size, result = 0, offset = 0;
  if (input_size == 0) {
    output[0] = 6;
    return 1;
  }
  output[result++] = 0x21;
  output[result++] = 0x03;
  while (size > 0) {
    uint32_t nibbles = 0;
    uint32_t chunk_size = size > (1u << 24) ? (1u << 24) : (uint32_t)size;
    if (chunk_size > (1u << 16)) nibbles = chunk_size > (1u << 20) ? 2 : 1;
    uint32_t bits = (nibbles << 1) | ((chunk_size - 1) << 3) | (1u << (19 + 4 * nibbles));
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

/* encode.rs:1436-1538 */
int orc_encoder_compress(int quality, int lgwin, int mode, size_t input_size, const uint8_t* input,
                         size_t* encoded_size, uint8_t* encoded, OrcStats* stats_out) {
  size_t out_size = *encoded_size;
  size_t max_out_size = orc_max_compressed_size(input_size);
  if (out_size == 0) return 0;
  if (input_size == 0) {
    *encoded_size = 1;
    encoded[0] = 6;
    return 1;
  }
  /* encode.rs:1468-1481: quality 10 through the one-shot entry runs the encoder at quality 9 with a hasher made ahead of
     time from {q9_5, quality 10} -- ChooseHasher gives that H9 with the parameters quality 9 would get anyway, and a
     fresh hasher is zeroed (:1147), so the call is a quality-9 call */
  if (quality == 10) quality = 9;
  {
    OrcEncoder* s = orc_encoder_create();
    size_t available_in = input_size;
    const uint8_t* next_in = input;
    size_t available_out = *encoded_size;
    uint8_t* next_out = encoded;
    size_t total_out = 0;
    orc_encoder_set_parameter(s, ORC_PARAM_QUALITY, (uint32_t)quality);
    orc_encoder_set_parameter(s, ORC_PARAM_LGWIN, (uint32_t)lgwin);
    orc_encoder_set_parameter(s, ORC_PARAM_MODE, (uint32_t)mode);
    orc_encoder_set_parameter(s, ORC_PARAM_SIZE_HINT, (uint32_t)input_size);
    if (lgwin > 24) orc_encoder_set_parameter(s, ORC_PARAM_LARGE_WINDOW, 1);
    int result = orc_encoder_compress_stream(s, ORC_OP_FINISH, &available_in, &next_in, &available_out, &next_out,
                                             &total_out);
    if (!orc_encoder_is_finished(s)) result = 0;
    *encoded_size = total_out;
    if (stats_out) *stats_out = s->stats;
    orc_encoder_destroy(s);
    if (result && !(max_out_size != 0 && *encoded_size > max_out_size)) return 1;
  }
  *encoded_size = 0;
  if (max_out_size == 0) return 0;
  if (out_size >= max_out_size) {
    *encoded_size = make_uncompressed_stream(input, input_size, encoded);
    return 1;
  }
  return 0;
}

/* CompressorWriter feeding pattern: src/enc/writer.rs:183-313 */
int orc_writer_compress(int quality, int lgwin, size_t chunk, size_t input_size, const uint8_t* input,
                        size_t* encoded_size, uint8_t* encoded, OrcStats* stats_out, OrcMetablockTrace cb,
                        void* opaque) {
  OrcEncoder* s = orc_encoder_create();
  size_t available_out = *encoded_size;
  uint8_t* next_out = encoded;
  size_t total_out = 0;
  size_t off = 0;
  int ok = 1;
  orc_encoder_set_parameter(s, ORC_PARAM_QUALITY, (uint32_t)quality);
  orc_encoder_set_parameter(s, ORC_PARAM_LGWIN, (uint32_t)lgwin);
  orc_encoder_set_trace(s, cb, opaque);
  if (chunk == 0) chunk = input_size ? input_size : 1;
  while (ok && off < input_size) {
    size_t available_in = ORC_MIN(chunk, input_size - off);
    const uint8_t* next_in = input + off;
    size_t before = available_in;
    while (ok && available_in != 0) {
      ok = orc_encoder_compress_stream(s, ORC_OP_PROCESS, &available_in, &next_in, &available_out, &next_out, &total_out);
      if (available_out == 0 && available_in != 0) ok = 0;
    }
    off += before;
  }
  while (ok && !orc_encoder_is_finished(s)) {
    size_t available_in = 0;
    const uint8_t* next_in = input + input_size;
    ok = orc_encoder_compress_stream(s, ORC_OP_FINISH, &available_in, &next_in, &available_out, &next_out, &total_out);
    if (ok && !orc_encoder_is_finished(s) && available_out == 0) ok = 0;
  }
  *encoded_size = total_out;
  if (stats_out) *stats_out = s->stats;
  orc_encoder_destroy(s);
  return ok;
}

/* BrotliCompressCustomIoCustomDict feeding pattern (src/enc/mod.rs:225-345; what `brotli::BrotliCompress` and the
   reference's integration tests use): the parameter struct is copied in, the input is read in `chunk`-byte pieces, each
   handed over with PROCESS, and FINISH follows with no input once the reader is dry. */
int orc_reader_compress(const int* param_keys, const uint32_t* param_values, size_t num_params, size_t chunk,
                        size_t input_size, const uint8_t* input, size_t* encoded_size, uint8_t* encoded,
                        OrcStats* stats_out, OrcMetablockTrace cb, void* opaque) {
  OrcEncoder* s = orc_encoder_create();
  size_t available_out = *encoded_size;
  uint8_t* next_out = encoded;
  size_t total_out = 0;
  size_t off = 0;
  int ok = 1;
  for (size_t i = 0; i < num_params; ++i) orc_encoder_set_parameter(s, param_keys[i], param_values[i]);
  orc_encoder_set_trace(s, cb, opaque);
  if (chunk == 0) chunk = input_size ? input_size : 1;
  for (;;) {
    size_t available_in = ORC_MIN(chunk, input_size - off);
    const uint8_t* next_in = input + off;
    int op = available_in == 0 ? ORC_OP_FINISH : ORC_OP_PROCESS;
    off += available_in;
    do {
      ok = orc_encoder_compress_stream(s, op, &available_in, &next_in, &available_out, &next_out, &total_out);
      if (ok && available_out == 0 && (available_in != 0 || orc_encoder_has_more_output(s))) ok = 0; /* buffer too small */
    } while (ok && (available_in != 0 || (op == ORC_OP_FINISH && !orc_encoder_is_finished(s))));
    if (!ok || orc_encoder_is_finished(s)) break;
  }
  *encoded_size = total_out;
  if (stats_out) *stats_out = s->stats;
  orc_encoder_destroy(s);
  return ok;
}
