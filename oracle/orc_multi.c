/* oracle/orc_multi.c -- CPU restatement of rust-brotli's multi-chunk path.
 * TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows src/enc/threading/mod.rs:333-411 (get_range, compress_part), :565-660 (join + stitch) and
 * src/concat/mod.rs:39-123 (header parsing), :274-608 (BroCatli::stream / finish).  Shards are
 * compressed one after the other on the calling thread: the result does not depend on scheduling.
 * favor_cpu_efficiency (shared pre-built hasher) is not restated.
 */
#include "orc_internal.h"

enum {
  CAT_SUCCESS = 0,
  CAT_NEEDS_MORE_INPUT = 1,
  CAT_NEEDS_MORE_OUTPUT = 2,
  CAT_NOT_CRAFTED_FOR_APPEND = 124,
  CAT_INVALID_WINDOW_SIZE = 125,
  CAT_WINDOW_SIZE_LARGER = 126,
  CAT_NOT_CRAFTED_FOR_CONCAT = 127
};

#define NUM_STREAM_HEADER_BYTES 5

typedef struct {
  uint8_t bytes_so_far[NUM_STREAM_HEADER_BYTES];
  uint8_t num_bytes_read;
  int has_written; /* Option<u8> */
  uint8_t num_bytes_written;
} NewStreamData;

typedef struct {
  uint8_t last_bytes[2];
  uint8_t last_bytes_len;
  int last_byte_sanitized;
  int any_bytes_emitted;
  uint8_t last_byte_bit_offset;
  uint8_t window_size;
  int has_pending;
  NewStreamData pending;
} BroCatli;

static int nsd_sufficient(const NewStreamData* d) {
  if (d->num_bytes_read == 4 && (127 & d->bytes_so_far[0]) != 17) return 1;
  return d->num_bytes_read == 5;
}

/* concat/mod.rs:39-74; returns 0 on error */
static int parse_window_size(const uint8_t* b, uint8_t* window_size, size_t* offset) {
  if ((b[0] & 1) == 0) {
    *window_size = 16;
    *offset = 1;
    return 1;
  }
  switch (b[0] & 15) {
    case 0x3: *window_size = 18; *offset = 4; return 1;
    case 0x5: *window_size = 19; *offset = 4; return 1;
    case 0x7: *window_size = 20; *offset = 4; return 1;
    case 0x9: *window_size = 21; *offset = 4; return 1;
    case 0xb: *window_size = 22; *offset = 4; return 1;
    case 0xd: *window_size = 23; *offset = 4; return 1;
    case 0xf: *window_size = 24; *offset = 4; return 1;
    default:
      switch (b[0] & 127) {
        case 0x71: *window_size = 15; *offset = 7; return 1;
        case 0x61: *window_size = 14; *offset = 7; return 1;
        case 0x51: *window_size = 13; *offset = 7; return 1;
        case 0x41: *window_size = 12; *offset = 7; return 1;
        case 0x31: *window_size = 11; *offset = 7; return 1;
        case 0x21: *window_size = 10; *offset = 7; return 1;
        case 0x1: *window_size = 17; *offset = 7; return 1;
        default: break;
      }
  }
  if (b[0] & 0x80) return 0;
  uint8_t ret = b[1] & 0x3f;
  if (ret < 10 || ret > 30) return 0;
  *window_size = ret;
  *offset = 14;
  return 1;
}

/* concat/mod.rs:76-123 */
static int detect_varlen_offset(const uint8_t* b, size_t n, size_t* out) {
  uint8_t ws;
  size_t offset;
  if (!parse_window_size(b, &ws, &offset)) return 0;
  uint64_t bytes = 0;
  for (size_t i = 0; i < n; ++i) bytes |= (uint64_t)b[i] << (i * 8);
  bytes >>= offset;
  offset += 1;
  if (bytes & 1) { /* ISLAST */
    bytes >>= 1;
    offset += 1;
    if (bytes & 1) { /* ISLASTEMPTY */
      *out = offset;
      return 1;
    }
  }
  bytes >>= 1;
  uint64_t mnibbles = bytes & 3;
  bytes >>= 2;
  offset += 2;
  if (mnibbles == 3) {
    if (bytes & 1) return 0;
    bytes >>= 1;
    offset += 1;
    uint64_t mskipbytes = bytes & 3;
    offset += 2;
    offset += (size_t)mskipbytes * 8;
    *out = offset;
    return 1;
  }
  mnibbles += 4;
  offset += (size_t)mnibbles * 4;
  bytes >>= mnibbles * 4;
  offset += 1;
  if ((bytes & 1) == 0) return 0;
  *out = offset;
  return 1;
}

/* concat/mod.rs:277-330 */
static int flush_previous_stream(BroCatli* c, uint8_t* out, size_t out_len, size_t* out_offset) {
  if (!c->last_byte_sanitized) {
    if (c->last_bytes_len == 0) {
      c->last_byte_sanitized = 1;
      return CAT_SUCCESS;
    }
    uint16_t last_bytes = (uint16_t)(c->last_bytes[0] + (c->last_bytes[1] << 8));
    uint8_t max = (uint8_t)(c->last_bytes_len * 8);
    uint8_t index = (uint8_t)(max - 1);
    for (uint8_t i = 0; i < max; ++i) {
      index = (uint8_t)(max - 1 - i);
      if ((1u << index) & last_bytes) break;
    }
    if (index == 0) return CAT_NOT_CRAFTED_FOR_APPEND;
    if ((last_bytes >> (index - 1)) != 3) return CAT_NOT_CRAFTED_FOR_APPEND;
    index -= 1;
    last_bytes &= (uint16_t)((1u << index) - 1);
    c->last_bytes[0] = (uint8_t)last_bytes;
    c->last_bytes[1] = (uint8_t)(last_bytes >> 8);
    if (index >= 8) {
      if (out_len > *out_offset) {
        out[*out_offset] = c->last_bytes[0];
        c->last_bytes[0] = c->last_bytes[1];
        *out_offset += 1;
        c->any_bytes_emitted = 1;
        index -= 8;
        c->last_bytes_len -= 1;
      } else {
        return CAT_NEEDS_MORE_OUTPUT;
      }
    }
    c->last_byte_bit_offset = index;
    c->last_byte_sanitized = 1;
  }
  return CAT_SUCCESS;
}

/* concat/mod.rs:332-449 */
static int shift_and_check_new_stream_header(BroCatli* c, NewStreamData nsp, uint8_t* out, size_t out_len,
                                             size_t* out_offset) {
  if (!nsp.has_written) {
    uint8_t window_size;
    size_t window_offset;
    if (!parse_window_size(nsp.bytes_so_far, &window_size, &window_offset)) return CAT_INVALID_WINDOW_SIZE;
    if (c->window_size == 0) {
      c->window_size = window_size;
      out[*out_offset] = nsp.bytes_so_far[0];
      nsp.has_written = 1;
      nsp.num_bytes_written = 1;
      c->any_bytes_emitted = 1;
      *out_offset += 1;
    } else {
      if (window_size > c->window_size) return CAT_WINDOW_SIZE_LARGER;
      uint8_t realigned_header[NUM_STREAM_HEADER_BYTES + 1] = {c->last_bytes[0], 0, 0, 0, 0, 0};
      size_t varlen_offset;
      if (!detect_varlen_offset(nsp.bytes_so_far, nsp.num_bytes_read, &varlen_offset)) return CAT_NOT_CRAFTED_FOR_CONCAT;
      uint64_t bytes_so_far = 0;
      for (size_t i = 0; i < nsp.num_bytes_read; ++i) bytes_so_far |= (uint64_t)nsp.bytes_so_far[i] << (i * 8);
      bytes_so_far >>= window_offset;
      bytes_so_far &= (1ull << (varlen_offset - window_offset)) - 1;
      size_t var_len_bytes = ((varlen_offset - window_offset) + 7) / 8;
      for (size_t byte_index = 0; byte_index < var_len_bytes; ++byte_index) {
        uint64_t cur_byte = bytes_so_far >> (byte_index * 8);
        realigned_header[byte_index] |=
            (uint8_t)((cur_byte & ((1u << (8 - c->last_byte_bit_offset)) - 1)) << c->last_byte_bit_offset);
        realigned_header[byte_index + 1] = (uint8_t)(cur_byte >> (8 - c->last_byte_bit_offset));
      }
      size_t whole_byte_destination = ((size_t)c->last_byte_bit_offset + varlen_offset - window_offset + 7) / 8;
      size_t whole_byte_source = (varlen_offset + 7) / 8;
      if (whole_byte_source > nsp.num_bytes_read) return CAT_NOT_CRAFTED_FOR_CONCAT;
      size_t num_whole_bytes_to_copy = nsp.num_bytes_read - whole_byte_source;
      for (size_t i = 0; i < num_whole_bytes_to_copy; ++i)
        realigned_header[whole_byte_destination + i] = nsp.bytes_so_far[whole_byte_source + i];
      out[*out_offset] = realigned_header[0];
      c->any_bytes_emitted = 1;
      *out_offset += 1;
      nsp.num_bytes_read = (uint8_t)(whole_byte_destination + num_whole_bytes_to_copy - 1);
      nsp.has_written = 1;
      nsp.num_bytes_written = 0;
      memcpy(nsp.bytes_so_far, &realigned_header[1], NUM_STREAM_HEADER_BYTES);
    }
  }
  size_t to_copy = ORC_MIN(out_len - *out_offset, (size_t)(nsp.num_bytes_read - nsp.num_bytes_written));
  memcpy(out + *out_offset, nsp.bytes_so_far + nsp.num_bytes_written, to_copy);
  *out_offset += to_copy;
  if (to_copy != 0) c->any_bytes_emitted = 1;
  nsp.num_bytes_written = (uint8_t)(nsp.num_bytes_written + to_copy);
  if (nsp.num_bytes_written != nsp.num_bytes_read) {
    c->pending = nsp;
    c->has_pending = 1;
    return CAT_NEEDS_MORE_OUTPUT;
  }
  c->has_pending = 0;
  c->last_byte_sanitized = 0;
  c->last_byte_bit_offset = 0;
  c->last_bytes_len = 0;
  c->last_bytes[0] = c->last_bytes[1] = 0;
  *out_offset -= 1;
  c->last_bytes[0] = out[*out_offset];
  c->last_bytes_len = 1;
  return CAT_SUCCESS;
}

/* concat/mod.rs:450-566 */
static int brocatli_stream(BroCatli* c, const uint8_t* in, size_t in_len, size_t* in_offset, uint8_t* out,
                           size_t out_len, size_t* out_offset) {
  if (c->has_pending) {
    NewStreamData nsp = c->pending;
    int flush_result = flush_previous_stream(c, out, out_len, out_offset);
    if (flush_result != CAT_SUCCESS) return flush_result;
    if (nsp.num_bytes_read < NUM_STREAM_HEADER_BYTES) {
      size_t room = NUM_STREAM_HEADER_BYTES - nsp.num_bytes_read;
      size_t to_copy = ORC_MIN(room, in_len - *in_offset);
      memcpy(nsp.bytes_so_far + nsp.num_bytes_read, in + *in_offset, to_copy);
      *in_offset += to_copy;
      nsp.num_bytes_read = (uint8_t)(nsp.num_bytes_read + to_copy);
      c->pending = nsp;
    }
    if (!nsd_sufficient(&nsp)) return CAT_NEEDS_MORE_INPUT;
    if (out_len == *out_offset) return CAT_NEEDS_MORE_OUTPUT;
    int shift_result = shift_and_check_new_stream_header(c, nsp, out, out_len, out_offset);
    if (shift_result != CAT_SUCCESS) return shift_result;
    if (*out_offset == out_len) return CAT_NEEDS_MORE_OUTPUT;
  }
  if (c->last_bytes_len != 2) {
    if (out_len == *out_offset) return CAT_NEEDS_MORE_OUTPUT;
    if (in_len == *in_offset) return CAT_NEEDS_MORE_INPUT;
    c->last_bytes[c->last_bytes_len++] = in[(*in_offset)++];
    if (c->last_bytes_len != 2) {
      if (out_len == *out_offset) return CAT_NEEDS_MORE_OUTPUT;
      if (in_len == *in_offset) return CAT_NEEDS_MORE_INPUT;
      c->last_bytes[c->last_bytes_len++] = in[(*in_offset)++];
    }
  }
  if (out_len == *out_offset) return CAT_NEEDS_MORE_OUTPUT;
  if (in_len == *in_offset) return CAT_NEEDS_MORE_INPUT;
  size_t to_copy = ORC_MIN(out_len - *out_offset, in_len - *in_offset);
  if (to_copy == 1) {
    out[*out_offset] = c->last_bytes[0];
    c->last_bytes[0] = c->last_bytes[1];
    c->last_bytes[1] = in[*in_offset];
    *in_offset += 1;
    *out_offset += 1;
    if (*out_offset == out_len) return CAT_NEEDS_MORE_OUTPUT;
    return CAT_NEEDS_MORE_INPUT;
  }
  out[*out_offset] = c->last_bytes[0];
  out[*out_offset + 1] = c->last_bytes[1];
  *out_offset += 2;
  c->last_bytes[0] = in[*in_offset + to_copy - 2];
  c->last_bytes[1] = in[*in_offset + to_copy - 1];
  memcpy(out + *out_offset, in + *in_offset, to_copy - 2);
  *out_offset += to_copy - 2;
  *in_offset += to_copy;
  if (*out_offset == out_len) return CAT_NEEDS_MORE_OUTPUT;
  return CAT_NEEDS_MORE_INPUT;
}

/* concat/mod.rs:567-608 */
static int brocatli_finish(BroCatli* c, uint8_t* out, size_t out_len, size_t* out_offset) {
  if (c->last_byte_sanitized && c->last_bytes_len != 0) {
    uint16_t last_bytes = (uint16_t)(c->last_bytes[0] | (c->last_bytes[1] << 8));
    uint8_t bit_end = (uint8_t)((c->last_bytes_len - 1) * 8 + c->last_byte_bit_offset);
    last_bytes |= (uint16_t)(3u << bit_end);
    c->last_bytes[0] = (uint8_t)last_bytes;
    c->last_bytes[1] = (uint8_t)(last_bytes >> 8);
    c->last_byte_sanitized = 0;
    c->last_byte_bit_offset += 2;
    if (c->last_byte_bit_offset >= 8) {
      c->last_byte_bit_offset -= 8;
      c->last_bytes_len += 1;
    }
  }
  while (c->last_bytes_len != 0) {
    if (*out_offset == out_len) return CAT_NEEDS_MORE_OUTPUT;
    out[(*out_offset)++] = c->last_bytes[0];
    c->last_bytes_len -= 1;
    c->last_bytes[0] = c->last_bytes[1];
    c->any_bytes_emitted = 1;
  }
  if (!c->any_bytes_emitted) {
    if (out_len == *out_offset) return CAT_NEEDS_MORE_OUTPUT;
    c->any_bytes_emitted = 1;
    out[(*out_offset)++] = ';';
  }
  return CAT_SUCCESS;
}

/* threading/mod.rs:337-411 (compress_part) for one shard; returns size or (size_t)-1 */
static size_t compress_part(const int* keys, const uint32_t* vals, size_t nparams, size_t thread_index,
                            size_t num_threads, const uint8_t* input, size_t input_size, uint8_t** mem_out) {
  size_t start = (thread_index * input_size) / num_threads;
  size_t end = ((thread_index + 1) * input_size) / num_threads;
  size_t cap = orc_max_compressed_size(end - start);
  uint8_t* mem = (uint8_t*)malloc(cap ? cap : 1);
  OrcEncoder* s = orc_encoder_create();
  for (size_t i = 0; i < nparams; ++i) orc_encoder_set_parameter(s, keys[i], vals[i]);
  if (thread_index != 0) {
    /* direct field writes in the reference (state.params.catable = true; magic_number = false) */
    orc_encoder_set_parameter(s, ORC_PARAM_CATABLE, 1);
    /* set_parameter(CATABLE) also flips appendable/use_dictionary; the reference writes the field only,
       then SanitizeParams (catable => appendable, !use_dictionary) yields the same state. */
    orc_encoder_set_parameter(s, ORC_PARAM_MAGIC_NUMBER, 0);
  }
  orc_encoder_set_parameter(s, ORC_PARAM_APPENDABLE, 1);
  if (thread_index != 0) orc_encoder_set_custom_dictionary(s, start, input, 1);
  size_t available_in = end - start;
  const uint8_t* next_in = input + start;
  size_t available_out = cap;
  uint8_t* next_out = mem;
  size_t total = 0;
  int ok = orc_encoder_compress_stream(s, ORC_OP_FINISH, &available_in, &next_in, &available_out, &next_out, &total);
  if (ok && !orc_encoder_is_finished(s)) ok = 0;
  orc_encoder_destroy(s);
  if (!ok) {
    free(mem);
    return (size_t)-1;
  }
  *mem_out = mem;
  return cap - available_out;
}

int orc_compress_multi(const int* keys, const uint32_t* vals, size_t nparams, size_t input_size,
                       const uint8_t* input, size_t* encoded_size, uint8_t* encoded, size_t num_threads) {
  BroCatli cat;
  size_t out_file_size = 0;
  size_t out_cap = *encoded_size;
  memset(&cat, 0, sizeof(cat));
  if (num_threads == 0) num_threads = 1;
  if (num_threads > 16) num_threads = 16; /* ffi/multicompress/mod.rs:26,103 */
  for (size_t t = 0; t < num_threads; ++t) {
    uint8_t* mem = NULL;
    size_t n = compress_part(keys, vals, nparams, t, num_threads, input, input_size, &mem);
    if (n == (size_t)-1) return 0;
    memset(&cat.pending, 0, sizeof(cat.pending)); /* new_brotli_file() */
    cat.has_pending = 1;
    size_t in_offset = 0;
    int r = brocatli_stream(&cat, mem, n, &in_offset, encoded, out_cap, &out_file_size);
    free(mem);
    if (r != CAT_SUCCESS && r != CAT_NEEDS_MORE_INPUT) return 0;
  }
  if (brocatli_finish(&cat, encoded, out_cap, &out_file_size) != CAT_SUCCESS) return 0;
  *encoded_size = out_file_size;
  return 1;
}

/* BroCatli::new_with_window_size, concat/mod.rs:232-271; returns 0 or CAT_INVALID_WINDOW_SIZE */
static int brocatli_init_with_window(BroCatli* c, uint8_t lg) {
  memset(c, 0, sizeof(*c));
  if (lg > 24) {
    c->last_bytes[0] = 17;
    c->last_bytes[1] = (uint8_t)(lg | 64 | 128);
    c->last_bytes_len = 2;
  } else if (lg == 16) {
    c->last_bytes[0] = 1 | 2 | 4;
    c->last_bytes_len = 1;
  } else if (lg > 17) {
    c->last_bytes[0] = (uint8_t)((3 + (lg - 18) * 2) | (16 | 32));
    c->last_bytes_len = 1;
  } else {
    switch (lg) {
      case 15: c->last_bytes[0] = 0x71 | 0x80; break;
      case 14: c->last_bytes[0] = 0x61 | 0x80; break;
      case 13: c->last_bytes[0] = 0x51 | 0x80; break;
      case 12: c->last_bytes[0] = 0x41 | 0x80; break;
      case 11: c->last_bytes[0] = 0x31 | 0x80; break;
      case 10: c->last_bytes[0] = 0x21 | 0x80; break;
      case 17: c->last_bytes[0] = 0x1 | 0x80; break;
      default: return CAT_INVALID_WINDOW_SIZE;
    }
    c->last_bytes[1] = 1;
    c->last_bytes_len = 2;
  }
  c->window_size = lg;
  return CAT_SUCCESS;
}

/* Test entry: concatenates `nfiles` brotli files the way src/bin/test_broccoli.rs:28-132 (`concat`) drives BroCatli -- reads
 * of `bs` bytes per file, an output buffer of `bs` bytes flushed on NeedsMoreOutput.  window < 0: BroCatli::new().
 * Returns CAT_SUCCESS (0) or the failure code (>= 124); *out_size = bytes written (<= out_cap). */
int orc_concat(size_t nfiles, const uint8_t* const* files, const size_t* sizes, int window, size_t bs, uint8_t* out, size_t out_cap,
               size_t* out_size) {
  BroCatli cat;
  uint8_t* obuf = (uint8_t*)malloc(bs ? bs : 1);
  size_t ooffset = 0, written = 0;
  int result = CAT_SUCCESS;
  memset(&cat, 0, sizeof(cat));
  if (window >= 0) {
    /* (broccoli.rs:60-65: an invalid size falls back to the plain constructor) */
    if (brocatli_init_with_window(&cat, (uint8_t)window) != CAT_SUCCESS) memset(&cat, 0, sizeof(cat));
  }
#define ORC_CONCAT_WRITE()                                        \
  do {                                                            \
    if (written + ooffset > out_cap) { result = -1; goto done; }  \
    memcpy(out + written, obuf, ooffset);                         \
    written += ooffset;                                           \
    ooffset = 0;                                                  \
  } while (0)
  for (size_t f = 0; f < nfiles; ++f) {
    memset(&cat.pending, 0, sizeof(cat.pending)); /* new_brotli_file() */
    cat.has_pending = 1;
    for (size_t pos = 0; pos < sizes[f];) {
      const size_t cur_read = ORC_MIN(bs, sizes[f] - pos);
      size_t ioffset = 0;
      for (;;) {
        const int r = brocatli_stream(&cat, files[f] + pos, cur_read, &ioffset, obuf, bs, &ooffset);
        if (r == CAT_NEEDS_MORE_OUTPUT) {
          ORC_CONCAT_WRITE();
        } else if (r == CAT_NEEDS_MORE_INPUT) {
          break;
        } else {
          result = r == CAT_SUCCESS ? -2 : r;
          goto done;
        }
      }
      pos += cur_read;
    }
  }
  for (;;) {
    const int r = brocatli_finish(&cat, obuf, bs, &ooffset);
    if (r == CAT_NEEDS_MORE_OUTPUT) {
      ORC_CONCAT_WRITE();
    } else if (r == CAT_SUCCESS) {
      ORC_CONCAT_WRITE();
      break;
    } else {
      result = r;
      goto done;
    }
  }
done:
  free(obuf);
  *out_size = written;
  return result;
}
