/* brotli_mi355x.h -- C ABI of the MI355X (gfx950) brotli encoder hot path.
 *
 * Drop-in for the encoder half of rust-brotli's C ABI: every entry point below has the same name,
 * signature and meaning as the one the reference exports from its cdylib (headers c/brotli/encode.h and
 * c/brotli/multiencode.h, implementation src/ffi/compressor.rs and src/ffi/multicompress/mod.rs).  The
 * reference location each function replaces is cited next to it.  Plain pointers and sizes only.
 *
 * Behavioural contract: for the accelerated configurations -- every quality 0..11 at every window the reference takes: the
 * H5 / H5q5 / H6 / H9 greedy path of qualities 5..9; quality 10 / 11 with BROTLI_PARAM_Q9_5 through the stream API (the
 * reference's "9.5": the H9 search with the quality >= 10 meta-block builder, metablock.rs:133-307) and without it (H10 + Zopfli);
 * qualities 2..4 (BasicHasher family) and 0 / 1 (the fragment compressors, on the fragment path and, for catable streams / custom
 * dictionaries / shards, block by block like the reference's ring-buffer path) -- the produced
 * stream is byte-identical to the reference encoder fed the same way.  Everything runs on the GPU through HIP; there is no CPU fallback: calls with
 * parameters outside the accelerated set, or on a machine without a usable gfx950 device, fail
 * (BROTLI_FALSE / 0 / NULL) and print the reason on stderr.
 *
 * Streaming (bounded memory): input handed over with BROTLI_OPERATION_PROCESS is copied into the encoder (as the
 * reference copies it into its ring buffer).  Where the reference runs encode_data whenever a 64 KiB input block is
 * full (encode.rs:2959-2964), this encoder waits until a batch has piled up (64 MiB; BROTLI_MI355X_STREAM_BATCH bytes),
 * then encodes the meta-blocks that the reference's flush rule (encode.rs:2454-2477) closes within the whole blocks
 * received so far and makes their bytes available at once: BrotliEncoderHasMoreOutput() becomes true in the middle of a
 * stream, before any FLUSH / FINISH.  Of what has been encoded only a window is kept as the LZ77 prefix of the next
 * piece (one to two ring-buffer sizes, 8..16 MiB at lgwin 22), together with the state the reference carries: distance
 * cache, static-dictionary throttle counters, which window positions are in the hash table, the per-key insertion
 * counters, the open last byte of the output.  Memory is therefore bounded by window + batch however long the stream;
 * the bytes are those of the reference's stream encoder fed with the same writes (tests/test_streaming.py).
 * BROTLI_OPERATION_FLUSH encodes everything received so far (byte-identical to the reference's flush,
 * encode.rs:2940-2975 + 1541-1566) at a cost proportional to window + new input; BROTLI_OPERATION_EMIT_METADATA
 * (encode.rs:2579-2685) flushes pending input the same way and writes the payload (<= 16 MiB) as a metadata block.
 * A stream of quality 5..9 (and "9.5") may be of any length: the hasher reset of the reference at its position wraps
 * (3, 5, 7 ... GiB, encode.rs:1623-1631, 1705-1710) is reproduced; qualities 0 / 1 keep no hasher between fragments and have no
 * such limit either.  Limits: a stream of quality 2..4 or 10 / 11 must stay below the first wrap (3 GiB) -- their hashers (the
 * BasicHasher table, the H10 trees) travel through the stream as they are and the reset is not reproduced for them: the call
 * that would cross the wrap fails (BrotliMi355xLastError says why; the state is unusable afterwards, like after any failed
 * call), and BrotliEncoderCompress refuses such an input up front, before any work is done.  Streams with a custom
 * dictionary or in the catable / appendable modes stream and flush the same way, piece by piece in bounded memory
 * (tests/test_streaming_dictionary.py).  All input offered to a call is always consumed (*available_in becomes 0).
 *
 * BrotliEncoderCompress (one shot) takes inputs of any size: above 1 GiB (BROTLI_MI355X_ONESHOT_STREAM_ABOVE bytes) the
 * call runs through the same stream state machine in batches -- encoder_compress is a loop over compress_stream with
 * FINISH (encode.rs:1484-1520) -- so its device memory stays bounded as well and the bytes are the reference's one-shot
 * stream (tests/test_oneshot_streamed.py, tests/test_large_gpu.py::test_one_shot_above_2GiB).
 */
#ifndef BROTLI_MI355X_H_
#define BROTLI_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BROTLI_BOOL int
#define BROTLI_TRUE 1
#define BROTLI_FALSE 0

/* c/brotli/types.h:71,81 */
typedef void* (*brotli_alloc_func)(void* opaque, size_t size);
typedef void (*brotli_free_func)(void* opaque, void* address);

/* c/brotli/encode.h:45-61 (+ the reference's extra prior modes, src/ffi/compressor.rs:30-38) */
typedef enum BrotliEncoderMode {
  BROTLI_MODE_GENERIC = 0,
  BROTLI_MODE_TEXT = 1,
  BROTLI_MODE_FONT = 2,
  BROTLI_FORCE_LSB_PRIOR = 3,
  BROTLI_FORCE_MSB_PRIOR = 4,
  BROTLI_FORCE_UTF8_PRIOR = 5,
  BROTLI_FORCE_SIGNED_PRIOR = 6
} BrotliEncoderMode;

/* c/brotli/encode.h:71-135 */
typedef enum BrotliEncoderOperation {
  BROTLI_OPERATION_PROCESS = 0,
  BROTLI_OPERATION_FLUSH = 1,
  BROTLI_OPERATION_FINISH = 2,
  BROTLI_OPERATION_EMIT_METADATA = 3
} BrotliEncoderOperation;

/* c/brotli/encode.h:138-232, src/enc/parameters.rs:3-33 */
typedef enum BrotliEncoderParameter {
  BROTLI_PARAM_MODE = 0,
  BROTLI_PARAM_QUALITY = 1,
  BROTLI_PARAM_LGWIN = 2,
  BROTLI_PARAM_LGBLOCK = 3,
  BROTLI_PARAM_DISABLE_LITERAL_CONTEXT_MODELING = 4,
  BROTLI_PARAM_SIZE_HINT = 5,
  BROTLI_PARAM_LARGE_WINDOW = 6,
  BROTLI_PARAM_NPOSTFIX = 7,
  BROTLI_PARAM_NDIRECT = 8,
  BROTLI_PARAM_Q9_5 = 150,
  BROTLI_METABLOCK_CALLBACK = 151,
  BROTLI_PARAM_STRIDE_DETECTION_QUALITY = 152,
  BROTLI_PARAM_HIGH_ENTROPY_DETECTION_QUALITY = 153,
  BROTLI_PARAM_LITERAL_BYTE_SCORE = 154,
  BROTLI_PARAM_CDF_ADAPTATION_DETECTION = 155,
  BROTLI_PARAM_PRIOR_BITMASK_DETECTION = 156,
  BROTLI_PARAM_SPEED = 157,
  BROTLI_PARAM_SPEED_MAX = 158,
  BROTLI_PARAM_CM_SPEED = 159,
  BROTLI_PARAM_CM_SPEED_MAX = 160,
  BROTLI_PARAM_SPEED_LOW = 161,
  BROTLI_PARAM_SPEED_LOW_MAX = 162,
  BROTLI_PARAM_CM_SPEED_LOW = 164,
  BROTLI_PARAM_CM_SPEED_LOW_MAX = 165,
  BROTLI_PARAM_AVOID_DISTANCE_PREFIX_SEARCH = 166,
  BROTLI_PARAM_CATABLE = 167,
  BROTLI_PARAM_APPENDABLE = 168,
  BROTLI_PARAM_MAGIC_NUMBER = 169,
  BROTLI_PARAM_NO_DICTIONARY = 170,
  BROTLI_PARAM_FAVOR_EFFICIENCY = 171,
  BROTLI_PARAM_BYTE_ALIGN = 172,
  BROTLI_PARAM_BARE_STREAM = 173
} BrotliEncoderParameter;

typedef struct BrotliEncoderStateStruct BrotliEncoderState;
typedef struct BrotliEncoderWorkPoolStruct BrotliEncoderWorkPool;

/* ---- single-stream encoder: c/brotli/encode.h:256-457 ---- */
/* src/ffi/compressor.rs:72 */
BrotliEncoderState* BrotliEncoderCreateInstance(brotli_alloc_func alloc_func, brotli_free_func free_func, void* opaque);
/* src/ffi/compressor.rs:115 (returns BROTLI_FALSE once the encoder has started, encode.rs:289-295) */
BROTLI_BOOL BrotliEncoderSetParameter(BrotliEncoderState* state, BrotliEncoderParameter p, uint32_t value);
/* src/ffi/compressor.rs:128 (NULL is accepted) */
void BrotliEncoderDestroyInstance(BrotliEncoderState* state);
/* src/ffi/compressor.rs:190, src/enc/encode.rs:1276-1299 */
size_t BrotliEncoderMaxCompressedSize(size_t input_size);
/* src/ffi/compressor.rs:194, src/enc/encode.rs:1436-1538 */
BROTLI_BOOL BrotliEncoderCompress(int quality, int lgwin, BrotliEncoderMode mode, size_t input_size,
                                  const uint8_t* input_buffer, size_t* encoded_size, uint8_t* encoded_buffer);
/* src/ffi/compressor.rs:280, src/enc/encode.rs:2873-2995 */
BROTLI_BOOL BrotliEncoderCompressStream(BrotliEncoderState* state, BrotliEncoderOperation op, size_t* available_in,
                                        const uint8_t** next_in, size_t* available_out, uint8_t** next_out,
                                        size_t* total_out);
/* src/ffi/compressor.rs:260 */
BROTLI_BOOL BrotliEncoderCompressStreaming(BrotliEncoderState* state, BrotliEncoderOperation op, size_t* available_in,
                                           const uint8_t* next_in, size_t* available_out, uint8_t* next_out);
/* src/ffi/compressor.rs:144,153,179 */
BROTLI_BOOL BrotliEncoderIsFinished(BrotliEncoderState* state);
BROTLI_BOOL BrotliEncoderHasMoreOutput(BrotliEncoderState* state);
const uint8_t* BrotliEncoderTakeOutput(BrotliEncoderState* state, size_t* size);
/* src/ffi/compressor.rs:162, src/enc/encode.rs:1196-1270 */
void BrotliEncoderSetCustomDictionary(BrotliEncoderState* state, size_t size, const uint8_t* dict);
/* src/ffi/compressor.rs:186: 0x01000f01 */
uint32_t BrotliEncoderVersion(void);
/* src/ffi/compressor.rs:359-417 */
uint8_t* BrotliEncoderMallocU8(BrotliEncoderState* state, size_t size);
void BrotliEncoderFreeU8(BrotliEncoderState* state, uint8_t* data, size_t size);
size_t* BrotliEncoderMallocUsize(BrotliEncoderState* state, size_t size);
void BrotliEncoderFreeUsize(BrotliEncoderState* state, size_t* data, size_t size);

/* ---- multi-chunk encoder: c/brotli/multiencode.h:41-127 ---- */
/* src/ffi/multicompress/mod.rs:49 */
size_t BrotliEncoderMaxCompressedSizeMulti(size_t input_size, size_t num_threads);
/* src/ffi/multicompress/mod.rs:93: the input is split into desired_num_threads (<= 16) chunks exactly as the
   reference's compress_multi does (src/enc/threading/mod.rs:333-411); chunks are compressed on the GPU and
   stitched with the BroCatli rules (src/concat/mod.rs). */
int32_t BrotliEncoderCompressMulti(size_t num_params, const BrotliEncoderParameter* param_keys, const uint32_t* param_values,
                                   size_t input_size, const uint8_t* input_buffer, size_t* encoded_size, uint8_t* encoded,
                                   size_t desired_num_threads, brotli_alloc_func alloc_func, brotli_free_func free_func,
                                   void** alloc_opaque_per_thread);
/* src/ffi/multicompress/mod.rs:240,294,312: the work pool owns no threads here (the GPU is the pool) */
BrotliEncoderWorkPool* BrotliEncoderCreateWorkPool(size_t num_threads, brotli_alloc_func alloc_func, brotli_free_func free_func,
                                                   void** alloc_opaque_per_thread);
void BrotliEncoderDestroyWorkPool(BrotliEncoderWorkPool* work_pool);
int32_t BrotliEncoderCompressWorkPool(BrotliEncoderWorkPool* work_pool, size_t num_params, const BrotliEncoderParameter* param_keys,
                                      const uint32_t* param_values, size_t input_size, const uint8_t* input_buffer,
                                      size_t* encoded_size, uint8_t* encoded, size_t desired_num_threads,
                                      brotli_alloc_func alloc_func, brotli_free_func free_func, void** alloc_opaque_per_thread);

/* ---- extensions of this implementation (not part of the reference ABI) ---- */
/* Compresses chunk `thread_index` of `num_threads` of a compress_multi job (what one reference worker thread
   does in compress_part, src/enc/threading/mod.rs:337-411).  input_on_device != 0: input_buffer is a device
   pointer (data resident in HBM).  Used to spread the chunks over several GPUs (one process per GPU). */
int32_t BrotliMi355xCompressChunk(size_t num_params, const BrotliEncoderParameter* param_keys, const uint32_t* param_values,
                                  size_t input_size, const uint8_t* input_buffer, int input_on_device, size_t thread_index,
                                  size_t num_threads, size_t* encoded_size, uint8_t* encoded);
/* Stitches already compressed chunks (in order) into one stream: BroCatli new_brotli_file/stream/finish,
   src/enc/threading/mod.rs:565-660.  Returns 1 on success. */
int32_t BrotliMi355xConcatChunks(size_t num_chunks, const uint8_t* const* chunks, const size_t* chunk_sizes,
                                 size_t* encoded_size, uint8_t* encoded);
/* The same for chunks whose bodies live elsewhere (device memory): only the first and last min(8, size) bytes of every
   chunk are given (heads[8 * i ..], tails[8 * i ..]).  The junction bytes are written into `encoded`; for chunk i the
   caller then copies body_copies[3 * i + 2] bytes from offset body_copies[3 * i + 1] of the chunk to offset
   body_copies[3 * i] of `encoded`. */
int32_t BrotliMi355xConcatChunkEnds(size_t num_chunks, const uint8_t* heads, const uint8_t* tails,
                                    const size_t* chunk_sizes, size_t* encoded_size, uint8_t* encoded,
                                    size_t* body_copies);
/* One-shot compression of a buffer that is already resident in device memory; per-stage timings (ms) are
   returned in stats[0..32) (may be NULL).  Same stream as BrotliEncoderCompress on the same bytes. */
BROTLI_BOOL BrotliMi355xCompressDevice(int quality, int lgwin, BrotliEncoderMode mode, size_t input_size,
                                       const uint8_t* input_device, size_t* encoded_size, uint8_t* encoded_host,
                                       double* stats);
/* Human-readable description of the device backing the library ("hip:gfx950 (...)"). */
const char* BrotliMi355xDeviceName(void);
/* Message of the last failure on the calling thread ("" if none). */
const char* BrotliMi355xLastError(void);

/* ---- BroCatli: concatenation of catable / appendable brotli streams, fed and drained piecewise --------------------
 * c/brotli/broccoli.h, src/ffi/broccoli.rs:55-175 (state machine src/concat/mod.rs:274-608).  Same names, signatures
 * and result codes as the reference; the state object is the same 256-byte by-value struct. */
typedef struct BroccoliState_ {
  void* unused;
  unsigned char data[248];
} BroccoliState;

typedef enum BroccoliResult_ {
  BroccoliSuccess = 0,
  BroccoliNeedsMoreInput = 1,
  BroccoliNeedsMoreOutput = 2,
  BroccoliBrotliFileNotCraftedForAppend = 124,
  BroccoliInvalidWindowSize = 125,
  BroccoliWindowSizeLargerThanPreviousFile = 126,
  BroccoliBrotliFileNotCraftedForConcatenation = 127
} BroccoliResult;

BroccoliState BroccoliCreateInstance(void);                                   /* broccoli.rs:56 */
BroccoliState BroccoliCreateInstanceWithWindowSize(uint8_t window_size);      /* broccoli.rs:60 */
void BroccoliDestroyInstance(BroccoliState state);                            /* broccoli.rs:67 */
void BroccoliNewBrotliFile(BroccoliState* state);                             /* broccoli.rs:70 */
BroccoliResult BroccoliConcatStream(BroccoliState* state, size_t* available_in, const uint8_t** input_buf_ptr,
                                    size_t* available_out, uint8_t** output_buf_ptr);  /* broccoli.rs:82 */
BroccoliResult BroccoliConcatStreaming(BroccoliState* state, size_t* available_in, const uint8_t* input_buf,
                                       size_t* available_out, uint8_t* output_buf);    /* broccoli.rs:110 */
BroccoliResult BroccoliConcatFinish(BroccoliState* state, size_t* available_out, uint8_t** output_buf);   /* broccoli.rs:133 */
BroccoliResult BroccoliConcatFinished(BroccoliState* state, size_t* available_out, uint8_t* output_buf);  /* broccoli.rs:156 */

/* returns the device memory the calling thread's encoder calls have pooled (and do not use at the moment) to the
 * driver; the number of bytes released.  Not part of the reference ABI. */
size_t BrotliMi355xTrimPool(void);

#ifdef __cplusplus
}
#endif
#endif /* BROTLI_MI355X_H_ */
