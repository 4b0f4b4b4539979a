/* brotli/decode.h -- C ABI of the MI355X-native Brotli decoder (libbrotli_decompressor.so).
 *
 * This is the drop-in boundary: the same 20 entry points the reference exports from its cdylib
 * (reference src/ffi/mod.rs, declared in its c/brotli/decode.h), with the same struct layouts, enum values
 * and calling conventions, so that a program written against the reference -- e.g. its c/main.c -- builds
 * and runs against this library unchanged.  Each declaration cites the reference symbol it replaces
 * (ffi/mod.rs line -> c/brotli/decode.h line).  Behind these symbols every byte is decoded by the HIP
 * kernels of rust-brotli-decompressor_amd/csrc/brotli_kernels.hip; there is no CPU decode path, and every
 * entry point fails with BROTLI_DECODER_ERROR_UNREACHABLE when no HIP device is usable.
 *
 * The batch entry points (the only shape in which a GPU wins) are declared in brotli/batch.h.
 */
#ifndef BROTLI_AMD_DEC_DECODE_H_
#define BROTLI_AMD_DEC_DECODE_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define BROTLI_BOOL int
#define BROTLI_TRUE 1
#define BROTLI_FALSE 0
#define BROTLI_DEC_API __attribute__((visibility("default")))

/* memory callbacks (reference c/brotli/types.h:71-81, src/ffi/interface.rs:7-15) */
typedef void* (*brotli_alloc_func)(void* opaque, size_t size);
typedef void (*brotli_free_func)(void* opaque, void* address);

typedef struct BrotliDecoderStateStruct BrotliDecoderState;

/* reference src/huffman/mod.rs:28-33 (repr(C)); only an ABI type of BrotliDecoderDecompressPrealloc here */
typedef struct HuffmanCodeStruct {
  uint16_t value;
  uint8_t bits;
} HuffmanCode;

/* reference src/ffi/interface.rs:17-22 */
typedef enum {
  BROTLI_DECODER_RESULT_ERROR = 0,
  BROTLI_DECODER_RESULT_SUCCESS = 1,
  BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT = 2,
  BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT = 3
} BrotliDecoderResult;

/* reference src/state.rs:22-65 */
typedef enum {
  BROTLI_DECODER_NO_ERROR = 0,
  BROTLI_DECODER_SUCCESS = 1,
  BROTLI_DECODER_NEEDS_MORE_INPUT = 2,
  BROTLI_DECODER_NEEDS_MORE_OUTPUT = 3,
  BROTLI_DECODER_ERROR_FORMAT_EXUBERANT_NIBBLE = -1,
  BROTLI_DECODER_ERROR_FORMAT_RESERVED = -2,
  BROTLI_DECODER_ERROR_FORMAT_EXUBERANT_META_NIBBLE = -3,
  BROTLI_DECODER_ERROR_FORMAT_SIMPLE_HUFFMAN_ALPHABET = -4,
  BROTLI_DECODER_ERROR_FORMAT_SIMPLE_HUFFMAN_SAME = -5,
  BROTLI_DECODER_ERROR_FORMAT_CL_SPACE = -6,
  BROTLI_DECODER_ERROR_FORMAT_HUFFMAN_SPACE = -7,
  BROTLI_DECODER_ERROR_FORMAT_CONTEXT_MAP_REPEAT = -8,
  BROTLI_DECODER_ERROR_FORMAT_BLOCK_LENGTH_1 = -9,
  BROTLI_DECODER_ERROR_FORMAT_BLOCK_LENGTH_2 = -10,
  BROTLI_DECODER_ERROR_FORMAT_TRANSFORM = -11,
  BROTLI_DECODER_ERROR_FORMAT_DICTIONARY = -12,
  BROTLI_DECODER_ERROR_FORMAT_WINDOW_BITS = -13,
  BROTLI_DECODER_ERROR_FORMAT_PADDING_1 = -14,
  BROTLI_DECODER_ERROR_FORMAT_PADDING_2 = -15,
  BROTLI_DECODER_ERROR_FORMAT_DISTANCE = -16,
  BROTLI_DECODER_ERROR_DICTIONARY_NOT_SET = -19,
  BROTLI_DECODER_ERROR_INVALID_ARGUMENTS = -20,
  BROTLI_DECODER_ERROR_ALLOC_CONTEXT_MODES = -21,
  BROTLI_DECODER_ERROR_ALLOC_TREE_GROUPS = -22,
  BROTLI_DECODER_ERROR_ALLOC_CONTEXT_MAP = -25,
  BROTLI_DECODER_ERROR_ALLOC_RING_BUFFER_1 = -26,
  BROTLI_DECODER_ERROR_ALLOC_RING_BUFFER_2 = -27,
  BROTLI_DECODER_ERROR_ALLOC_BLOCK_TYPE_TREES = -30,
  BROTLI_DECODER_ERROR_UNREACHABLE = -31
} BrotliDecoderErrorCode;
#define BROTLI_LAST_ERROR_CODE BROTLI_DECODER_ERROR_UNREACHABLE

/* reference src/lib.rs:336-342 (272 bytes on LP64) */
typedef struct BrotliDecoderReturnInfoStruct {
  size_t decoded_size;
  char error[256];
  BrotliDecoderResult result;
  BrotliDecoderErrorCode code;
} BrotliDecoderReturnInfo;

/* reference src/ffi/interface.rs:24-29 */
typedef enum BrotliDecoderParameter {
  BROTLI_DECODER_PARAM_DISABLE_RING_BUFFER_REALLOCATION = 0,
  BROTLI_DECODER_PARAM_LARGE_WINDOW = 1
} BrotliDecoderParameter;

/* ffi/mod.rs:156 -> decode.h:167.  Only while the instance has seen no input; else returns 0. */
BROTLI_DEC_API BROTLI_BOOL BrotliDecoderSetParameter(BrotliDecoderState* state, BrotliDecoderParameter param, uint32_t value);

/* ffi/mod.rs:108 -> decode.h:188.  Both callbacks or neither; the instance itself comes from alloc_func.
 * Instances created here reject the large-window header until the parameter is set (ffi/mod.rs:127). */
BROTLI_DEC_API BrotliDecoderState* BrotliDecoderCreateInstance(brotli_alloc_func alloc_func, brotli_free_func free_func, void* opaque);

/* ffi/mod.rs:533 -> decode.h:196 */
BROTLI_DEC_API void BrotliDecoderDestroyInstance(BrotliDecoderState* state);

/* ffi/mod.rs:263 -> decode.h:215.  One-shot; SUCCESS only if the whole stream fits; *decoded_size = bytes
 * written.  Accepts large-window streams (lib.rs:457). */
BROTLI_DEC_API BrotliDecoderResult BrotliDecoderDecompress(size_t encoded_size, const uint8_t* encoded_buffer, size_t* decoded_size,
                                                           uint8_t* decoded_buffer);

/* ffi/mod.rs:246 -> decode.h:221 */
BROTLI_DEC_API BrotliDecoderReturnInfo BrotliDecoderDecompressWithReturnInfo(size_t encoded_size, const uint8_t* encoded_buffer,
                                                                             size_t decoded_size, uint8_t* decoded_buffer);

/* ffi/mod.rs:179 -> decode.h:227.  The reference decodes out of the three scratch slices (src/lib.rs:374-401) and
 * reports a request they cannot serve as ERROR_UNREACHABLE with decoded_size 0 (ffi/mod.rs:686-713).  Decoder state
 * lives in device memory here, so nothing is stored in them, but the same requests are accounted against their sizes:
 * 1080 HuffmanCode cells at creation (state.rs:395), 6 x 1080 more at the first compressed metablock
 * (decode.rs:2958-2969), per metablock 1080 cells and one uint32_t per prefix code (huffman/mod.rs:61-72) and one byte
 * per context mode / context-map entry (decode.rs:1295, 3155), and ring size + 66 bytes for the ring buffer
 * (decode.rs:1843-1855).  Known divergence: the model is the peak of what is alive at once; slices that hold the peak
 * but are too fragmented for the reference's first-fit allocator succeed here and fail there. */
BROTLI_DEC_API BrotliDecoderReturnInfo BrotliDecoderDecompressPrealloc(size_t encoded_size, const uint8_t* encoded_buffer, size_t decoded_size,
                                                                       uint8_t* decoded_buffer, size_t scratch_u8_size,
                                                                       uint8_t* scratch_u8_buffer, size_t scratch_u32_size,
                                                                       uint32_t* scratch_u32_buffer, size_t scratch_hc_size,
                                                                       HuffmanCode* scratch_hc_buffer);

/* ffi/mod.rs:390 -> decode.h:278.  total_out may be NULL; input is never over-consumed on SUCCESS.
 * A call whose input ends inside the stream writes what fits *available_out, takes ALL of its input and returns
 * NEEDS_MORE_INPUT; what did not fit stays with the decoder and goes out with later calls (decode.rs:2835-2846).  Output the
 * decoder OWES -- at the end of the stream, in front of an error, or a whole window's worth not yet taken (the reference's
 * full ring buffer, decode.rs:1693-1738) -- comes first: while it does not fit, a call consumes no further input and returns
 * NEEDS_MORE_OUTPUT.  Cost model of this implementation: the compressed bytes of a call are copied to the device
 * and a kernel is launched that goes on where the call before got to: it parses the header of the metablock in
 * flight again (prefix codes are not kept between launches) and continues from the last command boundary that launch
 * reached, like the reference's resumable state.  A call costs a launch (about a millisecond) plus its bytes; an
 * instance keeps about one window plus the compressed metablock in flight on the device, not the whole stream.  The
 * batch and one-shot entry points are the fast paths. */
BROTLI_DEC_API BrotliDecoderResult BrotliDecoderDecompressStream(BrotliDecoderState* state, size_t* available_in, const uint8_t** next_in,
                                                                 size_t* available_out, uint8_t** next_out, size_t* total_out);

/* ffi/mod.rs:467 (reference-only extension: no double indirection, no total_out) */
BROTLI_DEC_API BrotliDecoderResult BrotliDecoderDecompressStreaming(BrotliDecoderState* state, size_t* available_in, const uint8_t* next_in,
                                                                    size_t* available_out, uint8_t* next_out);

/* ffi/mod.rs:546-580 -> decode.h:289-372 */
BROTLI_DEC_API BROTLI_BOOL BrotliDecoderHasMoreOutput(const BrotliDecoderState* state);
BROTLI_DEC_API const uint8_t* BrotliDecoderTakeOutput(BrotliDecoderState* state, size_t* size);
BROTLI_DEC_API BROTLI_BOOL BrotliDecoderIsUsed(const BrotliDecoderState* state);
BROTLI_DEC_API BROTLI_BOOL BrotliDecoderIsFinished(const BrotliDecoderState* state);
BROTLI_DEC_API BrotliDecoderErrorCode BrotliDecoderGetErrorCode(const BrotliDecoderState* state);
BROTLI_DEC_API const char* BrotliDecoderGetErrorString(const BrotliDecoderState* state);

/* ffi/mod.rs:582-590 -> decode.h:377-384; strings of src/state.rs:533-578 */
BROTLI_DEC_API const char* BrotliDecoderErrorString(BrotliDecoderErrorCode c);
BROTLI_DEC_API uint32_t BrotliDecoderVersion(void);

/* ffi/mod.rs:493-530 (reference-only helpers: allocate through the instance's allocator) */
BROTLI_DEC_API uint8_t* BrotliDecoderMallocU8(BrotliDecoderState* state, size_t size);
BROTLI_DEC_API void BrotliDecoderFreeU8(BrotliDecoderState* state, uint8_t* data, size_t size);
BROTLI_DEC_API size_t* BrotliDecoderMallocUsize(BrotliDecoderState* state, size_t size);
BROTLI_DEC_API void BrotliDecoderFreeUsize(BrotliDecoderState* state, size_t* data, size_t size);

#if defined(__cplusplus)
} /* extern "C" */
#endif
#endif /* BROTLI_AMD_DEC_DECODE_H_ */
