/* brotli/batch.h -- batch extension of the C ABI: many independent .br streams per call.
 *
 * Not in the reference: its API decodes one stream per BrotliState (src/state.rs:156-278).  A GPU only
 * pays off when many streams decode at once, so the same decode path is also exposed in batch form.  The
 * per-stream outcome has exactly the meaning of the reference's one-shot return info (src/lib.rs:336-370):
 * result, error code and the number of bytes the reference would have delivered.
 *
 * Threading: a batch object may be used by one thread at a time; different batch objects are independent.
 * One batch object is bound to the HIP device that was current when it was created.
 */
#ifndef BROTLI_AMD_BATCH_H_
#define BROTLI_AMD_BATCH_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#ifndef BROTLI_DEC_API
#define BROTLI_DEC_API __attribute__((visibility("default")))
#endif

typedef struct BrotliAmdBatch BrotliAmdBatch;

/* per-stream outcome (same fields as BrotliAmdStreamStatus without the resume block) */
typedef struct BrotliAmdResult {
  int32_t result;        /* BrotliDecoderResult */
  int32_t error_code;    /* BrotliDecoderErrorCode */
  uint64_t decoded_size; /* bytes delivered (reference: BrotliDecoderReturnInfo.decoded_size) */
  uint64_t consumed;     /* input bytes consumed */
  uint64_t produced;     /* bytes written to the output buffer (>= decoded_size after an error) */
  uint32_t num_metablocks;
  uint32_t spilled_metablocks; /* metablocks whose tables did not fit the LDS arena (slower path; raise lds_arena_bytes) */
  uint64_t num_commands;
  uint32_t engine_commands;    /* of num_commands, how many a command engine took (blocks of sixteen waves, one stream a CU) */
  uint32_t reserved;
} BrotliAmdResult;

#define BROTLI_AMD_BATCH_LARGE_WINDOW 1u /* accept large-window streams (reference one-shot default, lib.rs:457) */
#define BROTLI_AMD_BATCH_NO_CANNY 2u     /* BROTLI_DECODER_PARAM_DISABLE_RING_BUFFER_REALLOCATION */
#define BROTLI_AMD_BATCH_SPILL_IN_PLACE 16u /* no second launch with a larger LDS arena for streams whose prefix-code tables
                                              do not fit the arena of the first: they spill to global memory (slower) */
#define BROTLI_AMD_BATCH_EAGER_OUTPUT_LIMIT 64u /* A stream whose output buffer is too small is reported NEEDS_MORE_OUTPUT as soon
                                              as the buffer is full.  Without this flag such streams get the reference's
                                              verdict: the reference decodes into its ring buffer and only notices the full
                                              buffer at the next flush point (decode.rs:1693-1738), so an error or the end of
                                              the input in front of that point is what it reports; the batch decodes those
                                              streams a second time, into scratch memory with room up to that point (up to
                                              one window per stream, 2 GiB at a time), inside BrotliAmdBatchWait. */

/* Creates a batch context on the current HIP device for up to max_streams streams per call.
 * lds_arena_bytes = 0 and grid_blocks = 0 select the defaults.  NULL when no device is usable. */
BROTLI_DEC_API BrotliAmdBatch* BrotliAmdBatchCreate(uint32_t max_streams, uint32_t lds_arena_bytes, uint32_t grid_blocks);
BROTLI_DEC_API void BrotliAmdBatchDestroy(BrotliAmdBatch* batch);

/* Decodes n streams whose compressed bytes and output buffers already live in DEVICE memory.
 * d_in[i]/d_out[i] are device pointers (any alignment).  The launch is asynchronous on hip_stream
 * (a hipStream_t, NULL = default stream); call BrotliAmdBatchWait before reading results.
 * One case is NOT asynchronous: a batch of more streams than the device has compute units and at most four times as many, of a mean
 * compressed size of 8 KiB or more, is PROBED first -- a short launch that reads every stream's header and tells the host which streams
 * the command engines can take -- and the call waits on hip_stream for its answer (some tens of microseconds of kernel; it cannot be
 * captured into a graph).  The answer is kept with the batch object: the same descriptors again (pointers, sizes, flags) are not probed
 * again, and BrotliAmdBatchRelaunch never probes.  BrotliAmdBatchLastProbeMs says what the last call spent there.
 * Returns 0 on success, a negative value if the arguments or the device are unusable. */
BROTLI_DEC_API int BrotliAmdBatchDecodeDevice(BrotliAmdBatch* batch, uint32_t n, const void* const* d_in, const size_t* in_sizes,
                                             void* const* d_out, const size_t* out_caps, uint32_t flags, void* hip_stream);

/* Re-launches the decode of the streams described by the previous BrotliAmdBatchDecodeDevice call
 * (descriptors stay resident in device memory); used to time the kernel with inputs already in HBM. */
BROTLI_DEC_API int BrotliAmdBatchRelaunch(BrotliAmdBatch* batch, void* hip_stream);

/* Waits for the last launch and copies the per-stream results to the host. */
BROTLI_DEC_API int BrotliAmdBatchWait(BrotliAmdBatch* batch, BrotliAmdResult* results /* n entries, may be NULL */);

/* Convenience for host buffers: upload, decode, download.  Same per-stream semantics. */
BROTLI_DEC_API int BrotliAmdBatchDecodeHost(BrotliAmdBatch* batch, uint32_t n, const uint8_t* const* in, const size_t* in_sizes,
                                           uint8_t* const* out, const size_t* out_caps, uint32_t flags, BrotliAmdResult* results);

/* Milliseconds the last launch spent in the decode kernel (HIP events on the launch stream). */
BROTLI_DEC_API float BrotliAmdBatchLastKernelMs(BrotliAmdBatch* batch);

/* Host milliseconds the last BrotliAmdBatchDecodeDevice / DecodeHost call spent in the probe launch and its wait (0: no probe). */
BROTLI_DEC_API float BrotliAmdBatchLastProbeMs(BrotliAmdBatch* batch);

/* Streams the last BrotliAmdBatchWait had to continue in a second launch with a larger LDS arena (0 in the common case). */
BROTLI_DEC_API uint32_t BrotliAmdBatchLastSecondPassCount(BrotliAmdBatch* batch);

/* Blocks (CUs) that worked on each stream of the last launch: 1 as a rule; 2, 4, 8 or 16 where the batch had fewer streams than half the
 * device's CUs, at least one of them 64 KiB of compressed data or more, and each stream was given a gang of blocks (csrc/brotli_path_engine.h, PE_CFG_REMOTE; BROTLI_AMD_GANG=0 turns that off). */
BROTLI_DEC_API uint32_t BrotliAmdBatchLastGang(BrotliAmdBatch* batch);

/* 1 where the last launch was a POOL: more than 32 streams, at most as many as CUs, of very different sizes (the largest more than twice the median) --
 * every block that has no stream of its own, at once or when its stream is done, joins the largest stream still being decoded
 * (BROTLI_AMD_POOL=0 turns that off).  BrotliAmdBatchLastGang says 1 for such a launch: a stream's helpers come and go. */
BROTLI_DEC_API uint32_t BrotliAmdBatchLastPool(BrotliAmdBatch* batch);

/* Test hook (no device needed): what a launch of n streams of these compressed sizes gets on a device of `cus` compute units -- 0 one block a
 * stream, 2 / 4 / 8 / 16 gangs of that many blocks a stream, 0x108 a pool -- and its number of blocks in *grid.  gang_env / pool_env: the values of
 * BROTLI_AMD_GANG / BROTLI_AMD_POOL, -1 where unset. */
BROTLI_DEC_API uint32_t BrotliAmdDebugPlanGangs(uint32_t n, uint32_t cus, const size_t* in_sizes, int gang_env, int pool_env, uint32_t* grid);

/* Streaming (BrotliDecoderDecompressStream, decode.h): the commands the device has decoded for this stream in all the launches
 * of its calls together.  A call is a launch from the last command boundary reached, so this stays close to the stream's own
 * number of commands however the input is cut up; a test asserts that instead of timing calls. */
struct BrotliDecoderStateStruct;
BROTLI_DEC_API uint64_t BrotliAmdDecoderDeviceCommands(const struct BrotliDecoderStateStruct* state);

/* Text of the last HIP/runtime failure on this thread ("" if none). */
BROTLI_DEC_API const char* BrotliAmdLastError(void);

/* What the last call on this thread did differently without failing ("" if nothing): a device that refused blocks of sixteen
 * waves makes its batch context go on with blocks of eight (slower on batches of one stream a CU), and this says so. */
BROTLI_DEC_API const char* BrotliAmdLastNote(void);

/* Test hook, not part of the decode path: builds the device's prefix-code table for alphabet_size code lengths (0 = unused symbol;
 * a complete code, as src/huffman/mod.rs:273-386 is given them) and decodes every fifteen-bit value v through it the way the
 * kernel's symbol reader does: decoded[v] = symbol << 4 | code length (32768 entries).  table_entries = the table's size.
 * Returns 0 on success. */
BROTLI_DEC_API int BrotliAmdDebugBuildTree(const uint8_t* code_lengths, uint32_t alphabet_size, uint16_t* decoded, uint32_t* table_entries);

#if defined(__cplusplus)
} /* extern "C" */
#endif
#endif /* BROTLI_AMD_BATCH_H_ */
