// Host <-> kernel contract of the batch decoder (plain C structs, no HIP types).
//
// One StreamDesc per .br stream, one StreamStatus written back per stream.  A stream is the unit of
// work the reference calls "one BrotliState" (src/state.rs:156-278): its own window, distance ring and
// output.  Streams never share anything, so a batch partitions freely over wavefronts and over GPUs.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// flags
#define BROTLI_AMD_FLAG_LARGE_WINDOW 1u  // accept the 14-bit large-window header (state.rs:394, ffi/mod.rs:127)
#define BROTLI_AMD_FLAG_NO_CANNY 2u      // BROTLI_DECODER_PARAM_DISABLE_RING_BUFFER_REALLOCATION (ffi/mod.rs:167-169)
#define BROTLI_AMD_FLAG_RESUME 4u        // start from the metablock boundary stored in `resume`
#define BROTLI_AMD_FLAG_NO_SPILL 8u      // a metablock whose tables do not fit the LDS arena ends the decode with result
                                         // BROTLI_AMD_RESULT_RETRY_ARENA at the boundary before it (the host then
                                         // resumes the stream in a launch with a larger arena) instead of spilling
#define BROTLI_AMD_FLAG_ENGINE_ONLY 32u  // set by the host where sixteen-wave blocks (one per CU) serve more streams than
                                         // there are CUs: a large metablock the command engine cannot take (literals that
                                         // depend on context) ends the decode the same way, and the stream continues in a
                                         // launch of small blocks, several to a CU -- one wave decodes such a metablock
                                         // wherever it runs, so what counts for it is streams in flight
#define BROTLI_AMD_FLAG_PROBE 128u       // nothing is decoded: the stream's header is read up to the literal context map of its first compressed
                                         // metablock, and the status says what kind of stream this is (result BROTLI_AMD_RESULT_PROBE,
                                         // engine_commands = bit 0: such a metablock exists, bit 1: its literals do not depend on context,
                                         // bit 2: it is large enough for a command engine): the host picks the launch shape by it
#define BROTLI_AMD_FLAG_DEFER 256u       // the stream is not this launch's: reported as BROTLI_AMD_RESULT_RETRY_ARENA at once (it is decoded, from
                                         // its first byte, in the launch of small blocks that follows)
#define BROTLI_AMD_RESULT_RETRY_ARENA 4  // (never reaches the caller of the C ABI)
#define BROTLI_AMD_RESULT_PROBE 5        // (never reaches the caller of the C ABI)
#define BROTLI_AMD_GANG_POOL_FLAG 0x100u  // queue[2] of a POOL launch (with 8 in the low bits: the blocks a stream may gather); a gang launch: 2, 4, 8 or 16 blocks a stream
#define BROTLI_AMD_GANG_CTL_BYTES 50176u // a gang's control block in memory, one per stream of a gang launch (brotli_kernels.hip: GC_*): queue[2] = blocks of a
                                         // gang (0 or 1: none), queue[4], queue[5] = the address of the first stream's block, the host zeroes them before the launch
#define BROTLI_AMD_SPEC_SCRATCH 65536u   // bytes at the end of each block's global scratch that the helper waves use for
                                         // speculatively decoded literals (the table arena is the part in front of it)

// State at a metablock boundary: everything that survives from one metablock to the next in the
// reference (state.rs:422-450 resets the rest).  The kernel stores it after every completed metablock;
// the streaming ABI feeds it back so that a stream delivered in pieces is decoded metablock by metablock
// instead of from byte 0 (the device analogue of the reference's resumable state machine).
typedef struct BrotliAmdResume {
  uint64_t bit_pos;       // absolute bit position in the compressed stream
  uint64_t out_pos;       // bytes produced so far
  int32_t dist_rb[4];     // last four distances, most recent first (state.rs:295-296 as a shift register)
  int32_t dist_rb_idx;    // unused (always 0): the ring is stored already rotated
  uint32_t window_bits;   // 0 = stream header not parsed yet
  uint32_t large_window;  // stream carries the large-window header
  uint32_t rb_size_log2;  // emulated ring size (0 = not allocated yet), decode.rs:1808-1871
  uint32_t is_last_done;  // last metablock completed (stream finished up to the final padding)
  uint32_t reserved;
  // A command boundary inside the metablock that starts at bit_pos (mid_valid != 0): where a launch that ran out of
  // input had got to.  The next launch parses that metablock's header again (the prefix codes are not kept) and goes
  // on from here instead of from the metablock's first command, so a stream fed in small pieces costs what its bytes
  // cost, not (pieces x metablock).  The reference keeps the same things across calls in its state (state.rs:255-330:
  // block lengths and types, distance ring, meta_block_remaining_len, the bit reader).
  uint32_t mid_valid;
  int32_t mid_mlen;        // bytes of the metablock still to be produced
  uint64_t mid_bit_pos;    // bit position of the next command
  uint64_t mid_out_pos;    // bytes produced in front of it
  uint32_t mid_bl[3];      // literals / commands / distances left in the current blocks
  uint32_t mid_types[6];   // second last and last block type, per category (state.rs:429-435)
  int32_t mid_dist_rb[4];  // last four distances, most recent first
  uint32_t mid_reserved;
} BrotliAmdResume;

typedef struct BrotliAmdStreamDesc {
  const uint8_t* in;      // device pointer, any alignment
  uint64_t in_size;
  uint8_t* out;           // device pointer
  uint64_t out_cap;
  uint32_t flags;
  uint32_t reserved;
  BrotliAmdResume resume; // only read when BROTLI_AMD_FLAG_RESUME is set
} BrotliAmdStreamDesc;

typedef struct BrotliAmdStreamStatus {
  int32_t result;         // BrotliResult: 0 error, 1 success, 2 needs more input, 3 needs more output
  int32_t error_code;     // BrotliDecoderErrorCode (state.rs:22-65)
  uint64_t decoded_size;  // bytes the reference would have delivered to the caller (lib.rs:466)
  uint64_t consumed;      // input bytes consumed
  uint64_t produced;      // bytes written to `out` (>= decoded_size on errors)
  uint32_t num_metablocks;
  uint32_t spilled_metablocks;  // metablocks whose tables did not fit the LDS part of the arena
  uint64_t num_commands;
  BrotliAmdResume resume; // last completed metablock boundary
  // What the reference's three allocators would have been asked for (BrotliDecoderDecompressPrealloc accounts its scratch
  // slices with these): prefix codes alive in one metablock at most (each BROTLI_HUFFMAN_MAX_TABLE_SIZE HuffmanCode cells
  // and one u32, huffman/mod.rs:61-72), bytes of context modes and maps alive in one metablock at most (decode.rs:1295,
  // 3155), the emulated ring buffer (decode.rs:1843-1855), and whether any compressed metablock was started (the block
  // type and block length trees, decode.rs:2958-2969)
  uint32_t peak_trees, peak_map_bytes, any_compressed;
  uint32_t engine_commands;  // of num_commands, how many a command engine took (blocks of sixteen waves): lets tests prove which path ran
  uint64_t ring_bytes;
} BrotliAmdStreamStatus;

#ifdef __cplusplus
}
#endif
