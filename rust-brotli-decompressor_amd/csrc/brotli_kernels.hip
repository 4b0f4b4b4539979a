// brotli_kernels.hip -- the Brotli decode hot path as hand-written HIP for gfx950 (MI355X / CDNA4).
//
// What this replaces in the reference (paths relative to /root/reference):
//   src/decode.rs:2330-2744   ProcessCommandsInternal (command loop)          -> lean_commands() (common case) +
//                                                                                process_commands() (every check)
//   src/decode.rs:378-500     DecodeSymbol / ReadSymbol                       -> read_symbol(), LITERAL_BATCH_ASM
//   src/decode.rs:2017-2189   distance / command readers                      -> inline in both command functions
//   src/decode.rs:1469-1524   block switches                                  -> block_switch()
//   src/decode.rs:1754-1806   CopyUncompressedBlockToOutput                   -> copy_uncompressed()
//   src/bit_reader/mod.rs     64-bit window + BrotliFillBitWindow             -> struct BitReader (LDS-DMA input ring)
//   src/huffman/mod.rs        table builders                                  -> build_tree() (lane-parallel)
//   src/transform.rs:720-795  TransformDictionaryWord                          -> dictionary_word_bytes()
//   src/decode.rs:152-372, 516-1465, 2921-3288  stream/metablock headers       -> decode_stream() (must run on device:
//                             a metablock's compressed extent is only known after decoding it)
//
// Execution model: ONE DECODING WAVEFRONT PER STREAM (wave 0 of a block of one, four or eight waves), fully fused; the other waves of the block
// are helpers that decode chunks of long literal runs speculatively (spec_rounds / spec_chunk / helper_wave) and
// sleep otherwise.  The entropy decode of a Brotli stream is a serial
// dependent chain, so all 64 lanes of the wave execute it uniformly (values live in SGPRs; table entries come back
// from LDS through v_readfirstlane) and the lanes are used for everything that is data parallel inside one stream:
//   * input: 256-byte pieces of the compressed stream go straight into an LDS ring (global_load_lds); the wave takes
//     a 64-dword register window out of it and refills its 64-bit bit buffer with v_readlane -- no memory latency on
//     the serial chain;
//   * literal runs with one prefix code: all 64 bit offsets of the next 64 bits are decoded at once (gathered table
//     lookup), a scalar walk over the code lengths finds the real symbol boundaries, one masked store writes them;
//   * small LUTs (insert/copy code ranges, block-length code, the 32-entry code-length code, the context -> tree map
//     of the current literal block type) live one entry per lane in VGPRs and are indexed with v_readlane;
//   * Huffman tables are built lane-parallel (ballot counting sort + parallel replicate) into an LDS arena;
//     objects that do not fit the LDS arena spill to a per-block global scratch area (never straddling);
//   * literal runs of >= 768 literals: rounds of one 4096-bit chunk per wave, all but the first decoded by the helper waves from
//     a bit that need not start a literal -- prefix codes re-synchronise, and the decoding wave walks the true chain into
//     each chunk only until it meets the helper's; a chunk that does not fall in is not used;
//   * LZ77 copies, dictionary words and stored metablocks are moved by all 64 lanes (16 bytes per lane and step where
//     source and destination are far enough apart), overlapping copies as pattern fills.
// What bounds it is the instruction issue rate of one wave (about one instruction per 8 clocks): see DESIGN.md.
// Blocks are persistent: each pulls stream indices from an atomic queue until the batch is empty.
// Blocks of sixteen waves put all of them on their stream's commands where the metablock allows it (the command engines:
// brotli_scan_engine.h, brotli_path_engine.h), and -- round 5 -- batches of few streams put SEVERAL BLOCKS on a stream: gangs of
// blocks that take the path engine's regions in turns, dealt at the launch or, in a pool, joining the streams that last (GC_*
// below, the kernel at the end of this file).
//
// Roofline that bounds it: HBM traffic is (compressed bytes read + decompressed bytes written); there is no
// dense contraction, MFMA is not used.  See DESIGN.md.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <mutex>

#include "brotli_device_abi.h"
#ifndef BROTLI_AMD_DECODER_PRIO
#define BROTLI_AMD_DECODER_PRIO 1   // s_setprio of a block's decoding wave (its helper waves stay at 0)
#endif
#define BROTLI_TABLE_QUAL __constant__ const
#include "brotli_tables_gen.h"

namespace {

// ---- result / error codes (src/state.rs:22-65) ----
constexpr int E_SUCCESS = 1, E_NEEDS_MORE_INPUT = 2, E_NEEDS_MORE_OUTPUT = 3;
constexpr int E_EXUBERANT_NIBBLE = -1, E_RESERVED = -2, E_EXUBERANT_META_NIBBLE = -3, E_SIMPLE_HUFFMAN_ALPHABET = -4,
              E_SIMPLE_HUFFMAN_SAME = -5, E_CL_SPACE = -6, E_HUFFMAN_SPACE = -7, E_CONTEXT_MAP_REPEAT = -8,
              E_BLOCK_LENGTH_1 = -9, E_BLOCK_LENGTH_2 = -10, E_TRANSFORM = -11, E_DICTIONARY = -12, E_WINDOW_BITS = -13,
              E_PADDING_2 = -15, E_DISTANCE = -16, E_UNREACHABLE = -31;
constexpr int E_RETRY_ARENA = 100;  // internal: see BROTLI_AMD_FLAG_NO_SPILL
constexpr int E_PROBE = 101;        // internal: see BROTLI_AMD_FLAG_PROBE
#ifndef BROTLI_AMD_PROBE_LONG_PERCENT
#define BROTLI_AMD_PROBE_LONG_PERCENT 75
#endif
// A stream is the command engines' kind by its commands where, by SOME command code's own probabilities, at least this share of the code's BYTES comes from commands
// that insert or copy more than 63 bytes; otherwise its commands are "short" (the probe's bit 3).  Text: 4 - 11 %; map tiles 33 - 60 %: the engines take them at a
// quarter of what one-wave blocks do; the metric's make-up and the survey's: one of their codes 97 - 99 %.
constexpr uint32_t PROBE_LONG_PERCENT = BROTLI_AMD_PROBE_LONG_PERCENT;

// ---- small constant tables (RFC 7932 sections 3.5, 4, 5, 6) ----
// (decode.rs:801-853's three small tables -- kCodeLengthCodeOrder = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15},
// kCodeLengthPrefixLength = {2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4}, kCodeLengthPrefixValue = {0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2,
// 0, 4, 3, 5} -- are immediates where read_huffman_code uses them: a nibble or five bits an entry)
// per-lane LUT image: lanes 0..23 insert code (base | extra<<16), lanes 32..55 copy code, see lane_lut_init()
__constant__ const uint16_t kInsBase[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
__constant__ const uint8_t kInsExtra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
__constant__ const uint16_t kCopyBase[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
__constant__ const uint8_t kCopyExtra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
__constant__ const uint16_t kBlockLenBase[26] = {1, 5, 9, 13, 17, 25, 33, 41, 49, 65, 81, 97, 113, 145, 177, 209, 241, 305, 369, 497, 753, 1265, 2289, 4337, 8433, 16625};
__constant__ const uint8_t kBlockLenExtra[26] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24};
// upper bound of a 2-level table (8-bit root, max code length 15) for an alphabet of 32*i symbols
__constant__ const uint16_t kMaxTableSize[37] = {256, 402, 436, 468, 500, 534, 566, 598, 630, 662, 694, 726, 758, 790, 822, 854, 886, 920, 952,
                                                  984, 1016, 1048, 1080, 1112, 1144, 1176, 1208, 1240, 1272, 1304, 1336, 1368, 1400, 1432,
                                                  1464, 1496, 1528};

constexpr int ROOT_BITS = 8;
// the LDS part of the table arena as a cache of the trees in use (run_commands): a literal tree (630 entries at most), a command
// tree (1080), four distance trees (920 each: 520 symbols), sixteen-byte steps
constexpr uint32_t TREE_CACHE_LIT = 0, TREE_CACHE_LIT_BYTES = 1264, TREE_CACHE_CMD = 1280, TREE_CACHE_CMD_BYTES = 2160, TREE_CACHE_DIST = 3456, TREE_CACHE_DIST_BYTES = 1856,
                   TREE_CACHE_BYTES = TREE_CACHE_DIST + 4u * TREE_CACHE_DIST_BYTES,
                   // ... and, where literals depend on context, the trees of ONE literal block type: as many slots as the LDS part has room for (its 64
                   // contexts name 64 different trees at most), eight at least
                   TREE_CACHE_LIT_STRIDE = 1280, TREE_CACHE_CTX_BYTES = TREE_CACHE_BYTES + 8u * TREE_CACHE_LIT_STRIDE;
__device__ __forceinline__ uint32_t tree_cache_lit_slots(uint32_t lds_bytes) { const uint32_t n = (lds_bytes - TREE_CACHE_BYTES) / TREE_CACHE_LIT_STRIDE; return n < 64u ? n : 64u; }
constexpr uint32_t MAX_ALPHABET = 1152;  // 16 + 120 + (62 << 4) = 1128 for large-window distance codes

// ---- fixed LDS carve (bytes); the arena follows ----
constexpr uint32_t LDS_CTX_LUT = 0;                         // 2048: literal context lookup (all 4 modes)
constexpr uint32_t LDS_LENGTHS = LDS_CTX_LUT + 2048;        // 1152: code length per symbol while a tree is read
constexpr uint32_t LDS_SUBDEPTH = LDS_LENGTHS + MAX_ALPHABET;  // 256: depth of the 2nd-level table under each 8-bit prefix
constexpr uint32_t LDS_SUBBASE = LDS_SUBDEPTH + 256;        // 512: its base offset (u16)
constexpr uint32_t LDS_LENINFO = LDS_SUBBASE + 512;         // 128: spare
constexpr uint32_t LDS_WORD = LDS_LENINFO + 128;            // 128: dictionary word staging
constexpr uint32_t LDS_MTF = LDS_WORD + 128;                // 256: inverse move-to-front list
constexpr uint32_t LDS_INWIN = LDS_MTF + 256;               // 1024: compressed-input ring, four 256-byte halves filled by LDS-DMA
constexpr uint32_t LDS_BR = LDS_INWIN + 1024;               // 32: what the bit reader needs only when it moves its window (BitReader)
constexpr uint32_t LDS_HOT = LDS_BR + 32;                   // 96: what the command loop needs only at block switches (process_commands)
constexpr uint32_t LDS_LEAN = LDS_HOT + 96;                 // 192: state handed between process_commands and lean_commands
constexpr uint32_t LDS_LEANWIN = LDS_LEAN + 192;            // 256: the reader's register window, handed over with the state
constexpr uint32_t LDS_HCTL = LDS_LEANWIN + 256;            // 80: mailbox between the decoding wave and its helper waves (round-wide words)
constexpr uint32_t LDS_FIXED = LDS_HCTL + 80;               // = 6224, 16-byte aligned
// Blocks launched with helper waves have, behind the table arena, one slot per wave of the block with what the waves
// of a round leave for each other (slot w at mailbox word HC_BASE + w * HL_SLOT):
constexpr uint32_t SPEC_WINDOWS = 64;                        // windows of 64 bits per chunk
constexpr uint32_t SPEC_FIRST = 4;                           // windows at the start of a chunk the decoding wave may walk itself
constexpr uint32_t SPEC_MAX_WAVES = 16;                      // waves of a round at most (one decoding, fifteen helpers: the blocks of sixteen waves)
constexpr uint32_t HL_CTL = 0;                               // 64: the wave's mailbox words (HW_*)
constexpr uint32_t HL_MASK = HL_CTL + 64;                    // SPEC_WINDOWS x 8: per window of the chunk, which bit offsets start a literal
constexpr uint32_t HL_CUM = HL_MASK + SPEC_WINDOWS * 8;      // SPEC_WINDOWS x 4: literals in the chunk's windows before this one
constexpr uint32_t HL_FIRST = HL_CUM + SPEC_WINDOWS * 4;     // SPEC_FIRST x 128: code length and symbol at every offset of the chunk's first windows
constexpr uint32_t HL_SLOT = HL_FIRST + SPEC_FIRST * 128;    // = 1344
constexpr uint32_t SPEC_SLOT_BYTES = SPEC_WINDOWS * 64u;     // a chunk's literals at most (scratch slot of a wave, in HBM)
static_assert(SPEC_MAX_WAVES * SPEC_SLOT_BYTES <= BROTLI_AMD_SPEC_SCRATCH, "one scratch slot of a chunk's literals per wave");
static_assert(HL_SLOT % 16 == 0, "slots keep the 8-byte alignment of their masks");
static_assert(LDS_FIXED % 16 == 0, "arena base must stay 16-byte aligned");

// All LDS traffic goes through this file-scope array so that every access is a DS instruction (address space 3)
// and every global access through address-space-1 pointers (global_load/global_store).  Generic pointers would
// become FLAT instructions, and the hardware does not order a FLAT access to LDS against a DS access.
// The kernel has no static LDS, so its dynamic LDS starts at LDS address 0 and everything below addresses LDS
// absolutely: `g_smem` is a macro for address-space-3 address 0, not a symbol.  (A reference to an
// `extern __shared__` symbol from a non-inlined device function makes hipcc look its address up in a table in
// constant memory -- an s_load in front of table lookups in the command loop.)  The kernel checks the assumption.
extern __shared__ __attribute__((aligned(16))) uint8_t g_dynamic_lds[];
typedef __attribute__((address_space(3))) uint8_t lds_u8;
#define g_smem (reinterpret_cast<lds_u8*>(0u))
typedef __attribute__((address_space(1))) uint8_t gu8;
typedef __attribute__((address_space(1))) uint16_t gu16;
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) const uint8_t gcu8;
typedef __attribute__((address_space(1))) const uint32_t gcu32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
template <typename T, typename U>
__device__ __forceinline__ T* as_global(U* p) { return (T*)(uintptr_t)p; }

__device__ __forceinline__ uint32_t lds_ld8(uint32_t off) { return g_smem[off]; }
__device__ __forceinline__ uint32_t lds_ld16(uint32_t off) { return *reinterpret_cast<__attribute__((address_space(3))) const uint16_t*>(&g_smem[off]); }
__device__ __forceinline__ uint32_t lds_ld32(uint32_t off) { return *reinterpret_cast<__attribute__((address_space(3))) const uint32_t*>(&g_smem[off]); }
__device__ __forceinline__ void lds_st8(uint32_t off, uint32_t v) { g_smem[off] = (uint8_t)v; }
__device__ __forceinline__ void lds_st16(uint32_t off, uint32_t v) { *reinterpret_cast<__attribute__((address_space(3))) uint16_t*>(&g_smem[off]) = (uint16_t)v; }
__device__ __forceinline__ void lds_st32(uint32_t off, uint32_t v) { *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(&g_smem[off]) = v; }
// LDS operations of one wave execute in order; this only stops the compiler from moving accesses across it
__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Values that are the same in every lane but come out of per-lane storage (VGPRs, private memory) are passed
// through v_readfirstlane so that the compiler keeps them, and everything computed from them, in SGPRs and
// branches on them with scalar branches instead of exec masks.
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int32_t rfl(int32_t v) { return (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)v); }
__device__ __forceinline__ uint64_t rfl(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);  // the builtin returns int: no sign extension
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}
template <typename T>
__device__ __forceinline__ T* rfl_ptr(T* p) { return (T*)(uintptr_t)rfl((uint64_t)(uintptr_t)p); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t mask_bits(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }
__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t n) { return n ? (__brev(v) >> (32 - n)) : 0; }

// =========================================== bit reader ===========================================
// Replaces src/bit_reader/mod.rs: same values, different mechanics.  The compressed stream is fetched 256 bytes
// (one dword per lane) at a time straight into a four-slot ring in LDS (global_load_lds: no register destination,
// nothing for the compiler to wait on in the decode loops); the wave takes its 64-dword register window `cur`
// out of the ring at whatever dword the reader has reached, and only there waits for the transfers -- which were
// issued one or two windows earlier.
struct BitReader {
  // Per-stream constants that only matter when the window moves or the end of the input is near live in LDS
  // (written by set_input), not in registers: the command loop is short of SGPRs, not of LDS reads on cold paths.
  //   LDS_BR + 0: stream start rounded down to 4 bytes (u64)   + 8: dwords that contain stream bytes
  //   + 12: valid bytes of the last dword (mask)   + 16: 8 * (stream start & 3)   + 24: 8 * in_size (u64)
  __device__ __forceinline__ static gcu32* base() {
    return (gcu32*)(uintptr_t)rfl(*reinterpret_cast<__attribute__((address_space(3))) const uint64_t*>(&g_smem[LDS_BR]));
  }
  __device__ __forceinline__ static uint32_t n_dw() { return rfl(lds_ld32(LDS_BR + 8)); }
  __device__ __forceinline__ static uint32_t tail_mask() { return rfl(lds_ld32(LDS_BR + 12)); }
  __device__ __forceinline__ static uint32_t skip_bits() { return rfl(lds_ld32(LDS_BR + 16)); }
  __device__ __forceinline__ static uint64_t total_bits() { return rfl(*reinterpret_cast<__attribute__((address_space(3))) const uint64_t*>(&g_smem[LDS_BR + 24])); }
  uint32_t end_dw;       // index of the dword that contains the first bit after the stream
  uint32_t cur;          // per-lane dword of the window [chunk_base, chunk_base + 64)
  uint32_t chunk_base;   // uniform
  uint32_t issued_half;  // uniform: 64-dword pieces of the stream below this index have been requested into the ring
  uint32_t next_dw;      // uniform: next dword to shift into buf
  uint64_t buf;          // uniform
  uint32_t cnt;          // uniform: valid bits in buf

  // after a copy out of private memory: re-establish that everything but the window register is uniform
  __device__ __forceinline__ void uniformize() {
    end_dw = rfl(end_dw);
    chunk_base = rfl(chunk_base); issued_half = rfl(issued_half); next_dw = rfl(next_dw); buf = rfl(buf); cnt = rfl(cnt);
  }
  // bit reader over [in, in + in_size)
  __device__ __forceinline__ void set_input(uint64_t addr, uint64_t in_size) {
    uint32_t mis = (uint32_t)(addr & 3u);
    uint64_t span = (uint64_t)mis + in_size;
    uint32_t tail = (uint32_t)(span & 3u);
    lds_sync();
    if (lane_id() == 0) {
      *reinterpret_cast<__attribute__((address_space(3))) uint64_t*>(&g_smem[LDS_BR]) = addr - mis;
      lds_st32(LDS_BR + 8, (uint32_t)((span + 3) >> 2));
      lds_st32(LDS_BR + 12, tail ? ((1u << (tail * 8)) - 1u) : 0xFFFFFFFFu);
      lds_st32(LDS_BR + 16, mis * 8);
      *reinterpret_cast<__attribute__((address_space(3))) uint64_t*>(&g_smem[LDS_BR + 24]) = in_size * 8;
    }
    lds_sync();
    end_dw = (uint32_t)((in_size * 8 + mis * 8) >> 5);
  }
  // request dwords [64 h, 64 h + 64) of the stream into ring slot h & 3 (asynchronous; counted by vmcnt only)
  __device__ __forceinline__ void dma_half(uint32_t h) const {
    uint32_t i = (h << 6) + lane_id();
    uint32_t lds_dst = LDS_INWIN + ((h & 3u) << 8);
    uint32_t ndw = n_dw();
    if (__ballot(i < ndw)) {  // lanes past the end of the stream request nothing
      gcu32* src = base() + i;
      uint32_t keep_m0, t; uint64_t keep_exec;
      // (the operands are made uniform inside the statement: in code whose control flow the compiler takes for
      // divergent it would hand over VGPRs for "s" operands)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\tv_readfirstlane_b32 %2, %4\n\tv_cmp_gt_u32 vcc, %5, %6\n\t"
                   "s_mov_b32 m0, %2\n\ts_mov_b64 exec, vcc\n\t"
                   "global_load_lds_dword %3, off\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0), "=&s"(keep_exec), "=&s"(t) : "v"(src), "v"(lds_dst), "v"(ndw), "v"(i) : "memory", "vcc");
    }
  }
  // move the register window to [next_dw, next_dw + 64)
  __device__ __forceinline__ void rebase() {
    const uint32_t a = next_dw >> 6;
    while (issued_half <= a + 1u) { dma_half(issued_half); issued_half++; }  // normally requested long ago
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t j = next_dw + lane_id();
    uint32_t v = lds_ld32(LDS_INWIN + ((j & 255u) << 2));
    const uint32_t ndw = n_dw();
    v = j < ndw ? v : 0u;                        // beyond the stream: zero bits
    if (j == ndw - 1u) v &= tail_mask();          // bytes of the last dword that lie beyond the stream read as zero
    cur = v;
    chunk_base = next_dw;
    while (issued_half <= a + 2u) { dma_half(issued_half); issued_half++; }  // slots of pieces below a - 1 are dead
  }
  // ... to [next_dw - 2, next_dw + 62): the two dwords in front of next_dw may still be in buf, and a reader that goes by its POSITION
  // (the hand-written run of lean_rec_commands: bits into the window) wants them in the window too.  (Piece a - 1 of the ring is still
  // there: see above.)
  __device__ __forceinline__ void rebase_back2() {
    const uint32_t a = next_dw >> 6;
    while (issued_half <= a + 1u) { dma_half(issued_half); issued_half++; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t first = next_dw >= 2u ? next_dw - 2u : 0u;
    uint32_t j = first + lane_id();
    uint32_t v = lds_ld32(LDS_INWIN + ((j & 255u) << 2));
    const uint32_t ndw = n_dw();
    v = j < ndw ? v : 0u;
    if (j == ndw - 1u) v &= tail_mask();
    cur = v;
    chunk_base = first;
    while (issued_half <= a + 2u) { dma_half(issued_half); issued_half++; }
  }
  __device__ __forceinline__ void seek(uint64_t bit_pos) {
    uint64_t abs = bit_pos + skip_bits();
    uint32_t dw = (uint32_t)(abs >> 5);
    next_dw = dw;
    issued_half = dw >> 6;
    rebase();
    buf = 0; cnt = 0;
    pull();
    uint32_t r = (uint32_t)(abs & 31);
    buf >>= r; cnt -= r;
  }
  // request the pieces a reader will need after a jump to about dword `dw` (the reader is not used until seek_ahead)
  __device__ __forceinline__ void request_ahead(uint32_t dw) {
    const uint32_t a = dw >> 6;
    dma_half(a); dma_half(a + 1u); dma_half(a + 2u);
    issued_half = a + 3u;
  }
  // seek to a position whose pieces request_ahead(dw) has asked for, if it did; a plain seek otherwise
  __device__ __forceinline__ void seek_ahead(uint64_t bit_pos) {
    uint64_t abs = bit_pos + skip_bits();
    uint32_t dw = (uint32_t)(abs >> 5);
    const uint32_t a = dw >> 6;
    next_dw = dw;
    if (!(a + 3u >= issued_half && a + 2u <= issued_half)) issued_half = a;  // pieces a, a + 1 are not both there
    rebase();
    buf = 0; cnt = 0;
    pull();
    uint32_t r = (uint32_t)(abs & 31);
    buf >>= r; cnt -= r;
  }
  __device__ __forceinline__ void pull() {  // shift one more dword in (cnt <= 32 on entry)
    uint32_t idx = next_dw - chunk_base;
    if (idx >= 64) { rebase(); idx = 0; }
    uint32_t dw = rdlane(cur, idx);
    buf |= (uint64_t)dw << cnt;
    cnt += 32;
    next_dw++;
  }
  __device__ __forceinline__ void need32() { if (cnt < 32) pull(); }
  __device__ __forceinline__ uint32_t peek32() { need32(); return (uint32_t)buf; }
  __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
  __device__ __forceinline__ uint32_t read(uint32_t n) {  // n in 0..32
    need32();
    uint32_t v = (uint32_t)buf & mask_bits(n);
    if (n == 32) { buf >>= 16; buf >>= 16; } else buf >>= n;
    cnt -= n;
    return v;
  }
  __device__ __forceinline__ uint32_t read24(uint32_t n) {  // n in 0..24: no n == 32 case to care about
    need32();
    uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
    buf >>= n;
    cnt -= n;
    return v;
  }
  // the window holds at least `k` dwords that have not been shifted into buf yet (k <= 4)
  __device__ __forceinline__ void ensure_dwords(uint32_t k) { if (next_dw - chunk_base > 64u - k) rebase(); }
  // the two dwords after the ones already shifted into buf, without consuming them (after ensure_dwords(2))
  __device__ __forceinline__ uint64_t peek64() const {
    uint32_t idx = next_dw - chunk_base;
    return (uint64_t)rdlane(cur, idx) | ((uint64_t)rdlane(cur, idx + 1u) << 32);
  }
  // 128-bit view of the stream from the current position: at least cnt + 64 >= 96 valid bits (needs cnt >= 32)
  __device__ __forceinline__ void window128(uint64_t& lo, uint64_t& hi) {
    ensure_dwords(2);
    uint64_t e = peek64();
    lo = buf | (e << cnt);         // 32 <= cnt <= 63
    hi = e >> (64 - cnt);
  }
  __device__ __forceinline__ void advance(uint32_t n) {  // (a short way: the window follows piece by piece)
    if (n <= cnt) { drop(n); return; }
    n -= cnt; buf = 0; cnt = 0;
    next_dw += n >> 5; n &= 31;
    if (n) { pull(); drop(n); }
  }
  __device__ __forceinline__ uint64_t pos() const { return (uint64_t)next_dw * 32 - cnt - skip_bits(); }
  // bits consumed beyond the end of the input?  Only possible once the dword that holds the end has been pulled.
  __device__ __forceinline__ bool over() const { return next_dw > end_dw && pos() > total_bits(); }
};

// ============================================= arena =============================================
// Per-metablock tables.  Offsets below lds_limit are LDS, the rest is this block's global scratch.
struct Arena {
  gu8* glb;            // this block's global scratch, addressed with the same offsets as the LDS part
  uint32_t lds_limit;  // arena bytes that live in LDS (at g_smem + LDS_FIXED)
  uint32_t top;        // hot objects (prefix-code tables of literals, commands, distances) grow up from 0
  uint32_t cold;       // cold objects (what only block switches read: block-type/length trees, context maps, tree
                       // groups) grow down from the end of the global scratch and never take LDS

  __device__ __forceinline__ void uniformize() {
    glb = (gu8*)(uintptr_t)rfl((uint64_t)(uintptr_t)glb);
    lds_limit = rfl(lds_limit); top = rfl(top); cold = rfl(cold);
  }
  __device__ __forceinline__ uint32_t alloc(uint32_t max_bytes) {  // object never straddles the LDS/global split
    uint32_t off = (top + 3u) & ~3u;
    if (off < lds_limit && off + max_bytes > lds_limit) off = lds_limit;
    top = off + max_bytes;
    return off;
  }
  __device__ __forceinline__ uint32_t alloc_cold(uint32_t bytes) {
    cold = (cold - bytes) & ~3u;
    return cold;
  }
  __device__ __forceinline__ void shrink_to(uint32_t off_end) { top = off_end; }
  // uniform loads.  LDS_ONLY = the caller knows the object is in the LDS part: the load is a plain ds_read and
  // the wait in front of v_readfirstlane is lgkmcnt only (the two-way form must also wait for vmcnt, i.e. for
  // every global store the wave still has in flight)
  template <bool LDS_ONLY = false>
  __device__ __forceinline__ uint32_t ld16(uint32_t off) const {
    uint32_t v;
    if (LDS_ONLY || off < lds_limit) v = lds_ld16(LDS_FIXED + off); else v = *reinterpret_cast<gu16*>(glb + off);
    return rfl(v);
  }
  template <bool LDS_ONLY = false>
  __device__ __forceinline__ uint32_t ld8(uint32_t off) const {
    uint32_t v;
    if (LDS_ONLY || off < lds_limit) v = lds_ld8(LDS_FIXED + off); else v = glb[off];
    return rfl(v);
  }
  template <bool LDS_ONLY = false>
  __device__ __forceinline__ uint32_t ld32(uint32_t off) const {
    uint32_t v;
    if (LDS_ONLY || off < lds_limit) v = lds_ld32(LDS_FIXED + off); else v = *reinterpret_cast<gu32*>(glb + off);
    return rfl(v);
  }
  // per-lane accesses (every active lane uses its own offset)
  template <bool LDS_ONLY = false>
  __device__ __forceinline__ uint32_t ld8_lane(uint32_t off) const { return (LDS_ONLY || off < lds_limit) ? lds_ld8(LDS_FIXED + off) : (uint32_t)glb[off]; }
  __device__ __forceinline__ void st16_lane(uint32_t off, uint32_t v) const {
    if (off < lds_limit) lds_st16(LDS_FIXED + off, v); else *reinterpret_cast<gu16*>(glb + off) = (uint16_t)v;
  }
  __device__ __forceinline__ void st8_lane(uint32_t off, uint32_t v) const {
    if (off < lds_limit) lds_st8(LDS_FIXED + off, v); else glb[off] = (uint8_t)v;
  }
  // uniform stores (one lane)
  __device__ __forceinline__ void st16(uint32_t off, uint32_t v) const { if (lane_id() == 0) st16_lane(off, v); }
  __device__ __forceinline__ void st8(uint32_t off, uint32_t v) const { if (lane_id() == 0) st8_lane(off, v); }
  __device__ __forceinline__ void st32(uint32_t off, uint32_t v) const {
    if (lane_id() == 0) { if (off < lds_limit) lds_st32(LDS_FIXED + off, v); else *reinterpret_cast<gu32*>(glb + off) = v; }
  }
};

// table entry: (value << 4) | len.  Root entry with len > 8: value = offset of the 2nd-level table from the
// tree base (entries), len - 8 = its depth.  2nd-level entry: len = code length - 8.
template <bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t read_symbol(BitReader& br, const Arena& a, uint32_t tree) {
  uint32_t bits = br.peek32();
  uint32_t e = a.ld16<LDS_ONLY>(tree + ((bits & 0xFFu) << 1));
  uint32_t len = e & 15u;
  if (len > ROOT_BITS) {
    uint32_t idx = (e >> 4) + ((bits >> ROOT_BITS) & mask_bits(len - ROOT_BITS));
    e = a.ld16<LDS_ONLY>(tree + (idx << 1));
    len = ROOT_BITS + (e & 15u);
  }
  br.drop(len);
  return e >> 4;
}

// ========================================== decoder state ==========================================
// Everything one stream carries (reference: BrotliState, src/state.rs:156-278).  The object itself may live
// in private memory (its address is handed to the one non-inlined helper); every function works on register
// copies of the parts it touches and stores them back on exit.
struct Stream {
  BitReader br;
  Arena ar;
  uint32_t ar_end;         // size of the block's global scratch = where the cold objects start growing down
  gu8* out;
  uint64_t out_cap;
  uint64_t P;              // bytes produced
  uint64_t next_boundary;  // next ring-buffer flush point (multiple of rb_size); 0 = ring not allocated
  uint64_t rb_size;
  gcu8* dict;
  gcu8* in_bytes;
  uint32_t flags;
  uint32_t window_bits, large_window;
  int32_t max_backward;
  int32_t dist_rb0, dist_rb1, dist_rb2, dist_rb3;
  int32_t dist_rb_idx;
  // metablock
  uint32_t is_last, is_uncompressed, is_metadata;
  int32_t mlen;
  uint32_t nbt0, nbt1, nbt2;           // number of block types per category
  uint32_t bl0, bl1, bl2;              // remaining block length per category
  uint32_t bt_tree0, bt_tree1, bt_tree2, bl_tree0, bl_tree1, bl_tree2;  // arena offsets of block-type / block-length trees
  uint32_t postfix_bits, num_direct;
  uint32_t ctx_modes, ctx_map, dist_ctx_map;          // arena offsets
  uint32_t lit_trees, cmd_trees, dist_trees;          // arena offsets of u32 arrays of tree offsets
  uint32_t num_lit_trees, num_dist_trees;
  uint32_t lut_vgpr;       // per-lane: insert/copy code LUT image
  uint32_t bl_vgpr;        // per-lane: block length code LUT image
  uint32_t num_metablocks, num_spilled;
  uint64_t num_commands;
  uint32_t engine_commands;
  uint32_t general_engine;  // (see HotArgs)
  uint32_t peak_trees, peak_maps, any_compressed;  // what the reference's allocators would have been asked for (see BrotliAmdStreamStatus)
#ifdef BROTLI_AMD_PROFILE
  uint64_t prof[6];
#endif
};

#ifdef BROTLI_AMD_NO_COLD_UNIFORM
#define COLD_UNIFORMIZE(x)
#else
#define COLD_UNIFORMIZE(x) x
#endif
#ifdef BROTLI_AMD_TRACE
#define TRACE_STOP(br, e) do { if (lane_id() == 0) printf("stop line %d e=%d pos=%llu total=%llu\n", __LINE__, (int)(e), (unsigned long long)(br).pos(), (unsigned long long)(br).total_bits()); } while (0)
#else
#define TRACE_STOP(br, e) do { } while (0)
#endif
#define TRY(x) do { int _e = (x); if (_e != E_SUCCESS) return _e; } while (0)
#define NEED_INPUT(br) do { if ((br).over()) { TRACE_STOP(br, E_NEEDS_MORE_INPUT); return E_NEEDS_MORE_INPUT; } } while (0)
#define FAIL(br, e) do { TRACE_STOP(br, e); return (e); } while (0)

// ======================================= Huffman table build =======================================
// Lane-parallel construction of the 2-level table from LDS_LENGTHS[sym] (code length per symbol, 0 = unused).
// Canonical code = first_code[len] + rank of the symbol among the symbols of that length (rank from ballots
// over 64 symbols at a time).  Returns the table size in entries; the table is written at arena offset `tree`.
// Same codes as src/huffman/mod.rs:273-386, different (documented) entry layout.
struct CodeOfLane { uint32_t len, code; };
template <int LMIN>
__device__ __forceinline__ CodeOfLane lane_code(uint32_t L, const uint32_t (&first_code)[16], uint32_t (&run)[16], const uint32_t (&cnt)[16]) {
  const uint64_t lt = (1ull << lane_id()) - 1ull;
  CodeOfLane r; r.len = L; r.code = 0;
#pragma unroll
  for (int l = LMIN; l < 16; l++) {
    if (cnt[l] == 0u) continue;   // (no symbol of the alphabet has this length: uniform)
    uint64_t m = __ballot(L == (uint32_t)l);
    if (L == (uint32_t)l) r.code = first_code[l] + run[l] + (uint32_t)__popcll(m & lt);
    run[l] += (uint32_t)__popcll(m);
  }
  return r;
}

#ifdef BROTLI_AMD_PROFILE_HDR
__device__ unsigned long long g_hdr_prof[8];  // ticks: code-length code, symbol lengths, build_tree (3, 6, 7: its histogram, its first pass, the rest); counts: codes, symbols
#define BT_PROF(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && lane_id() == 0) g_hdr_prof[k] += t_ - bt_t; bt_t = t_; } while (0)
#else
#define BT_PROF(k) do { } while (0)
#endif
// (a function of its own -- and the reader of the code lengths below another: inlined into the kernel's body, at the limit of its 128
// registers a wave, their vectors spilt to scratch in the middle of the headers' loops -- 4096 tiny streams took half as long again)
__device__ __noinline__ uint32_t build_tree(const Arena& a, uint32_t tree, uint32_t n_sym) {
  const uint32_t lane = lane_id();
#ifdef BROTLI_AMD_PROFILE_HDR
  uint64_t bt_t = __builtin_amdgcn_s_memtime();
#endif
  // 1. histogram of code lengths (uniform, via ballots) and first code per length
  uint32_t cnt[16], first_code[16], run[16];
#pragma unroll
  for (int l = 0; l < 16; l++) { cnt[l] = 0; run[l] = 0; first_code[l] = 0; }
  for (uint32_t b = 0; b < n_sym; b += 64) {
    uint32_t sym = b + lane;
    uint32_t L = sym < n_sym ? lds_ld8(LDS_LENGTHS + sym) : 0;
#pragma unroll
    for (int l = 1; l < 16; l++) cnt[l] += (uint32_t)__popcll(__ballot(L == (uint32_t)l));
  }
  uint32_t max_len = 0;
  {
    uint32_t code = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) {
      first_code[l] = code;
      code = (code + cnt[l]) << 1;
      if (cnt[l]) max_len = l;
    }
  }
  BT_PROF(3);
  const bool two_level = max_len > ROOT_BITS;
  // The root's 256 entries by who OWNS them (round 5): in the order of the codes -- most significant bit first -- a code of L <= 8 bits
  // covers the 2^(8 - L) entries from code << (8 - L) on, and a complete code's ranges tile the root.  Every symbol notes its entry at
  // its range's first place (one store), the places in between take the last note in front of them (a scan), and the entry goes to
  // the table where the stream's bit order has it.  (Round 4: the symbol's lane wrote all its copies itself, 2^(8 - L) stores one after
  // the other -- a literal code whose best symbol has two bits kept its lane, and the wave, at it for sixty-four rounds.)
  constexpr uint32_t LDS_OWN = LDS_LENINFO;   // u16 x 256 (the spare room, the word staging and the move-to-front list: nobody's while a table is built)
  static_assert(LDS_MTF + 256u - LDS_LENINFO >= 512u, "the root's owners");
  for (uint32_t i = lane; i < 128; i += 64) lds_st32(LDS_OWN + 4u * i, 0u);
  if (two_level) for (uint32_t i = lane; i < 256; i += 64) lds_st8(LDS_SUBDEPTH + i, 0);
  lds_sync();
  // 2. pass A: codes of <= 8 bits fill the root (replicated); longer codes record the depth their
  //    8-bit prefix needs (byte-wise max through compare-and-swap on the containing LDS dword)
  for (uint32_t b = 0; b < n_sym; b += 64) {
    uint32_t sym = b + lane;
    CodeOfLane c = lane_code<1>(sym < n_sym ? lds_ld8(LDS_LENGTHS + sym) : 0, first_code, run, cnt);
    if (c.len != 0) {
      if (c.len <= ROOT_BITS) {
        lds_st16(LDS_OWN + ((c.code << (ROOT_BITS - c.len)) << 1), (sym << 4) | c.len);
      } else {
        // (the depth a prefix needs is the largest its codes ask for: every code sets the bit of its own depth in the prefix' byte -- one
        // atomic `or`, no answer waited for; round 4 raised the byte through compare-and-swap, a round trip and a retry for every lane
        // that shared a dword with another: half a table's time)
        uint32_t d = c.len - ROOT_BITS;
        uint32_t p = c.code >> d;
        __attribute__((address_space(3))) uint32_t* w = reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(&g_smem[LDS_SUBDEPTH + (p & ~3u)]);
        (void)__hip_atomic_fetch_or(w, (1u << (d - 1u)) << ((p & 3u) * 8u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  BT_PROF(6);
  uint32_t total = 256;
  if (two_level) {
    lds_sync();
    // 3. 2nd-level bases: exclusive prefix sum of 2^depth over the 256 prefixes (4 per lane, in code order)
    uint32_t dd = lds_ld32(LDS_SUBDEPTH + lane * 4);
    // (a byte of depth bits -> the largest depth; written back as a number for the second pass)
    uint32_t d0 = 32u - (uint32_t)__clz(dd & 0xFFu), d1 = 32u - (uint32_t)__clz((dd >> 8) & 0xFFu), d2 = 32u - (uint32_t)__clz((dd >> 16) & 0xFFu), d3 = 32u - (uint32_t)__clz(dd >> 24);
    lds_st32(LDS_SUBDEPTH + lane * 4, d0 | (d1 << 8) | (d2 << 16) | (d3 << 24));
    uint32_t s0 = d0 ? (1u << d0) : 0, s1 = d1 ? (1u << d1) : 0, s2 = d2 ? (1u << d2) : 0, s3 = d3 ? (1u << d3) : 0;
    uint32_t mine = s0 + s1 + s2 + s3;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= (uint32_t)off) incl += t;
    }
    total = 256 + rdlane(incl, 63);
    uint32_t b0 = 256 + incl - mine, b1 = b0 + s0, b2 = b1 + s1, b3 = b2 + s2;
    lds_st16(LDS_SUBBASE + lane * 8 + 0, b0); lds_st16(LDS_SUBBASE + lane * 8 + 2, b1);
    lds_st16(LDS_SUBBASE + lane * 8 + 4, b2); lds_st16(LDS_SUBBASE + lane * 8 + 6, b3);
    // root entries that point down (a prefix with longer codes below it owns its one place)
    if (d0) lds_st16(LDS_OWN + ((lane * 4 + 0) << 1), (b0 << 4) | (ROOT_BITS + d0));
    if (d1) lds_st16(LDS_OWN + ((lane * 4 + 1) << 1), (b1 << 4) | (ROOT_BITS + d1));
    if (d2) lds_st16(LDS_OWN + ((lane * 4 + 2) << 1), (b2 << 4) | (ROOT_BITS + d2));
    if (d3) lds_st16(LDS_OWN + ((lane * 4 + 3) << 1), (b3 << 4) | (ROOT_BITS + d3));
    lds_sync();
    // 4. pass B: fill the 2nd-level tables
#pragma unroll
    for (int l = 0; l < 16; l++) run[l] = 0;
    for (uint32_t b = 0; b < n_sym; b += 64) {
      uint32_t sym = b + lane;
      CodeOfLane c = lane_code<ROOT_BITS + 1>(sym < n_sym ? lds_ld8(LDS_LENGTHS + sym) : 0, first_code, run, cnt);
      if (c.len > ROOT_BITS) {
        uint32_t sl = c.len - ROOT_BITS;
        uint32_t p = c.code >> sl;
        uint32_t base = lds_ld16(LDS_SUBBASE + p * 2), depth = lds_ld8(LDS_SUBDEPTH + p);
        uint32_t e = (sym << 4) | sl;
        for (uint32_t j = rev_bits(c.code & mask_bits(sl), sl); j < (1u << depth); j += (1u << sl))
          a.st16_lane(tree + ((base + j) << 1), e);
      }
    }
  }
  lds_sync();
  {
    // the root: four places a lane, the last note in front of each (over the lanes: a scan that keeps the right-hand note where there is one)
    const uint32_t w0 = lds_ld32(LDS_OWN + lane * 8u), w1 = lds_ld32(LDS_OWN + lane * 8u + 4u);
    const uint32_t v0 = w0 & 0xFFFFu, v1 = w0 >> 16, v2 = w1 & 0xFFFFu, v3 = w1 >> 16;
    uint32_t x = v3 ? v3 : v2 ? v2 : v1 ? v1 : v0;
    { uint32_t t;
      t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x = x ? x : t;
      t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); x = x ? x : t;
      t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x = x ? x : t;
      t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); x = x ? x : t;
      t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x = x ? x : t;
      t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); x = x ? x : t; }
    const uint32_t carry = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);   // (the lane before's: wave_shr 1)
    const uint32_t e0 = v0 ? v0 : carry, e1 = v1 ? v1 : e0, e2 = v2 ? v2 : e1, e3 = v3 ? v3 : e2;
    a.st16_lane(tree + (rev_bits(lane * 4 + 0, 8) << 1), e0); a.st16_lane(tree + (rev_bits(lane * 4 + 1, 8) << 1), e1);
    a.st16_lane(tree + (rev_bits(lane * 4 + 2, 8) << 1), e2); a.st16_lane(tree + (rev_bits(lane * 4 + 3, 8) << 1), e3);
  }
  lds_sync();
  BT_PROF(7);
  return total;
}

__device__ __forceinline__ uint32_t max_table_entries(uint32_t alphabet) {
  uint32_t i = (alphabet + 31) >> 5;
  return kMaxTableSize[i > 36 ? 36 : i];
}

__device__ __forceinline__ uint32_t log2floor_plus1(uint32_t x) { return x ? 32u - (uint32_t)__clz(x) : 0u; }

// The symbol code lengths of one prefix code (decode.rs:661-797, 558-658) -- see read_huffman_code, which calls this where the
// code-length code has more than one symbol.  In: the reader (its copy in memory), the code-length code's table (lane k: symbol << 4 |
// bits for the five stream bits k).  Out: the lengths in LDS_LENGTHS, the reader behind them, what is left of the code space.
struct LengthsOut { uint32_t symbol, space; };
__device__ __noinline__ int read_symbol_lengths_wide(BitReader* const brp, const uint32_t cl_table, const uint32_t max_symbol_, LengthsOut* const res) {
  BitReader br = *brp; br.uniformize();
  const uint32_t lane = lane_id(), max_symbol = rfl(max_symbol_);
  uint32_t symbol = 0, prev_code_len = 8, repeat = 0, repeat_code_len = 0, space = 32768;
    // Sixty-four stream bits a step instead of one code word (round 5; one word at a time took 275 clocks a symbol, 40 % of a metablock
  // header): lane j decodes the code word that WOULD start at bit j; the chain of the words that do -- from bit 0 on, a word
  // ends where the next one starts -- comes out of pointer doubling (R_k[j]: the bit 2^k words on from bit j), lane t taking the
  // t-th word's place by the binary digits of t; then the loop's arithmetic for all words side by side: the repeat codes' run
  // lengths (decode.rs:607-650: a run of sixteens, or of seventeens, is a number in base four, or eight) link by link along
  // the runs -- few and short --, symbol indices and code space by prefix sums, the first word that ends the loop (alphabet full,
  // space used up, input short, a repeat that overshoots) by ballots.  Same words, same order, same verdicts as the loop below,
  // which stays for a code-length code of ONE symbol (its words are no bits long: no chain).
  while (symbol < max_symbol && space > 0) {
      const auto bp = [](uint32_t lane_idx, uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane_idx << 2), (int)v); };
      br.need32();
      uint64_t wlo, whi;
      br.window128(wlo, whi);   // (at least 96 bits from the reader's position)
      const uint64_t tb = BitReader::total_bits(), ps = br.pos();
      const uint32_t rem = tb > ps ? (tb - ps > 4096ull ? 4096u : (uint32_t)(tb - ps)) : 0u;   // bits of input left
      const uint32_t bj = (uint32_t)(lane == 0 ? wlo : (wlo >> lane) | (whi << (64u - lane))) & 0xFFu;
      const uint32_t ce = bp(bj & 31u, cl_table);
      const uint32_t sym_j = ce >> 4, nb_j = ce & 15u, eb_j = sym_j == 16u ? 2u : sym_j == 17u ? 3u : 0u;
      const uint32_t ev = sym_j | ((nb_j + eb_j) << 5) | (((bj >> nb_j) & ((1u << eb_j) - 1u)) << 9);   // symbol, bits of the word with its extra bits, their value
      // the chain
      uint32_t R0 = lane + nb_j + eb_j, R1, R2, R3, R4, R5;
      { const uint32_t g = bp(R0 & 63u, R0); R1 = R0 < 64u ? g : R0; }   // (every lane asks, whatever it keeps: a lane that does not take part in a permute is read as zero)
      { const uint32_t g = bp(R1 & 63u, R1); R2 = R1 < 64u ? g : R1; }
      { const uint32_t g = bp(R2 & 63u, R2); R3 = R2 < 64u ? g : R2; }
      { const uint32_t g = bp(R3 & 63u, R3); R4 = R3 < 64u ? g : R3; }
      { const uint32_t g = bp(R4 & 63u, R4); R5 = R4 < 64u ? g : R4; }
      uint32_t pt = 0;   // where word t starts (64 and more: behind the window)
      { uint32_t g;
        g = bp(pt & 63u, R0); pt = ((lane & 1u) && pt < 64u) ? g : pt;
        g = bp(pt & 63u, R1); pt = ((lane & 2u) && pt < 64u) ? g : pt;
        g = bp(pt & 63u, R2); pt = ((lane & 4u) && pt < 64u) ? g : pt;
        g = bp(pt & 63u, R3); pt = ((lane & 8u) && pt < 64u) ? g : pt;
        g = bp(pt & 63u, R4); pt = ((lane & 16u) && pt < 64u) ? g : pt;
        g = bp(pt & 63u, R5); pt = ((lane & 32u) && pt < 64u) ? g : pt; }
      const bool valid = pt < 64u;
      const uint32_t et = bp(pt & 63u, ev);
      const uint32_t cl = et & 31u, endp = pt + ((et >> 5) & 15u), xv = et >> 9;
      const bool islen = valid && cl < 16u, isrep = valid && cl >= 16u;
      const uint64_t below = (1ull << lane) - 1ull;
      // the code length a sixteen repeats: the last one that was not zero
      const uint64_t nzm = __ballot(islen && cl != 0u);
      const uint32_t pvg = bp((nzm & below) != 0ull ? 63u - (uint32_t)__clzll((long long)(nzm & below)) : 0u, cl);
      const uint32_t pv = (nzm & below) != 0ull ? pvg : prev_code_len;
      const uint32_t new_len = cl == 16u ? pv : 0u;
      const uint32_t clp = bp((lane + 63u) & 63u, cl);   // the word before
      const bool cont0 = repeat > 0u && repeat_code_len == new_len;   // (lane 0: the run goes on from the window before)
      const bool conts = isrep && (lane == 0u ? cont0 : clp == cl);
      const uint64_t heads = __ballot(isrep && (lane == 0u || !conts));
      const uint32_t depth = isrep ? lane - (63u - (uint32_t)__clzll((long long)(heads & (below | (1ull << lane))))) : 0u;
      const uint32_t ebt = cl == 16u ? 2u : 3u;
      uint32_t rp = 0, delta = 0;
      for (uint32_t dd = 0; __ballot(isrep && depth >= dd) != 0ull; dd++) {
        const uint32_t before = bp((lane + 63u) & 63u, rp);
        if (isrep && depth == dd) {
          const uint32_t pr = dd == 0u ? (lane == 0u && cont0 ? repeat : 0u) : before;
          rp = (pr > 0u ? (pr - 2u) << ebt : 0u) + xv + 3u;
          delta = rp - pr;
        }
      }
      const uint32_t adv = islen ? 1u : isrep ? delta : 0u;
      const uint32_t used = (islen && cl != 0u) ? 32768u >> cl : (isrep && new_len != 0u) ? delta << (15u - new_len) : 0u;
      const auto scan = [](uint32_t v) -> uint32_t {   // inclusive prefix sum over the lanes (row shifts and row broadcasts, as the engines' sc_scan)
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
        return v;
      };
      const uint32_t a_in = scan(adv), u_in = scan(used);
      const uint32_t symidx = symbol + a_in - adv, space_before = space - (u_in - used);
      const bool goes = valid && symidx < max_symbol && space_before != 0u;   // the loop's condition in front of this word
      const uint64_t stopm = ~__ballot(goes);
      const uint32_t s_n = stopm != 0ull ? (uint32_t)__builtin_ctzll(stopm) : 64u;   // words the loop takes here
      const uint64_t taken = s_n >= 64u ? ~0ull : (1ull << s_n) - 1ull;
      const uint64_t shortm = __ballot(valid && endp > rem) & taken, overm = __ballot(isrep && symidx + delta > max_symbol) & taken;
      if ((shortm | overm) != 0ull) {
        const uint32_t fs = shortm != 0ull ? (uint32_t)__builtin_ctzll(shortm) : 64u, fo = overm != 0ull ? (uint32_t)__builtin_ctzll(overm) : 64u;
        if (fs <= fo) { br.advance(rdlane(endp, fs)); *brp = br; return E_NEEDS_MORE_INPUT; }
        { *brp = br; return E_HUFFMAN_SPACE; }   // (decode.rs:640-643: the repeat overshoots the alphabet)
      }
      if (s_n == 0u) break;   // (cannot happen: the loop's condition held)
      if (lane < s_n) {
        if (islen && cl != 0u) lds_st8(LDS_LENGTHS + symidx, cl);
        if (isrep && new_len != 0u) for (uint32_t k = 0; k < delta; k++) lds_st8(LDS_LENGTHS + symidx + k, new_len);
      }
      const uint32_t last = s_n - 1u;
      symbol = rdlane(symidx + adv, last);
      space = rdlane(space_before - used, last);
      { const uint64_t nzt = nzm & taken; if (nzt != 0ull) prev_code_len = rdlane(cl, 63u - (uint32_t)__clzll((long long)nzt)); }
      { const uint32_t lc = rdlane(cl, last); repeat = lc >= 16u ? rdlane(rp, last) : 0u; if (lc >= 16u) repeat_code_len = rdlane(new_len, last); }
      br.advance(rdlane(endp, last));
    }
  *brp = br;
  res->symbol = symbol; res->space = space;
  return E_SUCCESS;
}

// src/decode.rs:868-1013.  Reads one prefix code, builds its table at a fresh arena allocation, returns the
// arena offset in *tree_off.  The one helper that is a real function call (7 call sites, cold).
#ifdef BROTLI_AMD_PROFILE_HDR
#define HDR_PROF(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && lane_id() == 0) g_hdr_prof[k] += t_ - hp_t; hp_t = t_; } while (0)
#else
#define HDR_PROF(k) do { } while (0)
#endif
__device__ __noinline__ int read_huffman_code(Stream& s, uint32_t alphabet_size, uint32_t max_symbol, uint32_t* tree_off, bool cold = false) {
#ifdef BROTLI_AMD_PROFILE_HDR
  uint64_t hp_t = __builtin_amdgcn_s_memtime();
#endif
  struct Scope {  // register copies of the reader and the arena, stored back on every exit
    Stream& s; BitReader br; Arena ar;
    __device__ __forceinline__ Scope(Stream& s_) : s(s_), br(s_.br), ar(s_.ar) { COLD_UNIFORMIZE(br.uniformize(); ar.uniformize();) }
    __device__ __forceinline__ ~Scope() { s.br = br; s.ar = ar; }
  } sc(s);
  BitReader& br = sc.br;
  Arena& ar = sc.ar;
  const uint32_t lane = lane_id();
  alphabet_size = rfl(alphabet_size) & 0x7ffu;
  max_symbol = rfl(max_symbol);
  uint32_t max_entries = max_table_entries(alphabet_size);
  uint32_t tree = cold ? ar.alloc_cold(max_entries * 2) : ar.alloc(max_entries * 2);
  *tree_off = tree;
  uint32_t hskip = br.read(2); NEED_INPUT(br);
  for (uint32_t i = lane; i < MAX_ALPHABET; i += 64) lds_st8(LDS_LENGTHS + i, 0);
  lds_sync();
  uint32_t size;
  if (hskip == 1) {
    uint32_t nsym = br.read(2); NEED_INPUT(br);
    uint32_t max_bits = log2floor_plus1(alphabet_size - 1);
    uint32_t sy0 = 0, sy1 = 0, sy2 = 0, sy3 = 0;
    for (uint32_t i = 0; i <= nsym; i++) {
      uint32_t v = br.read(max_bits); NEED_INPUT(br);
      if (v >= max_symbol) FAIL(br, E_SIMPLE_HUFFMAN_ALPHABET);
      if (i == 0) sy0 = v; else if (i == 1) sy1 = v; else if (i == 2) sy2 = v; else sy3 = v;
    }
    if (nsym >= 1 && sy0 == sy1) FAIL(br, E_SIMPLE_HUFFMAN_SAME);
    if (nsym >= 2 && (sy0 == sy2 || sy1 == sy2)) FAIL(br, E_SIMPLE_HUFFMAN_SAME);
    if (nsym >= 3 && (sy0 == sy3 || sy1 == sy3 || sy2 == sy3)) FAIL(br, E_SIMPLE_HUFFMAN_SAME);
    if (nsym == 3) { nsym += br.read(1); NEED_INPUT(br); }
    if (nsym == 0) {  // single symbol, zero-length code
      for (uint32_t j = lane; j < 256; j += 64) ar.st16_lane(tree + (j << 1), sy0 << 4);
      lds_sync();
      size = 256;
    } else {
      if (lane == 0) {
        if (nsym == 1) { lds_st8(LDS_LENGTHS + sy0, 1); lds_st8(LDS_LENGTHS + sy1, 1); }
        else if (nsym == 2) { lds_st8(LDS_LENGTHS + sy0, 1); lds_st8(LDS_LENGTHS + sy1, 2); lds_st8(LDS_LENGTHS + sy2, 2); }
        else if (nsym == 3) { lds_st8(LDS_LENGTHS + sy0, 2); lds_st8(LDS_LENGTHS + sy1, 2); lds_st8(LDS_LENGTHS + sy2, 2); lds_st8(LDS_LENGTHS + sy3, 2); }
        else { lds_st8(LDS_LENGTHS + sy0, 1); lds_st8(LDS_LENGTHS + sy1, 2); lds_st8(LDS_LENGTHS + sy2, 3); lds_st8(LDS_LENGTHS + sy3, 3); }
      }
      lds_sync();
      size = build_tree(ar, tree, alphabet_size);
    }
  } else {
    // code-length code (decode.rs:801-853): 18 lengths of 0..5 bits, kept one per lane (lane = symbol)
    uint32_t cl_vgpr = 0;  // lane i: length of code-length symbol i
    uint32_t space = 32, num_codes = 0;
    for (uint32_t i = hskip; i < 18; i++) {
      uint32_t ix = br.peek32() & 0xFu;
      // (the three small tables of decode.rs:801-853 as immediates, a nibble or five bits an entry: out of constant memory every
      // one of them was a scalar load on the chain -- 830 clocks a symbol)
      br.drop((uint32_t)(0x4222322242223222ull >> (ix << 2)) & 15u); NEED_INPUT(br);
      uint32_t v = (uint32_t)(0x5340234013402340ull >> (ix << 2)) & 15u;
      if (lane == ((uint32_t)((i < 12u ? 0x4a0f0344a020c41ull >> (5u * i) : 0x1ee6b16aull >> (5u * (i - 12u)))) & 31u)) cl_vgpr = v;
      if (v != 0) {
        space -= (32u >> v); num_codes++;
        if (space - 1u >= 32u) break;
      }
    }
    if (!(num_codes == 1 || space == 0)) FAIL(br, E_CL_SPACE);
    // 32-entry lookup for the code-length code, one entry per lane: (symbol << 4) | bits  (huffman/mod.rs:196-271)
    uint32_t cl_table;
    {
      uint32_t L = lane < 18 ? cl_vgpr : 0;
      if (num_codes == 1) {
        uint64_t m = __ballot(L != 0);
        cl_table = ((uint32_t)__ffsll((long long)m) - 1u) << 4;  // zero-length code
      } else {
        uint32_t code = 0, my_code = 0;
#pragma unroll
        for (int l = 1; l <= 5; l++) {
          uint64_t m = __ballot(L == (uint32_t)l);
          if (L == (uint32_t)l) my_code = code + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          code = (code + (uint32_t)__popcll(m)) << 1;
        }
        cl_table = 0;
        for (uint32_t sy = 0; sy < 18; sy++) {
          uint32_t sl = rdlane(L, sy);
          if (sl == 0) continue;
          uint32_t rc = rev_bits(rdlane(my_code, sy), sl);
          if ((lane & mask_bits(sl)) == rc) cl_table = (sy << 4) | sl;
        }
      }
    }
    HDR_PROF(0);
    // symbol code lengths (decode.rs:661-797, 558-658)
    uint32_t symbol = 0, prev_code_len = 8, repeat = 0, repeat_code_len = 0;
    space = 32768;
#ifndef BROTLI_AMD_SERIAL_LENGTHS
    if (num_codes != 1) {
      LengthsOut lo_;
      const int e_ = rfl(read_symbol_lengths_wide(&br, cl_table, max_symbol, &lo_));
      COLD_UNIFORMIZE(br.uniformize();)
      if (e_ == E_NEEDS_MORE_INPUT) { NEED_INPUT(br); return E_NEEDS_MORE_INPUT; }
      if (e_ != E_SUCCESS) FAIL(br, e_);
      symbol = max_symbol; space = rfl(lo_.space);   // (the loop below has nothing left to do)
    }
#endif
    while (symbol < max_symbol && space > 0) {
      uint32_t p = rdlane(cl_table, br.peek32() & 31u);
      uint32_t code_len = p >> 4;
      br.drop(p & 15u);
      if (code_len < 16) {
        NEED_INPUT(br);
        repeat = 0;
        if (code_len != 0) {
          if (lane == 0) lds_st8(LDS_LENGTHS + symbol, code_len);
          prev_code_len = code_len;
          space -= (32768u >> code_len);
        }
        symbol++;
      } else {
        uint32_t extra = code_len - 14;
        uint32_t repeat_delta = br.read(extra); NEED_INPUT(br);
        uint32_t new_len = (code_len == 16) ? prev_code_len : 0;
        if (repeat_code_len != new_len) { repeat = 0; repeat_code_len = new_len; }
        uint32_t old_repeat = repeat;
        if (repeat > 0) { repeat -= 2; repeat <<= extra; }
        repeat += repeat_delta + 3;
        repeat_delta = repeat - old_repeat;
        if (symbol + repeat_delta > max_symbol) { symbol = max_symbol; space = 0xFFFFF; continue; }
        if (repeat_code_len != 0) {
          for (uint32_t k = lane; k < repeat_delta; k += 64) lds_st8(LDS_LENGTHS + symbol + k, repeat_code_len);
          space -= (repeat_delta << (15 - repeat_code_len));
        }
        symbol += repeat_delta;
      }
    }
    if (space != 0) FAIL(br, E_HUFFMAN_SPACE);
    lds_sync();
    HDR_PROF(1);
#ifdef BROTLI_AMD_PROFILE_HDR
    if (blockIdx.x == 0 && lane == 0) { g_hdr_prof[4] += 1; g_hdr_prof[5] += symbol; }
#endif
    size = build_tree(ar, tree, max_symbol);
    HDR_PROF(2);
  }
  if (!cold) ar.shrink_to(tree + size * 2);
  return E_SUCCESS;
}

// Working copy of the bit reader for inlined cold code: lives in registers between the helper calls that
// go through the Stream object in memory.
struct ColdScope {
  Stream& s; BitReader br;
  __device__ __forceinline__ ColdScope(Stream& s_) : s(s_), br(s_.br) { COLD_UNIFORMIZE(br.uniformize();) }
  __device__ __forceinline__ ~ColdScope() { s.br = br; }
  __device__ __forceinline__ int huffman(uint32_t alphabet, uint32_t max_symbol, uint32_t* tree, bool cold = false) {
    s.br = br;
    int e = rfl(read_huffman_code(s, alphabet, max_symbol, tree, cold));
    br = s.br; COLD_UNIFORMIZE(br.uniformize(); *tree = rfl(*tree);)
    return e;
  }
};

// decode.rs:193-241
__device__ __forceinline__ int decode_varlen_uint8(BitReader& br, uint32_t* value) {
  uint32_t b = br.read(1); NEED_INPUT(br);
  if (!b) { *value = 0; return E_SUCCESS; }
  uint32_t n = br.read(3); NEED_INPUT(br);
  if (!n) { *value = 1; return E_SUCCESS; }
  uint32_t v = br.read(n); NEED_INPUT(br);
  *value = (1u << n) + v;
  return E_SUCCESS;
}

// decode.rs:243-372
__device__ __forceinline__ int decode_metablock_length(BitReader& br, Stream& s) {
  uint32_t is_last = br.read(1); NEED_INPUT(br);
  int32_t mlen = 0;
  s.is_last = is_last; s.mlen = 0; s.is_uncompressed = 0; s.is_metadata = 0;
  if (is_last) {
    uint32_t empty = br.read(1); NEED_INPUT(br);
    if (empty) return E_SUCCESS;
  }
  uint32_t nib = br.read(2); NEED_INPUT(br);
  bool is_metadata = false;
  if (nib == 3) {
    is_metadata = true;
    s.is_metadata = 1;
    uint32_t reserved = br.read(1); NEED_INPUT(br);
    if (reserved) return E_RESERVED;
    uint32_t nbytes = br.read(2); NEED_INPUT(br);
    if (nbytes == 0) return E_SUCCESS;
    for (uint32_t i = 0; i < nbytes; i++) {
      uint32_t b = br.read(8); NEED_INPUT(br);
      if (i + 1 == nbytes && nbytes > 1 && b == 0) return E_EXUBERANT_META_NIBBLE;
      mlen |= (int32_t)(b << (i * 8));
    }
  } else {
    uint32_t size_nibbles = nib + 4;
    for (uint32_t i = 0; i < size_nibbles; i++) {
      uint32_t b = br.read(4); NEED_INPUT(br);
      if (i + 1 == size_nibbles && size_nibbles > 4 && b == 0) return E_EXUBERANT_NIBBLE;
      mlen |= (int32_t)(b << (i * 4));
    }
  }
  if (!is_last && !is_metadata) { s.is_uncompressed = br.read(1); NEED_INPUT(br); }
  s.mlen = mlen + 1;
  return E_SUCCESS;
}

__device__ __forceinline__ bool jump_to_byte_boundary(BitReader& br) {  // bit_reader/mod.rs:378-385
  uint32_t pad = (uint32_t)((8u - (uint32_t)(br.pos() & 7u)) & 7u);
  return br.read(pad) == 0;
}

// decode.rs:1016-1026: block length = base[code] + extra bits
template <bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t read_block_length(BitReader& br, const Arena& ar, uint32_t bl_vgpr, uint32_t tree) {
  uint32_t code = read_symbol<LDS_ONLY>(br, ar, tree);
  uint32_t e = rdlane(bl_vgpr, code);
  return (e & 0xFFFFu) + br.read(e >> 16);
}

// decode.rs:1469-1524.  0 = single block type, 1 = switched, 2 = needs more input
enum { BS_SINGLE_TYPE = 0, BS_SWITCHED = 1, BS_NEEDS_INPUT = 2 };
template <bool LDS_ONLY>
__device__ __forceinline__ int block_switch(BitReader& br, const Arena& ar, uint32_t bl_vgpr, uint32_t bt_tree, uint32_t bl_tree, uint32_t nbt,
                                            uint32_t& block_len, uint32_t& t0, uint32_t& t1) {
  if (nbt <= 1) return BS_SINGLE_TYPE;
  uint32_t block_type = read_symbol<LDS_ONLY>(br, ar, bt_tree);
  uint32_t len = read_block_length<LDS_ONLY>(br, ar, bl_vgpr, bl_tree);
  if (br.over()) return BS_NEEDS_INPUT;
  block_len = len;
  if (block_type == 1) block_type = t1 + 1;
  else if (block_type == 0) block_type = t0;
  else block_type -= 2;
  if (block_type >= nbt) block_type -= nbt;
  t0 = t1; t1 = block_type;
  return BS_SWITCHED;
}

// decode.rs:1272-1428.  The map is written to the arena; *num_trees gets NTREES.
__device__ __noinline__ int decode_context_map(Stream& s, uint32_t size, uint32_t* num_trees, uint32_t* map_off) {
  ColdScope c(s);
  BitReader& br = c.br;
  const uint32_t lane = lane_id();
  uint32_t n;
  TRY(decode_varlen_uint8(br, &n));
  n += 1; *num_trees = n;
  // decoded where it is fast to work on (arena top, LDS while there is room), then parked with the cold objects
  const uint32_t hot_top = s.ar.top;
  uint32_t map = s.ar.alloc(size);
  const uint32_t parked = s.ar.alloc_cold(size);
  *map_off = parked;
  { const Arena ar = s.ar; for (uint32_t i = lane; i < size; i += 64) { ar.st8_lane(map + i, 0); ar.st8_lane(parked + i, 0); } }
  lds_sync();
  if (n <= 1) { s.ar.top = hot_top; return E_SUCCESS; }
  if (br.pos() + 5 > br.total_bits()) return E_NEEDS_MORE_INPUT;  // SafeGetBits(5), decode.rs:1311
  uint32_t bits = br.peek32() & 31u;
  uint32_t max_rle;
  if (bits & 1u) { max_rle = (bits >> 1) + 1; br.drop(5); } else { max_rle = 0; br.drop(1); }
  uint32_t saved_top = s.ar.top, tree;
  TRY(c.huffman(n + max_rle, n + max_rle, &tree));
  const Arena ar = s.ar;
  uint32_t i = 0;
  while (i < size) {
    uint32_t code = read_symbol(br, ar, tree); NEED_INPUT(br);
    if (code == 0) { i++; continue; }
    if (code > max_rle) { ar.st8(map + i, code - max_rle); i++; continue; }
    uint32_t reps = br.read(code); NEED_INPUT(br);
    reps += 1u << code;
    if (i + reps > size) return E_CONTEXT_MAP_REPEAT;
    i += reps;
  }
  uint32_t imtf = br.read(1); NEED_INPUT(br);
  lds_sync();
  if (imtf) {  // decode.rs:1096-1128; serial over the map, the move itself uses the lanes
    for (uint32_t k = lane; k < 256; k += 64) lds_st8(LDS_MTF + k, k);
    lds_sync();
    for (uint32_t k = 0; k < size; k++) {
      uint32_t idx = ar.ld8(map + k);
      if (idx == 0) { ar.st8(map + k, rfl(lds_ld8(LDS_MTF))); lds_sync(); continue; }
      uint32_t val = rfl(lds_ld8(LDS_MTF + idx));
      // mtf[1..idx] = mtf[0..idx-1]: 64 entries per step, top chunk first, reads of a step before its writes
      for (int b = (int)((idx - 1) & ~63u); b >= 0; b -= 64) {
        uint32_t j = (uint32_t)b + lane;
        uint32_t t = j < idx ? lds_ld8(LDS_MTF + j) : 0;
        lds_sync();
        if (j < idx) lds_st8(LDS_MTF + j + 1, t);
        lds_sync();
      }
      if (lane == 0) lds_st8(LDS_MTF, val);
      ar.st8(map + k, val);
      lds_sync();
    }
  }
  // park the map; its working copy and its prefix code are dead now: give their arena space back
  lds_sync();
  for (uint32_t i = lane; i < size; i += 64) ar.st8_lane(parked + i, ar.ld8_lane(map + i));
  lds_sync();
  (void)saved_top;
  s.ar.top = hot_top;
  return E_SUCCESS;
}

// decode.rs:1130-1219: `ntrees` prefix codes; their arena offsets go to a u32 array
__device__ __noinline__ int decode_tree_group(Stream& s, uint32_t alphabet, uint32_t max_symbol, uint32_t ntrees, uint32_t* group_off) {
  uint32_t g = s.ar.alloc_cold(ntrees * 4);
  *group_off = g;
  for (uint32_t t = 0; t < ntrees; t++) {
    uint32_t tree;
    TRY(read_huffman_code(s, alphabet, max_symbol, &tree));
    s.ar.st32(g + t * 4, tree);
  }
  lds_sync();
  return E_SUCCESS;
}

// decode.rs:2766-2777
__device__ __forceinline__ uint32_t max_distance_symbol(uint32_t ndirect, uint32_t npostfix) {
  uint32_t postfix = 1u << npostfix;
  uint32_t b = npostfix == 0 ? 0u : npostfix == 1 ? 4u : npostfix == 2 ? 12u : 28u;
  uint32_t d = npostfix == 0 ? 73u : npostfix == 1 ? 126u : npostfix == 2 ? 228u : 424u;
  if (ndirect < b) return ndirect + d + postfix;
  if (ndirect > b + postfix) return ndirect + d;
  return b + d + postfix;
}

// ============================== output: literals, copies, dictionary words ==============================
// Length and affixes of a transformed dictionary word (transform.rs:737-795), everything uniform.
struct WordShape { uint32_t pre, plen, suf, slen, skip, wlen, t, total; };
__device__ __forceinline__ WordShape word_shape(uint32_t len, uint32_t transform_idx) {
  WordShape w;
  // (one load: the affix lengths are in the table -- counting them out of kAffixPool was a dependent load a byte, and a word of the
  // dictionary is one command in nine of the reference's text fixtures)
  const uint32_t ti = kTransformInfo[transform_idx];
  w.pre = ti & 0xFFu; w.t = (ti >> 8) & 31u; w.suf = (ti >> 13) & 0xFFu; w.plen = (ti >> 21) & 15u; w.slen = ti >> 25;
  w.skip = w.t < 12 ? 0 : w.t - 11;
  if (w.skip > len) w.skip = len;
  int32_t wl = (int32_t)(len - w.skip);
  if (w.t <= 9) wl -= (int32_t)w.t;
  w.wlen = wl > 0 ? (uint32_t)wl : 0;
  w.total = w.plen + w.wlen + w.slen;
  return w;
}

// decode.rs:2593-2640 + transform.rs:737-795.  Returns the bytes of the (transformed) word, one per lane.
// (`stage`: 128 bytes of LDS for the two transforms that walk the word: the decoding wave's LDS_WORD, or a wave's own where several
// waves put words together side by side -- the path engine's execute)
__device__ __forceinline__ uint32_t dictionary_word_bytes(gcu8* dict, uint32_t offset, const WordShape& w, const uint32_t stage = LDS_WORD) {
  const uint32_t lane = lane_id();
  uint32_t b = 0;
  if (lane < w.plen) b = kAffixPool[w.pre + lane];
  else if (lane < w.plen + w.wlen) b = dict[offset + w.skip + (lane - w.plen)];
  if (w.t == 10 || w.t == 11) {  // transform.rs:720-735 -- serial over UTF-8 sequences, staged through LDS
    lds_st8(stage + lane, b);
    lds_st8(stage + 64u + lane, 0);  // bytes after the word are zero so that a stray write past it is harmless
    lds_sync();
    if (lane == 0) {
      uint32_t p = stage + w.plen;
      int32_t remaining = (w.t == 10) ? 1 : (int32_t)w.wlen;
      while (remaining > 0) {
        int step;
        uint32_t c0 = lds_ld8(p);
        if (c0 < 0xc0) { if (c0 >= 'a' && c0 <= 'z') lds_st8(p, c0 ^ 32); step = 1; }
        else if (c0 < 0xe0) { lds_st8(p + 1, lds_ld8(p + 1) ^ 32); step = 2; }
        else { lds_st8(p + 2, lds_ld8(p + 2) ^ 5); step = 3; }
        p += step; remaining -= step;
        if (w.t == 10) break;
      }
    }
    lds_sync();
    b = lds_ld8(stage + lane);
    lds_sync();
  }
  if (lane >= w.plen + w.wlen && lane < w.total) b = kAffixPool[w.suf + (lane - w.plen - w.wlen)];
  return b;
}

#ifdef BROTLI_AMD_PROFILE
#define PROF_T() __builtin_amdgcn_s_memtime()
#define PROF_ADD(acc, t0) do { uint64_t _t = __builtin_amdgcn_s_memtime(); (acc) += _t - (t0); (t0) = _t; } while (0)
#else
#define PROF_T() 0ull
#define PROF_ADD(acc, t0) do { } while (0)
#endif
// BROTLI_AMD_PROFILE_LIT: the dist / copy slots report the batched / one-by-one literal loops instead
#ifdef BROTLI_AMD_PROFILE_LIT
#define PROF_LIT(acc, t0) PROF_ADD(acc, t0)
#define PROF_REST(acc, t0) PROF_ADD(prof_cmd, t0)
#else
#define PROF_LIT(acc, t0) do { } while (0)
#define PROF_REST(acc, t0) PROF_ADD(acc, t0)
#endif

// ===================================== helper waves: speculative literal runs =====================================
// A block is up to eight waves: wave 0 decodes its stream, the others wait for work in an LDS mailbox.  The work: a
// run of literals with one prefix code (high-entropy data: tens of thousands of literals between two copies).  A round
// is one chunk of SPEC_WINDOWS * 64 stream bits per wave; the decoding wave takes the first, whose first bit is known to
// start a literal, each helper decodes one of the others *as if* a literal started at its first bit.  It usually does
// not -- but prefix codes re-synchronise: after a few symbols the helper's chain of literal starts falls in with the
// true one.  The decoding wave then walks the true chain from where the chunk before ended only until it steps on a
// start the helper has marked too; from there on the helper's literals are the stream's.  Per window a helper records
// the start mask and the running literal count (LDS), its literals go to a scratch slot in global memory and it moves
// them into place itself once the decoding wave has told it which and where; for a chunk's first SPEC_FIRST windows it
// also leaves code length and symbol of every bit offset, which is all the decoding wave needs to walk there.  No chunk
// that fails to fall in within those windows is used: the round ends in front of it.
//   HCTL words (round-wide): round number, kind (1 = round, 2 = exit, 3 = no rounds in this launch), first dword of
//   the round and bit offset in it, LDS address of the literal tree, base of the per-wave slots, waves in the block,
//   address of the stream's output.  Per wave (HW_*, in its slot): round decoded, literals in the chunk, bit offset
//   into the next chunk at which its chain ends; then, posted by the decoding wave once the chunk's place is known:
//   move wanted for round, first literal of the chunk to move, where to (offset from the output), how many; and the
//   helper's answer, round moved.
enum { HC_SEQ = 0, HC_KIND = 1, HC_DW0 = 2, HC_SHIFT = 3, HC_TREE = 4, HC_BASE = 5, HC_NW = 6, HC_OUT_LO = 7, HC_OUT_HI = 8, HC_CAPPED = 9, HC_FAILED = 10,
       HC_SCAN_BASE = 11 /* LDS base of the command engine (brotli_scan_engine.h); 0 = this block has none */, HC_NW_ALL = 12 /* waves in the block */,
       HC_EXT_BASE = 13 /* LDS base of the command-record ring of the parse / copy split (SPX_BYTES, in the free tail of the table arena); 0 = this metablock has none */,
       // a gang of blocks on one stream (see GC_* below): this block's number in it (0: the stream's owner), the gang's blocks, its control block in memory,
       // the owner's count of the engine's invocations (a helper's: the last one it has seen), the bytes of the table arena in use
       HC_GANG_ROLE = 14, HC_GANG_M = 15, HC_GANG_LO = 16, HC_GANG_HI = 17, HC_GANG_EPOCH = 18, HC_GANG_READY = 19 /* the owner's: what the control block's READY says when the helpers of every invocation so far have left */ };
enum { HK_ROUND = 1, HK_EXIT = 2, HK_NO_ROUNDS = 3, HK_SCAN = 4, HK_PATH = 5, HK_SPLIT = 6, HK_PATH2 = 7, HK_PATHG = 8, HK_PATHR = 9 };  // HC_KIND
enum { HW_DONE = 0, HW_N = 1, HW_EXIT = 2, HW_MVGO = 3, HW_MVSRC = 4, HW_MVDST_LO = 5, HW_MVDST_HI = 6, HW_MVN = 7, HW_MVDONE = 8,
       // a helper's own account of how the chain of the chunk before (entered where that chunk's chain ends) falls in
       // with its own: round resolved, fell in (1/0), literals of that chain before it did, the helper's literals before
       // the common start, windows walked, offset at which the walk left the last of them; and, from the decoding wave
       // with the move: whether those literals are the stream's
       HW_RES = 9, HW_RSYNC = 10, HW_RNM = 11, HW_RSKIP = 12, HW_RWIN = 13, HW_REXIT = 14, HW_MVOWN = 15 };
constexpr uint32_t SPEC_ROUND_MIN = 768;       // literals a run must still have for a round to pay
// input a round of nw chunks may look at, from the reader's next dword on
__device__ __forceinline__ uint32_t spec_input_dwords(uint32_t nw) { return nw * SPEC_WINDOWS * 2u + 74u; }
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
__device__ __forceinline__ uint32_t hc_ld(uint32_t w) { return rfl(*reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * w])); }
__device__ __forceinline__ void hc_st(uint32_t w, uint32_t v) { if (lane_id() == 0) *reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * w]) = v; }
__device__ __forceinline__ uint32_t hw_ld(uint32_t slot, uint32_t k) { return rfl(*reinterpret_cast<lds_vu32*>(&g_smem[slot + HL_CTL + 4u * k])); }
__device__ __forceinline__ void hw_st(uint32_t slot, uint32_t k, uint32_t v) { if (lane_id() == 0) *reinterpret_cast<lds_vu32*>(&g_smem[slot + HL_CTL + 4u * k]) = v; }
__device__ __forceinline__ void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// ---- Several CUs on one stream: the gang's control block in memory (one per stream of a gang launch; the host zeroes it) ----
// A batch of fewer streams than half the CUs is launched as GANGS: the stream's owner -- the block that decodes it, as ever -- and up to
// seven helper blocks, which do nothing but the path engine's regions in turns with it (brotli_path_engine.h, PE_CFG_REMOTE).  What they
// tell each other goes through these words, every access an agent-scope atomic (it bypasses the CU's L1; the L2s of the chip's eight XCDs
// are not coherent for plain accesses):
//   JOINED   helpers that have started (the owner takes the gang or leaves it: a helper that is not running cannot be waited for)
//   EPOCH    the engine's invocations so far, the owner's word: a helper waits for the next one; GC_QUIT: the stream is done
//   READY    invocations the helpers have left, summed (the owner starts the next one when all of them have left the last)
//   PLAN     where the regions' windows lie: generation << 48 | first region << 32 | its first bit; region k's window starts
//            (k - first) strides behind it.  The engine whose turn it is writes a new one where the stream does not enter its window.
//   STOP     epoch << 32 | regions resolved in all: the invocation is over
//   EXEC     epoch << 32 | regions whose output is in memory
//   ENTRY    bit << 0 | tag << 32: where the stream enters region k -- a granule as STATE's, written as soon as region k - 1's WALK knows where
//            it ends: the walk and the details of region k need no more than that, and run while region k - 1 is still being resolved.  (It is
//            what the stream does if all the commands region k - 1 listed go through; if its resolve finds otherwise, the invocation ends there.)
//   STATE    the stream's state behind region k - 1's resolve: 26 granules of value | tag << 32, tag = epoch << 12 | k -- a granule is
//            one eight-byte store and says itself whether it is the one waited for: no flag, no fence
//   PARAMS, BR, ARENA   the invocation's parameters, the bit reader's words and the image of the owner's table arena (plain stores
//            behind a release fence, in front of EPOCH; a helper's acquire fence stands behind its look at EPOCH)
constexpr uint32_t GC_JOINED = 0, GC_EPOCH = 4, GC_READY = 8, GC_PLAN = 16, GC_STOP = 24, GC_EXEC = 32, GC_ARENA_BYTES = 40, GC_ENTRY = 48, GC_MEMBERS = 56 /* u64: invocation << 32 | blocks of its gang */, GC_STATE = 64,
                   GC_PARAMS = 512, GC_BR = 640, GC_ARENA = 1024, GC_ARENA_CAP = 48u << 10, GC_STRIDE = GC_ARENA + GC_ARENA_CAP;
constexpr uint32_t GC_QUIT = 0xFFFFFFFFu, GC_STATE_WORDS = 28, GC_MAX_REGIONS = 4000;   // (the state's granules: 25 words of PeStream, the resolve's flags, the bytes its region put out, those of the region before it)
static_assert(GC_STRIDE == BROTLI_AMD_GANG_CTL_BYTES && GC_STATE + 8u * GC_STATE_WORDS <= GC_PARAMS, "the gang's control block");
__device__ __forceinline__ gu8* gang_ctl() { return (gu8*)(uintptr_t)((uint64_t)hc_ld(HC_GANG_LO) | ((uint64_t)hc_ld(HC_GANG_HI) << 32)); }
__device__ __forceinline__ uint32_t gang_ld32(gu8* gc, uint32_t off) { return __hip_atomic_load(reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(gc + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t gang_ld64(gu8* gc, uint32_t off) { return __hip_atomic_load(reinterpret_cast<__attribute__((address_space(1))) uint64_t*>(gc + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gang_st32(gu8* gc, uint32_t off, uint32_t v) { __hip_atomic_store(reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(gc + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gang_st64(gu8* gc, uint32_t off, uint64_t v) { __hip_atomic_store(reinterpret_cast<__attribute__((address_space(1))) uint64_t*>(gc + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t gang_add32(gu8* gc, uint32_t off, uint32_t v) { return __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(gc + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (-DBROTLI_AMD_GANG_STATS: counters of the gang's life in its control block, from GC_STATS on; the host prints them -- BrotliAmdBatchWait)
constexpr uint32_t GC_STATS = 704;   // u64 x 40
#ifdef BROTLI_AMD_GANG_STATS
#define GANG_STAT(gc_, k_, v_) do { if (lane_id() == 0) __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(1))) unsigned long long*>((gc_) + GC_STATS + 8u * (k_)), (unsigned long long)(v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define GANG_STAT(gc_, k_, v_) do { } while (0)
#endif
// (the stores so far have left this wave: what is stored next cannot overtake them)
__device__ __forceinline__ void gang_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// plain stores of this CU (every wave has waited for its own) reach the other CUs / a look at another CU's word is followed by fresh data
__device__ __forceinline__ void gang_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void gang_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// walk over code lengths from bit `entry` of a window: which offsets start a symbol, and where the chain leaves it
#define SPEC_WALK(Lw, entry, starts, woff) do { uint32_t t1_; \
    asm volatile("s_mov_b64 %0, 0\n\ts_add_u32 %1, %4, 0xffffffc0\n" \
                 "1:\n\tv_readlane_b32 %2, %3, %1\n\ts_bitset1_b64 %0, %1\n\ts_add_u32 %1, %1, %2\n\ts_cbranch_scc0 1b\n\t" \
                 "s_add_u32 %1, %1, 64\n" \
                 : "=&s"(starts), "=&s"(woff), "=&s"(t1_) : "v"(Lw), "s"(entry) : "scc"); } while (0)
// lanes of `mask` store their byte at out[rank]
#define SPEC_STORE(mask, rank, val, ptr) asm volatile("s_mov_b64 exec, %0\n\tglobal_store_byte %1, %2, %3\n\ts_mov_b64 exec, -1" \
                                                      :: "s"(mask), "v"(rank), "v"(val), "s"(ptr) : "memory")

// The same walk two symbols per step: Mv = code length where the symbol after this offset's still starts inside the
// window (else 0), Jv = that length plus the next symbol's (else the length alone).
#define SPEC_WALK2(Mv, Jv, entry, starts, woff) do { uint32_t t1_, t2_; \
    asm volatile("s_mov_b64 %0, 0\n\ts_add_u32 %1, %6, 0xffffffc0\n" \
                 "1:\n\tv_readlane_b32 %2, %4, %1\n\tv_readlane_b32 %3, %5, %1\n\ts_bitset1_b64 %0, %1\n\ts_add_u32 %2, %2, %1\n\t" \
                 "s_bitset1_b64 %0, %2\n\ts_add_u32 %1, %1, %3\n\ts_cbranch_scc0 1b\n\t" \
                 "s_add_u32 %1, %1, 64\n" \
                 : "=&s"(starts), "=&s"(woff), "=&s"(t1_), "=&s"(t2_) : "v"(Mv), "v"(Jv), "s"(entry) : "scc"); } while (0)
__device__ __forceinline__ uint32_t bperm(uint32_t byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); }

// LDS operations of the chunk decoder are issued by hand and waited for by hand (SPEC_WAIT names the registers they
// fill), so that a result can be asked for in one iteration of the loop and used in the next.
#define SPEC_BPERM(dst, addr, data, OFF) asm volatile("ds_bpermute_b32 %0, %1, %2 offset:" #OFF : "=v"(dst) : "v"(addr), "v"(data))
#define SPEC_RD16(dst, addr) asm volatile("ds_read_u16 %0, %1" : "=v"(dst) : "v"(addr))
#define SPEC_WAIT5(a, b, c, d, e) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
#define SPEC_WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
// address of the entry that decides the symbol at an offset: second-level where the root entry says so, else the root
// entry once more (x = the 32 bits at the offset, r = its root entry)
__device__ __forceinline__ uint32_t spec_entry_addr(uint32_t tree, uint32_t x, uint32_t r) {
  const uint32_t Lr = r & 15u;
  const bool sec = Lr > ROOT_BITS;
  const uint32_t idx = (sec ? r >> 4 : 0u) + __builtin_amdgcn_ubfe(x, sec ? ROOT_BITS : 0u, sec ? Lr - ROOT_BITS : ROOT_BITS);
  return tree + (idx << 1);
}

// One chunk: SPEC_WINDOWS windows from stream dword dw0 (+ sh bits), chain entered at bit `entry` of the first window.
// Literals go to symout[0..n), per-window start masks / running counts to the LDS areas of wave slot w.
// The loop is a software pipeline over windows, every LDS result is used one iteration after its request: while window
// j is walked, the code lengths of j + 1 are on their way through ds_bpermute (every offset learns the length of the
// symbol after its own, so the walk takes two symbols per step), the deciding table entries of j + 2 and the root
// entries of j + 3 are being read and the stream bits of j + 4 gathered.  Two copies of the loop body hand the
// registers to each other (A -> B -> A), so nothing is moved at the back edge.
__device__ __noinline__ void spec_chunk(uint32_t w, uint32_t dw0, uint32_t sh, uint32_t tree, gu8* symout, uint32_t entry) {
  const uint32_t lane = lane_id();
  w = rfl(w); dw0 = rfl(dw0); sh = rfl(sh); tree = rfl(tree); entry = rfl(entry);
  symout = rfl_ptr(symout);
  const uint32_t slot = hc_ld(HC_BASE) + w * HL_SLOT;
  const uint32_t hfirst = slot + HL_FIRST;
  gcu32* const base = BitReader::base();
  const uint32_t ndw = BitReader::n_dw(), tmask = BitReader::tail_mask();
  // sa[l], sb[l], sc[l]: the 32 bits of the stream from bit 32 * l + sh of dword dw0, dw0 + 64, dw0 + 128 on
  uint32_t sa, sb, sc = 0;
  {
    constexpr bool three = SPEC_WINDOWS > 32u;
    uint32_t i0 = dw0 + lane, i1 = dw0 + 64u + lane, i2 = dw0 + 128u + lane, v0 = 0, v1 = 0, v2 = 0;
    if (i0 < ndw) v0 = base[i0];
    if (i1 < ndw) v1 = base[i1];
    if (three && i2 < ndw) v2 = base[i2];
    if (i0 == ndw - 1u) v0 &= tmask;
    if (i1 == ndw - 1u) v1 &= tmask;
    if (three && i2 == ndw - 1u) v2 &= tmask;
    const uint32_t nxt = ((lane + 1u) & 63u) << 2;
    uint32_t n0 = bperm(nxt, v0), n1 = bperm(nxt, v1);
    const uint32_t first1 = rdlane(v1, 0);
    n0 = lane == 63u ? first1 : n0;
    sa = __builtin_amdgcn_alignbit(n0, v0, sh);
    if (three) {
      uint32_t n2 = bperm(nxt, v2);
      const uint32_t first2 = rdlane(v2, 0);
      n1 = lane == 63u ? first2 : n1;
      sc = __builtin_amdgcn_alignbit(n2, v2, sh);  // (its last dword is never looked at)
    }
    sb = __builtin_amdgcn_alignbit(n1, v1, sh);     // (two vectors: its last dword is never looked at)
  }
  uint32_t sv = sa;  // the 64 dwords the current sixteen windows (and the four the pipeline is ahead) come from
  // a lane's view of window k: dwords 2k + (lane >> 5) and the one after, shifted by lane & 31
  uint32_t va = (lane >> 5) << 2;
  uint32_t g0A, g1A, x2A, r2A, Lr1A, e1A, L0A, sym0A, nl0A;
  uint32_t g0B, g1B, x2B, r2B, Lr1B, e1B, L0B, sym0B, nl0B;
  {  // fill the pipeline: windows 0 .. 3
    uint32_t ga0, ga1, gb0, gb1, gc0, gc1;
    SPEC_BPERM(ga0, va, sv, 0); SPEC_BPERM(ga1, va, sv, 4); SPEC_BPERM(gb0, va, sv, 8); SPEC_BPERM(gb1, va, sv, 12);
    SPEC_WAIT4(ga0, ga1, gb0, gb1);
    SPEC_BPERM(gc0, va, sv, 16); SPEC_BPERM(gc1, va, sv, 20); SPEC_BPERM(g0A, va, sv, 24); SPEC_BPERM(g1A, va, sv, 28);
    const uint32_t x0 = __builtin_amdgcn_alignbit(ga1, ga0, lane), x1 = __builtin_amdgcn_alignbit(gb1, gb0, lane);
    uint32_t r0, r1, e0;
    SPEC_RD16(r0, tree + ((x0 & 0xFFu) << 1)); SPEC_RD16(r1, tree + ((x1 & 0xFFu) << 1));
    SPEC_WAIT4(gc0, gc1, r0, r1);
    asm volatile("" : "+v"(g0A), "+v"(g1A));
    x2A = __builtin_amdgcn_alignbit(gc1, gc0, lane);
    SPEC_RD16(r2A, tree + ((x2A & 0xFFu) << 1));
    SPEC_RD16(e0, spec_entry_addr(tree, x0, r0)); SPEC_RD16(e1A, spec_entry_addr(tree, x1, r1));
    Lr1A = r1 & 15u;
    SPEC_WAIT4(r2A, e0, e1A, Lr1A);
    const uint32_t Lr0 = r0 & 15u;
    L0A = Lr0 > ROOT_BITS ? ROOT_BITS + (e0 & 15u) : Lr0; sym0A = e0 >> 4;
    // code length and symbol of every offset of the first windows: what the decoding wave walks into the chunk with
    lds_st8(hfirst + lane, L0A); lds_st8(hfirst + 64u + lane, sym0A);
    const uint32_t na = (lane + L0A) << 2;
    SPEC_BPERM(nl0A, na, L0A, 0);
    SPEC_WAIT4(nl0A, g0A, g1A, L0A);
    va += 24u;
  }
  uint32_t cnt = 0, e = entry, mlo = 0, mhi = 0, mcum = 0, j = 0;
#define SPEC_BODY(I, O) do { \
    va += 8u; SPEC_BPERM(g0##O, va, sv, 0); SPEC_BPERM(g1##O, va, sv, 4);                      /* window j + 4: stream bits */ \
    x2##O = __builtin_amdgcn_alignbit(g1##I, g0##I, lane);                                     /* window j + 3: root entries */ \
    { const uint32_t ra_ = tree + ((x2##O & 0xFFu) << 1); SPEC_RD16(r2##O, ra_); } \
    { const uint32_t ea_ = spec_entry_addr(tree, x2##I, r2##I); Lr1##O = r2##I & 15u; SPEC_RD16(e1##O, ea_); }   /* j + 2: deciding entries */ \
    L0##O = Lr1##I > ROOT_BITS ? ROOT_BITS + (e1##I & 15u) : Lr1##I; sym0##O = e1##I >> 4;     /* j + 1: lengths; ask for the next symbol's */ \
    { const uint32_t na_ = (lane + L0##O) << 2; SPEC_BPERM(nl0##O, na_, L0##O, 0); } \
    if (j + 1u < SPEC_FIRST) { lds_st8(hfirst + (j + 1u) * 128u + lane, L0##O); lds_st8(hfirst + (j + 1u) * 128u + 64u + lane, sym0##O); } \
    const bool in_ = lane + L0##I < 64u;                                                       /* window j: walk and store */ \
    const uint32_t Mv_ = in_ ? L0##I : 0u, Jv_ = L0##I + (in_ ? nl0##I : 0u); \
    uint64_t starts_; uint32_t woff_; \
    SPEC_WALK2(Mv_, Jv_, e, starts_, woff_); \
    const uint32_t rank_ = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts_, 0u)); \
    gu8* so_ = symout + cnt; \
    SPEC_STORE(starts_, rank_, sym0##I, so_); \
    /* lane j keeps window j's start mask and the literals before it (m0 is free in this function) */ \
    asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0" \
                 : "+v"(mlo), "+v"(mhi), "+v"(mcum) : "s"((uint32_t)starts_), "s"((uint32_t)(starts_ >> 32)), "s"(cnt), "s"(j)); \
    cnt += (uint32_t)__popcll(starts_); \
    e = woff_ - 64u; j++; \
    SPEC_WAIT5(g0##O, g1##O, r2##O, e1##O, nl0##O); \
  } while (0)
  _Pragma("nounroll") for (uint32_t q = 0; q < SPEC_WINDOWS / 16u; q++) {
    _Pragma("nounroll") for (uint32_t jj = 0; jj < 8u; jj++) {
      SPEC_BODY(A, B);
      SPEC_BODY(B, A);
    }
    // the next sixteen windows: 32 dwords further on
    if ((q & 1u) == 0u) {
      const uint32_t far = ((lane + 32u) & 63u) << 2;
      const uint32_t t0 = bperm(far, sa), t1 = bperm(far, sb);
      sv = lane < 32u ? t0 : t1;
    } else {
      sa = sb; sb = sc; sv = sa;
    }
    va -= 128u;
  }
#undef SPEC_BODY
  if (lane < SPEC_WINDOWS) {
    *reinterpret_cast<__attribute__((address_space(3))) uint64_t*>(&g_smem[slot + HL_MASK + lane * 8u]) = (uint64_t)mlo | ((uint64_t)mhi << 32);
    lds_st32(slot + HL_CUM + lane * 4u, mcum);
  }
  hw_st(slot, HW_N, cnt);
  hw_st(slot, HW_EXIT, e);
}

__device__ __noinline__ uint32_t scan_engine(const uint32_t me_);
namespace pe16 { __device__ __noinline__ uint32_t path_engine(const uint32_t me_); }   // one engine of sixteen waves, the lean form: no words of the static dictionary
namespace pe16g { __device__ __noinline__ uint32_t path_engine(const uint32_t me_); }  // ... the general form
namespace pe8 { __device__ __noinline__ uint32_t path_engine(const uint32_t me_); }    // two engines of eight, regions in turns
#ifdef BROTLI_AMD_GANG_KERNEL
namespace pe16r { __device__ __noinline__ uint32_t path_engine(const uint32_t me_); }  // one engine of sixteen a block, the blocks of a gang taking the regions in turns
#endif
// which command engine blocks of sixteen waves use: 0 = the path engine where it applies (brotli_path_engine.h), 1 = the scan
// engine only (experiments, A/B tests: BROTLI_AMD_ENGINE=scan)
__device__ uint32_t g_engine_mode = 2;


// ===================================== parse / copy split (context-modelled metablocks) =====================================
// A metablock whose literals depend on context cannot be parsed ahead of its output by more than the output allows: the
// tree of a literal is a function of the two bytes before it (decode.rs:2463-2551), which after a copy come out of the
// output.  What CAN leave the decoding wave is everything that moves bytes (decode.rs:2641-2720): in a block with helper
// waves the decoding wave only PARSES such a metablock (lean_split_commands) and leaves one 16-byte record per command in
// an LDS ring -- literals (bytes in a second ring), copy length, resolved distance -- and wave 1 of the block EXECUTES
// the records in order: literal stores, LZ77 loads and stores.  The parser never waits for a store; the two context
// bytes behind a copy are fetched from the copy's SOURCE (out[P - dist + n - 2 ..], asked for when the distance is
// known, used a command later) once the copier says that part of the output is in memory, else the parser waits for
// the copier to catch up.  Everything the lean loop does not take (limits, dictionary words, block switches) ends the
// split: the parser waits until the ring is drained and goes through the checked stages as before.
//   control words (CW_*, in the body of wave slot 3; the record ring is the body of slot 1, the literal ring of slot 2 --
//   the literal rounds that own these slots never run inside a context-modelled metablock):
//   records posted (parser), records / literal bytes taken out of the rings, records whose stores have completed and the
//   output position below which everything is in memory (copier), stop request / acknowledgement, output address.
enum { CW_HEAD = 0, CW_TAIL = 1, CW_LIT_TAIL = 2, CW_DONE_SEQ = 3, CW_DONE_P = 4, CW_STOP = 5, CW_OUT_LO = 6, CW_OUT_HI = 7, CW_WORDS = 8 };
enum { SPR_COPY = 0, SPR_SETP = 1 };  // record kinds: literals + copy; the copier's output position (the parser came back from the checked stages)
constexpr uint32_t SP_RECS = 64u, SP_LIT_BYTES = 1024u;
constexpr uint32_t SP_MAX_INSERT = 256u;      // longer literal runs go through the checked stages
constexpr uint32_t SP_MIN_MLEN = 4096u;       // smaller metablocks are not worth waking the copier for
constexpr uint32_t SP_SPIN_CAP = 1u << 24;    // polls of the parser before it gives the stream up (never seen)
static_assert(SP_RECS * 16u <= HL_SLOT - HL_MASK && SP_LIT_BYTES <= HL_SLOT - HL_MASK, "the rings live in the bodies of wave slots");
__device__ __forceinline__ uint32_t sp_rec_base() { return hc_ld(HC_BASE) + HL_SLOT + HL_MASK; }
__device__ __forceinline__ uint32_t sp_lit_base() { return hc_ld(HC_BASE) + 2u * HL_SLOT + HL_MASK; }
__device__ __forceinline__ uint32_t sp_ctl_base() { return hc_ld(HC_BASE) + 3u * HL_SLOT + HL_MASK; }
__device__ __forceinline__ uint32_t sp_ld(uint32_t ctl, uint32_t k) { return rfl(*reinterpret_cast<lds_vu32*>(&g_smem[ctl + 4u * k])); }
__device__ __forceinline__ void sp_st(uint32_t ctl, uint32_t k, uint32_t v) { if (lane_id() == 0) *reinterpret_cast<lds_vu32*>(&g_smem[ctl + 4u * k]) = v; }
#ifdef BROTLI_AMD_PROFILE_SPLIT
__device__ unsigned long long g_split_prof[40];
#define SPLIT_PROF(k, t0) do { if (blockIdx.x == 0 && lane_id() == 0) g_split_prof[k] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#define SPLIT_COUNT(k, v) do { if (blockIdx.x == 0 && lane_id() == 0) g_split_prof[k] += (v); } while (0)
#define SPLIT_T() __builtin_amdgcn_s_memtime()
#define SPLIT_LAP(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); lap_acc[k] += t_ - lap_t; lap_t = t_; } while (0)
#else
#define SPLIT_LAP(k) do { } while (0)
#define SPLIT_PROF(k, t0) do { (void)(t0); } while (0)
#define SPLIT_COUNT(k, v) do { } while (0)
#define SPLIT_T() 0ull
#endif


// ---- command records ahead of the parser (wave 2 of the block, rec_wave) ----
// What the parser's chain costs is its instruction count (one wave issues an instruction every ~8 clocks), and two thirds of
// a text stream's commands have no literals: command symbol, extra bits, distance symbol, extra bits -- nothing that depends
// on the output.  Wave 2 parses the command that WOULD start at every bit of the stream in front of the parser (64 positions
// an instruction, ReadCommandInternal + ReadDistanceInternal per lane: sc_head / sc_dist of the scan engine) into a ring of
// 8-byte records indexed by bit position; the parser takes the record at its position instead of parsing (a command with
// literals: its head only) and parses by hand wherever there is no usable record.  A record is the exact parse of its
// position or marked unusable -- nothing depends on a guess.
//   record: word 0 = copy length (16 bits) | bits of the command (head, or head + distance) << 16 | flags; word 1 = insert
//   length (commands with literals), else the distance (explicit) or the distance symbol (ring codes 0..15).
constexpr uint32_t SPX_POS = 1024u;                   // ring size in stream bits (at most: see XW_RING_MASK)
constexpr uint32_t SPX_POS_MIN = 512u;                // ... and at least: a metablock whose tables leave less room than that has no records
constexpr uint32_t SPX_CTL_BYTES = 256u;
constexpr uint32_t SPX_BYTES = SPX_CTL_BYTES + SPX_POS * 8u;
enum { XW_FRONT = 0 /* u64: positions below are written (low), for parameter epoch (high) */, XW_POS = 2 /* the parser's position (lags) */,
       XW_EPOCH = 3, XW_STOP = 4, XW_ORIGIN_DW = 5, XW_LIMIT = 6, XW_CMD_TREE = 7, XW_DT0 = 8, XW_POSTFIX = 12, XW_NUM_DIRECT = 13,
       XW_DICT_LO = 14, XW_DICT_HI = 15 /* the static dictionary (device address) */, XW_RING_MASK = 16 /* the ring's positions less one: SPX_POS - 1, or half of that where the arena has no more room (round 6) */, XW_WORDS = 17 };
enum : uint32_t { XR_VALID = 1u << 23, XR_LITERALS = 1u << 24, XR_IMPLICIT = 1u << 25, XR_DCTX_SHIFT = 26, XR_SHORT = 1u << 28,
       XR_NOT_RUN = 1u << 31 /* not a record the hand-written run takes: not valid, or a copy of more than 63 bytes (its tests are one sign test) */ };
__device__ __noinline__ void rec_wave();

// Wave 1 of the block while the decoding wave parses a context-modelled metablock.
__device__ __noinline__ void copier_wave() {
  const uint32_t lane = lane_id();
  const uint32_t rec = sp_rec_base(), lit = sp_lit_base(), ctl = sp_ctl_base();
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)sp_ld(ctl, CW_OUT_LO) | ((uint64_t)sp_ld(ctl, CW_OUT_HI) << 32));
  uint64_t P = 0;
  uint32_t tail = 0, lit_tail = 0, idle = 0;
  bool quiet = false;
  for (;;) {
    const uint32_t head = sp_ld(ctl, CW_HEAD);
    if (head == tail) {
      if (!quiet) {  // nothing to do: everything so far is in memory once the stores have been acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sp_st(ctl, CW_DONE_P, (uint32_t)P);
        lds_release();
        sp_st(ctl, CW_DONE_SEQ, tail);
        quiet = true;
      }
      if (sp_ld(ctl, CW_STOP) != 0u) break;
      // (a SIMD issues one scalar instruction every four clocks for ALL its waves, and the parser of another block lives on
      // this one: a poll every 64 clocks took a third of them)
      idle++;
      if (idle < 16u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(16);
      continue;
    }
    const uint64_t busy_t0 = SPLIT_T();
    idle = 0; quiet = false;
    lds_acquire();
    while (tail != head) {
      uint32_t w = 0;
      if (lane < 4u) w = lds_ld32(rec + ((tail & (SP_RECS - 1u)) << 4) + 4u * lane);
      const uint32_t w0 = rdlane(w, 0), w1 = rdlane(w, 1), w2 = rdlane(w, 2), w3 = rdlane(w, 3);
      tail++;
      if ((w0 & 0xFFu) == (uint32_t)SPR_SETP) {
        P = (uint64_t)w1 | ((uint64_t)w2 << 32);
        sp_st(ctl, CW_TAIL, tail);
        continue;
      }
      const uint32_t ins = w0 >> 8;
      for (uint32_t k = 0; k < ins; k += 64u) {
        if (k + lane < ins) { const uint32_t b = lds_ld8(lit + ((w3 + k + lane) & (SP_LIT_BYTES - 1u))); out[P + k + lane] = (uint8_t)b; }
      }
      P += ins;
      lit_tail += ins;
      if (lane == 0) {  // the ring entries are free again (LDS operations of a wave execute in order: the reads above are done)
        *reinterpret_cast<lds_vu32*>(&g_smem[ctl + 4u * CW_LIT_TAIL]) = lit_tail;
        *reinterpret_cast<lds_vu32*>(&g_smem[ctl + 4u * CW_TAIL]) = tail;
      }
      // ---- the copy: chunks of up to 64 lanes; a chunk reads what lies `span` bytes before it, a multiple of the distance that
      // is at least the chunk's length (a copy that overlaps itself repeats its first `dist` bytes: decode.rs:2641-2720) ----
      uint32_t rem = w1, span = w2, done = 0;
      const uint32_t dist = w2;
      uint64_t dst = P;
      bool first = true;
      while (rem != 0u) {
        uint32_t bytes;
        if (span >= 64u && rem >= 16u) {
          uint32_t n16 = rem >> 4;
          if (n16 > 64u) n16 = 64u;
          if (n16 > (span >> 4)) n16 = span >> 4;
          bytes = n16 << 4;
          u32x4 v = {0, 0, 0, 0};
          if (lane < n16) v = *reinterpret_cast<gu32x4*>(out + dst - span + (uint64_t)lane * 16u);
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
          if (first) { sp_st(ctl, CW_DONE_P, (uint32_t)P); sp_st(ctl, CW_DONE_SEQ, tail - 1u); first = false; }
          if (lane < n16) *reinterpret_cast<gu32x4*>(out + dst + (uint64_t)lane * 16u) = v;
        } else {
          bytes = rem < 64u ? rem : 64u;
          if (bytes > span) bytes = span;
          uint32_t b = 0;
          if (lane < bytes) b = (out + dst - span)[lane];
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(b) :: "memory");
          if (first) { sp_st(ctl, CW_DONE_P, (uint32_t)P); sp_st(ctl, CW_DONE_SEQ, tail - 1u); first = false; }
          if (lane < bytes) (out + dst)[lane] = (uint8_t)b;
        }
        dst += bytes; rem -= bytes; done += bytes;
        while (span < 64u && done + dist >= 2u * span) span <<= 1;
      }
      P = dst;
      SPLIT_COUNT(12, 1);
    }
    SPLIT_PROF(5, busy_t0);
  }
  lds_release();
  sp_st(ctl, CW_STOP, 2u);
}

__device__ __noinline__ void helper_wave(const uint32_t me /* 1 .. waves - 1 */, gu8* scratch_sym) {
  const uint32_t lane = lane_id();
  const uint32_t slot = hc_ld(HC_BASE) + me * HL_SLOT;
  gu8* const mine = scratch_sym + me * SPEC_SLOT_BYTES;
  uint32_t seq = 0;
  for (;;) {
    uint32_t j;
    for (uint32_t idle = 0;; idle++) {  // idle until the decoding wave posts a round (it always posts `exit` at the end)
      j = hc_ld(HC_SEQ);
      if (j != seq) break;
      // right after a round the next one is due any moment; otherwise look every 8 K clocks only (a stream without
      // long literal runs never posts one, and seven waves polling at full rate are a billion instructions per launch)
      if (idle < 256u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(127);
    }
    seq = j;
    lds_acquire();
    const uint32_t kind = hc_ld(HC_KIND);
    if (kind == HK_SCAN) { scan_engine(me); continue; }  // every wave of the block runs the command engine
    if (kind == HK_PATH) { seq = rfl(pe16::path_engine(me)); continue; }   // (back with the last request it answered: see there)
    if (kind == HK_PATHG) { seq = rfl(pe16g::path_engine(me)); continue; }
    if (kind == HK_PATH2) { seq = rfl(pe8::path_engine(me)); continue; }
#ifdef BROTLI_AMD_GANG_KERNEL
    if (kind == HK_PATHR) { seq = rfl(pe16r::path_engine(me)); continue; }
#endif
    if (kind == HK_SPLIT) {  // a context-modelled metablock: wave 1 copies (if asked to), wave 2 parses command records; the others go back to sleep
      if (rfl(me) == 1u && (g_engine_mode & 2u) == 0u) copier_wave(); else if (rfl(me) == 2u && hc_ld(HC_EXT_BASE) != 0u) rec_wave();
      continue;
    }
    if (kind != HK_ROUND) return;
    if (rfl(me) >= hc_ld(HC_NW)) continue;                // (rounds are for the first eight waves of a block)
    spec_chunk(me, hc_ld(HC_DW0) + me * (SPEC_WINDOWS * 2u), hc_ld(HC_SHIFT), hc_ld(HC_TREE), mine, 0u);
    lds_release();
    hw_st(slot, HW_DONE, seq);
    // Where does the chain of the chunk before meet this chunk's?  If that chunk's own chain is the stream's (the
    // decoding wave checks that, chunk by chunk), its exit offset is where the stream's chain enters this chunk: every
    // helper walks it into its own first windows at the same time.  (Capped rounds: the decoding wave does it all.)
    uint64_t own_mask[SPEC_FIRST] = {}; uint32_t own_sym[SPEC_FIRST] = {}, own_cnt[SPEC_FIRST] = {}, own_win = 0;
    const bool self = hc_ld(HC_CAPPED) == 0u;
    bool go = true;
    if (self) {
      while (hw_ld(slot - HL_SLOT, HW_DONE) != seq) {
        if (hc_ld(HC_SEQ) != seq) { go = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!go) continue;
      lds_acquire();
      uint32_t e = hw_ld(slot - HL_SLOT, HW_EXIT), nm = 0, skip = 0;
      bool synced = false;
      _Pragma("unroll") for (uint32_t k = 0; k < SPEC_FIRST; k++) {
        if (synced) continue;
        const uint64_t sm = *reinterpret_cast<__attribute__((address_space(3))) const uint64_t*>(&g_smem[slot + HL_MASK + k * 8u]);
        const uint64_t smask = ((uint64_t)rfl((uint32_t)(sm >> 32)) << 32) | rfl((uint32_t)sm);
        const uint32_t Lv = lds_ld8(slot + HL_FIRST + k * 128u + lane);
        own_sym[k] = lds_ld8(slot + HL_FIRST + k * 128u + 64u + lane);
        uint64_t tstarts; uint32_t woff;
        SPEC_WALK(Lv, e, tstarts, woff);
        const uint64_t common = tstarts & smask;
        uint64_t m = tstarts;
        if (common) {
          const uint32_t p = (uint32_t)__builtin_ctzll(common);
          m = tstarts & ((1ull << p) - 1ull);
          skip = rfl(lds_ld32(slot + HL_CUM + k * 4u)) + (uint32_t)__popcll(smask & ((1ull << p) - 1ull));
          synced = true;
        }
        own_mask[k] = m; own_cnt[k] = nm; nm += (uint32_t)__popcll(m);
        e = woff - 64u; own_win = k + 1u;
      }
      hw_st(slot, HW_RSYNC, synced ? 1u : 0u); hw_st(slot, HW_RNM, nm); hw_st(slot, HW_RSKIP, skip); hw_st(slot, HW_RWIN, own_win); hw_st(slot, HW_REXIT, e);
      lds_release();
      hw_st(slot, HW_RES, seq);
    }
    // the decoding wave says which of the chunk's literals are the stream's and where they go (none: n = 0)
    while (hw_ld(slot, HW_MVGO) != seq) {
      if (hc_ld(HC_SEQ) != seq) { go = false; break; }  // the round was given up (or the kernel is about to end)
      __builtin_amdgcn_s_sleep(2);
    }
    if (!go) continue;
    lds_acquire();
    const uint32_t n = hw_ld(slot, HW_MVN);
    if (self && hw_ld(slot, HW_MVOWN) != 0u) {  // the literals of the walk above are the stream's: in front of the chunk's own
      gu8* const d0 = (gu8*)(uintptr_t)(((uint64_t)hc_ld(HC_OUT_LO) | ((uint64_t)hc_ld(HC_OUT_HI) << 32)) +
                                       ((uint64_t)hw_ld(slot, HW_MVDST_LO) | ((uint64_t)hw_ld(slot, HW_MVDST_HI) << 32)));
      _Pragma("unroll") for (uint32_t k = 0; k < SPEC_FIRST; k++) {
        if (k >= own_win) continue;
        const uint64_t m = own_mask[k];
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        gu8* wq = d0 + own_cnt[k];
        SPEC_STORE(m, rank, own_sym[k], wq);
      }
    }
    if (n != 0) {
      gu8* const src = mine + hw_ld(slot, HW_MVSRC);
      gu8* const dst = (gu8*)(uintptr_t)(((uint64_t)hc_ld(HC_OUT_LO) | ((uint64_t)hc_ld(HC_OUT_HI) << 32)) +
                                        ((uint64_t)hw_ld(slot, HW_MVDST_LO) | ((uint64_t)hw_ld(slot, HW_MVDST_HI) << 32))) +
                        (self ? hw_ld(slot, HW_RNM) : 0u);
      constexpr uint32_t H = SPEC_SLOT_BYTES / 1024u;
      const uint32_t n16 = n >> 4;
      u32x4 t[H] = {};
      _Pragma("unroll") for (uint32_t h = 0; h < H; h++) {
        const uint32_t c = lane + 64u * h;
        if (c < n16) t[h] = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16);
      }
      uint32_t tail = 0;
      if (lane < (n & 15u)) tail = src[(n16 << 4) + lane];
      _Pragma("unroll") for (uint32_t h = 0; h < H; h++) {
        const uint32_t c = lane + 64u * h;
        if (c < n16) *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = t[h];
      }
      if (lane < (n & 15u)) dst[(n16 << 4) + lane] = (uint8_t)tail;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in memory before the move is reported done
    lds_release();
    hw_st(slot, HW_MVDONE, seq);
  }
}

// One hand-scheduled pass of the literal batch loop (operands: see its two users).
#define LITERAL_BATCH_ASM \
  "s_nop 4\n" \
  "1:\n\t" \
  "s_cmp_gt_u32 %[cnt], 31\n\t" \
  "s_cbranch_scc1 2f\n\t" \
  "s_sub_u32 %[t0], %[ndw], %[cb]\n\t" \
  "v_readlane_b32 s90, %[cur], %[t0]\n\t" \
  "s_mov_b32 s91, 0\n\t" \
  "s_lshl_b64 vcc, s[90:91], %[cnt]\n\t" \
  "s_or_b64 %[buf], %[buf], vcc\n\t" \
  "s_add_u32 %[cnt], %[cnt], 32\n\t" \
  "s_add_u32 %[ndw], %[ndw], 1\n" \
  "2:\n\t" \
  "s_sub_u32 %[t0], %[ndw], %[cb]\n\t" \
  "s_add_u32 %[t1], %[t0], 1\n\t" \
  "v_readlane_b32 s90, %[cur], %[t0]\n\t" \
  "v_readlane_b32 s91, %[cur], %[t1]\n\t" \
  "s_lshl_b64 vcc, s[90:91], %[cnt]\n\t" \
  "s_or_b64 s[92:93], %[buf], vcc\n\t" \
  "s_sub_u32 %[t0], 64, %[cnt]\n\t" \
  "s_lshr_b64 s[94:95], s[90:91], %[t0]\n\t" \
  "v_mov_b32 %[v0], s92\n\t" \
  "v_mov_b32 %[v1], s93\n\t" \
  "v_alignbit_b32 %[v0], s93, %[v0], %[lane]\n\t" \
  "v_alignbit_b32 %[v1], s94, %[v1], %[lane]\n\t" \
  "v_bfi_b32 %[v0], %[lomask], %[v0], %[v1]\n\t" \
  "v_and_b32 %[v1], 0xff, %[v0]\n\t" \
  "v_lshl_add_u32 %[v1], %[v1], 1, %[tree]\n\t" \
  "ds_read_u16 %[v2], %[v1]\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_and_b32 %[v3], 15, %[v2]\n\t" \
  "v_cmp_lt_u32 vcc, 8, %[v3]\n\t" \
  "s_cbranch_vccz 3f\n\t" \
  "s_and_saveexec_b64 s[94:95], vcc\n\t" \
  "v_lshrrev_b32 %[v1], 8, %[v0]\n\t" \
  "v_add_u32 %[v3], -8, %[v3]\n\t" \
  "v_bfe_u32 %[v1], %[v1], 0, %[v3]\n\t" \
  "v_lshrrev_b32 %[v4], 4, %[v2]\n\t" \
  "v_add_u32 %[v1], %[v1], %[v4]\n\t" \
  "v_lshl_add_u32 %[v1], %[v1], 1, %[tree]\n\t" \
  "ds_read_u16 %[v2], %[v1]\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_and_b32 %[v3], 15, %[v2]\n\t" \
  "v_add_u32 %[v3], 8, %[v3]\n\t" \
  "s_mov_b64 exec, s[94:95]\n" \
  "3:\n\t" \
  "s_mul_i32 %[t0], %[i], 15\n\t" \
  "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t" \
  "s_nop 1\n\t" \
  "v_cndmask_b32 %[v3], 64, %[v3], vcc\n\t" \
  "s_mov_b64 s[92:93], 0\n\t" \
  "s_mov_b32 %[off], 0xffffffc0\n\t" \
  /* two symbols per step: ds_bpermute hands every offset the length of the symbol after its own; v4 = length where */ \
  /* that symbol still starts in the window, else 0; v1 = both lengths, else the one.  (Measured against one per step */ \
  /* for runs below a threshold: two per step is faster from three literals on, which is every batch.) */ \
  "v_add_u32 %[v4], %[v3], %[lane]\n\t" \
  "v_lshlrev_b32 %[v1], 2, %[v4]\n\t" \
  "ds_bpermute_b32 %[v1], %[v1], %[v3]\n\t" \
  "v_cmp_gt_u32 vcc, 64, %[v4]\n\t" \
  "s_nop 1\n\t" \
  "v_cndmask_b32 %[v4], 0, %[v3], vcc\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_cndmask_b32 %[v1], 0, %[v1], vcc\n\t" \
  "v_add_u32 %[v1], %[v1], %[v3]\n\t" \
  "s_nop 0\n" \
  "9:\n\t" \
  "v_readlane_b32 %[t1], %[v4], %[off]\n\t" \
  "v_readlane_b32 %[t0], %[v1], %[off]\n\t" \
  "s_bitset1_b64 s[92:93], %[off]\n\t" \
  "s_add_u32 %[t1], %[t1], %[off]\n\t" \
  "s_bitset1_b64 s[92:93], %[t1]\n\t" \
  "s_add_u32 %[off], %[off], %[t0]\n\t" \
  "s_cbranch_scc0 9b\n" \
  "10:\n\t" \
  "s_add_u32 %[off], %[off], 64\n\t" \
  "s_bcnt1_i32_b64 %[n], s[92:93]\n\t" \
  "v_mbcnt_lo_u32_b32 %[v4], s92, 0\n\t" \
  "v_mbcnt_hi_u32_b32 %[v4], s93, %[v4]\n\t" \
  "s_cmp_le_u32 %[n], %[i]\n\t" \
  "s_cbranch_scc1 5f\n\t" \
  "v_cmp_eq_u32 vcc, %[i], %[v4]\n\t" \
  "s_and_b64 vcc, vcc, s[92:93]\n\t" \
  "s_ff1_i32_b64 %[off], vcc\n\t" \
  "s_lshl_b64 vcc, -1, %[off]\n\t" \
  "s_andn2_b64 s[92:93], s[92:93], vcc\n\t" \
  "s_mov_b32 %[n], %[i]\n" \
  "5:\n\t" \
  "v_lshrrev_b32 %[v2], 4, %[v2]\n\t" \
  "v_add_u32 %[v4], %[woff], %[v4]\n\t" \
  "s_mov_b64 exec, s[92:93]\n\t" \
  "global_store_byte %[v4], %[v2], %[wp]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "s_cmp_le_u32 %[off], %[cnt]\n\t" \
  "s_cbranch_scc1 6f\n\t" \
  "s_sub_u32 %[t0], %[off], %[cnt]\n\t" \
  "s_lshr_b64 %[buf], s[90:91], %[t0]\n\t" \
  "s_sub_u32 %[cnt], 64, %[t0]\n\t" \
  "s_add_u32 %[ndw], %[ndw], 2\n\t" \
  "s_branch 7f\n" \
  "6:\n\t" \
  "s_lshr_b64 %[buf], %[buf], %[off]\n\t" \
  "s_sub_u32 %[cnt], %[cnt], %[off]\n" \
  "7:\n\t" \
  "s_add_u32 %[woff], %[woff], %[n]\n\t" \
  "s_sub_u32 %[i], %[i], %[n]\n\t" \
  "s_cmp_lt_u32 %[i], 3\n\t" \
  "s_cbranch_scc1 8f\n\t" \
  "s_sub_u32 %[t0], %[ndw], %[cb]\n\t" \
  "s_cmp_lt_u32 %[t0], 62\n\t" \
  "s_cbranch_scc1 1b\n" \
  "8:\n\t" \
  "s_nop 4\n"

// ===================================== lean command loop =====================================
// The common case of the command loop as a function of its own: commands of a metablock whose literal block types
// all have a constant context map, while every stage stays clear of the limits (see `quota` in process_commands).
// It owns only the state such commands touch -- its register allocation is not shared with the checked stages and
// their error handling -- and hands a command over at the first stage it cannot take (L_STAGE); process_commands
// finishes that command and comes back.  State crosses through LDS_LEAN (uniform values) and function arguments.
enum { L_BUF_LO, L_BUF_HI, L_CNT, L_NEXT_DW, L_ISSUED, L_END_DW, L_P_LO, L_P_HI, L_QUOTA, L_MLEN, L_BL0, L_BL1, L_BL2, L_D0, L_D1, L_D2,
       L_D3, L_NCMD_LO, L_NCMD_HI, L_CMD_TREE, L_LIT_TREE, L_DT0, L_DT1, L_DT2, L_DT3, L_MAX_BACKWARD, L_POSTFIX, L_NUM_DIRECT, L_OUT_LO,
       L_OUT_HI, L_INSERT, L_COPY, L_DCODE, L_DCTX, L_LITS_LEFT, L_P1, L_P2, L_CTX_REGS, L_TRIVIAL, L_CTX_LUT, L_CHUNK_BASE, L_SPEC_LO, L_SPEC_HI, L_COUNT };
static_assert(L_COUNT * 4 <= 192, "LDS_LEAN too small");
// (parse / copy split and command records, see copier_wave / rec_wave)
enum { L_SP_HEAD = L_COUNT, L_SP_LIT = L_COUNT + 1, L_SP_ORIGIN = L_COUNT + 2 /* dword the record ring's positions count from */,
       L_SP_REC = L_COUNT + 3 /* command records (rec_wave) in use */, L_SP_WAITED = L_COUNT + 4 /* why the record loop left: 1 wave 2's records were not there yet, 2 a command with more than 63 literals or a copy of more than 63 bytes, 0 anything else */,
       L_SP_COUNT = L_COUNT + 5 };
static_assert(L_SP_COUNT * 4 <= 192, "LDS_LEAN too small");
enum { LS_BEGIN = 0, LS_AFTER_HEAD = 1, LS_LITERALS_REST = 2, LS_DISTANCE = 3, LS_POST_DISTANCE = 4, LS_COMMAND_DONE = 5,
       LS_LITERALS_AT_LIMIT = 6, LS_NEEDS_INPUT = 7, LS_LITERAL_ROUNDS = 8 };
#define LEAN_LD(k) rfl(lds_ld32(LDS_LEAN + 4u * (uint32_t)(k)))
#define LEAN_ST(k, v) lds_st32(LDS_LEAN + 4u * (uint32_t)(k), (uint32_t)(v))

// Rounds of a long literal run (see the helper waves above); state through LDS_LEAN like the lean function's.
#ifdef BROTLI_AMD_PROFILE_LEAN
__device__ unsigned long long g_lean_prof[8];
#define LEAN_PROF(k) do { uint64_t _t = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0) lp_acc[k] += _t - lp_t; lp_t = _t; } while (0)
#else
#define LEAN_PROF(k) do { } while (0)
#endif
#ifdef BROTLI_AMD_PROFILE_SPEC
__device__ unsigned long long g_spec_prof[8];
#define SPEC_PROF(k) do { uint64_t _t = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && lane == 0) g_spec_prof[k] += _t - sp_t; sp_t = _t; } while (0)
#define SPEC_COUNT(k, v) do { if (blockIdx.x == 0 && lane == 0) g_spec_prof[k] += (v); } while (0)
#else
#define SPEC_PROF(k) do { } while (0)
#define SPEC_COUNT(k, v) do { } while (0)
#endif
// position of the n-th (0-based) set bit of a mask that has more than n
__device__ __forceinline__ uint32_t nth_set_bit(uint64_t m, uint32_t n) {
  const uint32_t lane = lane_id();
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  return (uint32_t)__builtin_ctzll(__ballot(((m >> lane) & 1ull) != 0 && rank == n));
}
// bit offset inside chunk w at which the t-th literal of its chain starts (t < the chunk's literal count)
__device__ __noinline__ uint32_t spec_locate(uint32_t w, uint32_t t) {
  const uint32_t lane = lane_id();
  w = rfl(w); t = rfl(t);
  const uint32_t slot = hc_ld(HC_BASE) + w * HL_SLOT;
  const uint32_t cum = lane < SPEC_WINDOWS ? lds_ld32(slot + HL_CUM + lane * 4u) : 0xFFFFFFFFu;
  const uint32_t k = (uint32_t)__popcll(__ballot(cum <= t)) - 1u;  // the window the literal starts in
  const uint64_t sm = *reinterpret_cast<__attribute__((address_space(3))) const uint64_t*>(&g_smem[slot + HL_MASK + k * 8u]);
  const uint64_t m = ((uint64_t)rfl((uint32_t)(sm >> 32)) << 32) | rfl((uint32_t)sm);
  return k * 64u + nth_set_bit(m, t - rdlane(cum, k));
}

__device__ __noinline__ void spec_rounds(uint32_t tree_addr) {
  const uint32_t lane = lane_id();
  tree_addr = rfl(tree_addr);
  BitReader br;
  br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
  br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED); br.end_dw = LEAN_LD(L_END_DW);
  br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)LEAN_LD(L_OUT_LO) | ((uint64_t)LEAN_LD(L_OUT_HI) << 32));
  gu8* const spec = (gu8*)(uintptr_t)((uint64_t)LEAN_LD(L_SPEC_LO) | ((uint64_t)LEAN_LD(L_SPEC_HI) << 32));
  uint64_t P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
  uint32_t i = LEAN_LD(L_LITS_LEFT);
  const uint32_t safe_dw = br.end_dw > 72u ? br.end_dw - 72u : 0u;
  const uint32_t hb = hc_ld(HC_BASE), nw = hc_ld(HC_NW);
  hc_st(HC_OUT_LO, (uint32_t)(uintptr_t)out); hc_st(HC_OUT_HI, (uint32_t)((uint64_t)(uintptr_t)out >> 32));
  uint32_t seq = hc_ld(HC_SEQ);
  uint32_t moving = 0, mv_seq = 0;  // helpers whose move (of round mv_seq) has not been seen finished
  // no round holds more literals than its bits divided by the shortest code length of the tree (root entries tell)
  uint32_t round_max;
  {
    uint32_t m = 9u;
    for (uint32_t k = 0; k < 4u; k++) { const uint32_t L = lds_ld16(tree_addr + ((lane + 64u * k) << 1)) & 15u; m = L < m ? L : m; }
    uint32_t minlen = 9u;
    for (uint32_t k = 1; k < 9u; k++) if (__ballot(m == k) != 0ull) { minlen = k; break; }
    round_max = nw * SPEC_WINDOWS * 64u / minlen + 64u;
  }
  while (i >= SPEC_ROUND_MIN && br.next_dw + spec_input_dwords(nw) < safe_dw && hc_ld(HC_KIND) != 3u) {
#ifdef BROTLI_AMD_PROFILE_SPEC
    uint64_t sp_t = __builtin_amdgcn_s_memtime();
#endif
    // A round decodes up to round_max literals.  Where the run has fewer left, the round is
    // *capped*: nothing goes to the output directly (chunk 0 uses scratch slot 0 like the helpers), and the round ends
    // at the run's last literal, whose bit position the start masks give.
    const uint32_t cap = i;
    const bool capped = i < round_max;
    const uint64_t run_pos = br.pos();
    const uint64_t abs0 = run_pos + BitReader::skip_bits();
    const uint32_t dw0 = (uint32_t)(abs0 >> 5), sh = (uint32_t)abs0 & 31u;
    seq++;
    hc_st(HC_DW0, dw0); hc_st(HC_SHIFT, sh); hc_st(HC_TREE, tree_addr); hc_st(HC_KIND, 1); hc_st(HC_CAPPED, capped ? 1u : 0u);
    lds_release();
    hc_st(HC_SEQ, seq);
    br.request_ahead(dw0 + nw * SPEC_WINDOWS * 2u);  // where the round will normally end: there by the time it does
    spec_chunk(0, dw0, sh, tree_addr, capped ? spec : out + P, 0);  // the first chunk: its first bit does start a literal
    lds_release();
    hw_st(hb, HW_DONE, seq);     // (helper 1 takes this chunk's exit offset from here)
    SPEC_PROF(0);
    uint32_t acc = hw_ld(hb, HW_N);  // literals of the round so far
    uint32_t e = hw_ld(hb, HW_EXIT);
    uint32_t bits_done = SPEC_WINDOWS * 64u + e;
    uint32_t own = 0;            // literals of chunk 0 to move out of scratch slot 0 (capped rounds)
    bool full = false;           // the run's last literal has been reached
    if (capped) {
      if (acc >= cap) { bits_done = acc == cap ? bits_done : spec_locate(0, cap); acc = cap; full = true; }
      own = acc;
    }
    bool lost = false;
    {  // lane w looks at helper w's words: round decoded and, in uncapped rounds, its account of how the chain enters its
       // chunk (bounded: helpers that never answer must not hang the GPU)
      const bool helper_lane = lane >= 1u && lane < nw;
      lds_vu32* const flag = reinterpret_cast<lds_vu32*>(&g_smem[hb + (helper_lane ? lane : 0u) * HL_SLOT + HL_CTL + 4u * (capped ? HW_DONE : HW_RES)]);
      uint32_t polls = 0;
      while (__ballot(helper_lane && *flag != seq) != 0ull) { if (++polls > (1u << 20)) { lost = true; break; } __builtin_amdgcn_s_sleep(1); }
    }
    lds_acquire();
    SPEC_PROF(1);
    if (!lost) { moving = 0; mv_seq = seq; }  // (every helper has finished the move of the round before: it decoded this round's chunk after it)
    bool open = !lost && !full;  // chunks are still being taken
    if (!capped && open) {
      // Every helper has walked the chain of the chunk before into its own chunk.  Up to the first chunk that did not
      // fall in, that chain was the stream's, so the accounts stand: lane w adds up helper w's and posts its move.
      const bool helper_lane = lane >= 1u && lane < nw;
      const uint32_t slot = hb + (helper_lane ? lane : 0u) * HL_SLOT + HL_CTL;
      const uint32_t rnm = lds_ld32(slot + 4u * HW_RNM), rsync = lds_ld32(slot + 4u * HW_RSYNC), rskip = lds_ld32(slot + 4u * HW_RSKIP);
      const uint32_t cn = lds_ld32(slot + 4u * HW_N), cexit = lds_ld32(slot + 4u * HW_EXIT), rwin = lds_ld32(slot + 4u * HW_RWIN), rexit = lds_ld32(slot + 4u * HW_REXIT);
      const uint64_t unsynced = __ballot(helper_lane && rsync == 0u);
      const uint32_t u = unsynced ? (uint32_t)__builtin_ctzll(unsynced) : nw;  // first chunk that did not fall in
      const uint32_t valid = lane < u ? cn - rskip : 0u;
      const uint32_t contrib = lane == 0u ? acc : helper_lane && lane <= u ? rnm + valid : 0u;
      uint32_t before = 0, total = 0;  // exclusive prefix sum over lanes 0 .. nw - 1
      for (uint32_t k = 0; k < nw; k++) { before = lane == k ? total : before; total += rdlane(contrib, k); }
      acc = total;
      bits_done = u >= nw ? nw * SPEC_WINDOWS * 64u + rdlane(cexit, nw - 1u) : (u * SPEC_WINDOWS + rdlane(rwin, u)) * 64u + rdlane(rexit, u);
      if (helper_lane) {
        const uint64_t d = P + before;
        lds_st32(slot + 4u * HW_MVSRC, rskip); lds_st32(slot + 4u * HW_MVDST_LO, (uint32_t)d); lds_st32(slot + 4u * HW_MVDST_HI, (uint32_t)(d >> 32));
        lds_st32(slot + 4u * HW_MVN, valid); lds_st32(slot + 4u * HW_MVOWN, lane <= u ? 1u : 0u);
      }
      lds_release();
      if (helper_lane) *reinterpret_cast<lds_vu32*>(&g_smem[slot + 4u * HW_MVGO]) = seq;
      moving = (uint32_t)__ballot(helper_lane && lane <= u);
      open = false;
    }
    for (uint32_t w = 1; w < nw; w++) {
      if (!capped) break;
      const uint32_t slot = hb + w * HL_SLOT;
      uint32_t mv_src = 0, mv_n = 0, mv_dst = acc, mv_own = 0;
      if (open) {
        // walk the true chain into chunk w until it steps on a start the helper marked too
        bool synced = false;
        uint32_t skip = 0;
        for (uint32_t j = 0; j < SPEC_FIRST && !synced && !full; j++) {
          const uint64_t sm = *reinterpret_cast<__attribute__((address_space(3))) const uint64_t*>(&g_smem[slot + HL_MASK + j * 8u]);
          const uint64_t smask = ((uint64_t)rfl((uint32_t)(sm >> 32)) << 32) | rfl((uint32_t)sm);
          const uint32_t Lv = lds_ld8(slot + HL_FIRST + j * 128u + lane), sv = lds_ld8(slot + HL_FIRST + j * 128u + 64u + lane);
          uint64_t tstarts; uint32_t woff;
          SPEC_WALK(Lv, e, tstarts, woff);
          const uint64_t common = tstarts & smask;
          uint64_t mine = tstarts;  // literals of this window that only the true chain has
          if (common) {
            const uint32_t p = (uint32_t)__builtin_ctzll(common);
            mine = tstarts & ((1ull << p) - 1ull);
            skip = rfl(lds_ld32(slot + HL_CUM + j * 4u)) + (uint32_t)__popcll(smask & ((1ull << p) - 1ull));
            synced = true;
          }
          uint32_t nm = (uint32_t)__popcll(mine);
          if (capped && acc + nm >= cap) {  // the run ends among them
            if (acc + nm > cap || !synced) {
              // (ends exactly with the window's last own literal and no common start: the position after it is what
              // the walk left the window at, unless the chain goes on inside this window -- then it is the next start)
              const uint32_t n = cap - acc;
              if (n < nm) { const uint32_t pos = nth_set_bit(mine, n); mine &= (1ull << pos) - 1ull; bits_done = (w * SPEC_WINDOWS + j) * 64u + pos; }
              else bits_done = (w * SPEC_WINDOWS + j + 1u) * 64u + (woff - 64u);
              nm = n; full = true; synced = false;
            } else {
              // ends exactly in front of the common start
              bits_done = (w * SPEC_WINDOWS + j) * 64u + (uint32_t)__builtin_ctzll(common);
              full = true; synced = false;
            }
          }
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
          gu8* wq = out + P + acc;
          SPEC_STORE(mine, rank, sv, wq);
          acc += nm;
          if (!synced && !full) { e = woff - 64u; bits_done = (w * SPEC_WINDOWS + j + 1u) * 64u + e; }
        }
        if (!synced) open = false;  // no common start within the first windows (or the run is complete): the round ends here
        else {
          // the helper's literals from the common start on are the stream's: it moves them into place itself
          uint32_t valid = hw_ld(slot, HW_N) - skip;
          e = hw_ld(slot, HW_EXIT);
          bits_done = (w + 1u) * SPEC_WINDOWS * 64u + e;
          if (capped && acc + valid >= cap) {
            if (acc + valid > cap) { valid = cap - acc; bits_done = w * SPEC_WINDOWS * 64u + spec_locate(w, skip + valid); }
            full = true; open = false;
          }
          mv_src = skip; mv_dst = acc; mv_n = valid;
          acc += valid;
        }
      }
      if (!lost) {
        const uint64_t d = P + mv_dst;
        hw_st(slot, HW_MVSRC, mv_src); hw_st(slot, HW_MVDST_LO, (uint32_t)d); hw_st(slot, HW_MVDST_HI, (uint32_t)(d >> 32)); hw_st(slot, HW_MVN, mv_n);
        hw_st(slot, HW_MVOWN, mv_own);
        lds_release();
        hw_st(slot, HW_MVGO, seq);
        if (mv_n | mv_own) moving |= 1u << w;
      }
    }
    SPEC_PROF(2);
    if (own) {  // chunk 0 of a capped round: out of scratch slot 0
      constexpr uint32_t H = SPEC_SLOT_BYTES / 1024u;
      const uint32_t n16 = own >> 4;
      u32x4 t[H] = {};
      _Pragma("unroll") for (uint32_t h = 0; h < H; h++) {
        const uint32_t c = lane + 64u * h;
        if (c < n16) t[h] = *reinterpret_cast<gu32x4*>(spec + (uint64_t)c * 16);
      }
      uint32_t tail = 0;
      if (lane < (own & 15u)) tail = spec[(n16 << 4) + lane];
      gu8* dst = out + P;
      _Pragma("unroll") for (uint32_t h = 0; h < H; h++) {
        const uint32_t c = lane + 64u * h;
        if (c < n16) *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = t[h];
      }
      if (lane < (own & 15u)) dst[(n16 << 4) + lane] = (uint8_t)tail;
    }
    SPEC_PROF(3);
    if (lost) hc_st(HC_KIND, 3);  // helpers unusable from now on
    P += acc; i -= acc;
    br.seek_ahead(run_pos + bits_done);
    SPEC_PROF(4); SPEC_COUNT(5, 1); SPEC_COUNT(6, acc); SPEC_COUNT(7, bits_done);
  }
  // what the helpers are still moving into place is part of the output before anything after the run reads it
  for (uint32_t w = 1; w < nw; w++) {
    if (!((moving >> w) & 1u)) continue;
    uint32_t polls = 0;
    while (hw_ld(hb + w * HL_SLOT, HW_MVDONE) != mv_seq) {
      if (++polls > (1u << 22)) {  // (never seen; the launch must not hang and the output must not be silently wrong:
        hc_st(HC_KIND, 3); hc_st(HC_FAILED, 1);  // the caller reports the stream as failed)
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  lds_acquire();
  {  // the literals of the rounds come off the block length and the quota (the metablock length already has the run)
    const uint32_t got = LEAN_LD(L_LITS_LEFT) - i, bl0 = LEAN_LD(L_BL0), quota = LEAN_LD(L_QUOTA);
    if (lane == 0) { LEAN_ST(L_BL0, bl0 - got); LEAN_ST(L_QUOTA, quota - got); }
  }
  lds_st32(LDS_LEANWIN + 4u * lane, br.cur);
  if (lane == 0) {
    LEAN_ST(L_CHUNK_BASE, br.chunk_base);
    LEAN_ST(L_BUF_LO, (uint32_t)br.buf); LEAN_ST(L_BUF_HI, (uint32_t)(br.buf >> 32)); LEAN_ST(L_CNT, br.cnt); LEAN_ST(L_NEXT_DW, br.next_dw);
    LEAN_ST(L_ISSUED, br.issued_half); LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32));
    LEAN_ST(L_LITS_LEFT, i);
  }
  lds_sync();
}

#include "brotli_scan_engine.h"
#define PE_CFG_NS pe16
#define PE_CFG_WAVES 16
#define PE_CFG_RBL 32768
#define PE_CFG_PIPE 0
#define PE_CFG_DICT 0
#define PE_CFG_REMOTE 0
#include "brotli_path_engine.h"
#undef PE_CFG_NS
#undef PE_CFG_DICT
#define PE_CFG_NS pe16g
#define PE_CFG_DICT 1
#include "brotli_path_engine.h"
#undef PE_CFG_NS
#undef PE_CFG_DICT
#ifdef BROTLI_AMD_GANG_KERNEL
#undef PE_CFG_REMOTE
#define PE_CFG_NS pe16r
#define PE_CFG_DICT 0
#define PE_CFG_REMOTE 1
#include "brotli_path_engine.h"
#undef PE_CFG_NS
#undef PE_CFG_REMOTE
#define PE_CFG_REMOTE 0
#else
#define PE_CFG_DICT 0
#endif
#undef PE_CFG_WAVES
#undef PE_CFG_RBL
#undef PE_CFG_PIPE
#undef PE_CFG_DICT
#define PE_CFG_NS pe8
#define PE_CFG_WAVES 8
#define PE_CFG_RBL 16384
#define PE_CFG_PIPE 1
#define PE_CFG_DICT 0
#include "brotli_path_engine.h"
#undef PE_CFG_NS
#undef PE_CFG_WAVES
#undef PE_CFG_RBL
#undef PE_CFG_PIPE
#undef PE_CFG_DICT
#undef PE_CFG_REMOTE
using pe16::PE_MIN_INPUT;

// The pending copy of the lean loop lives in registers the compiler does not know about: v[120:123] (16 bytes per lane)
// and v124 (one byte per lane), named in inline asm only.  As C++ variables they were shuffled through other registers
// at the loop's back edge, and every such move waited for the load that had just been issued (or, before the next
// load, for the acknowledgement of the stores); loaded here, nothing waits until the bytes are stored.
#define PEND_REGS "v120", "v121", "v122", "v123", "v124"
__device__ __forceinline__ uint64_t lanes_below(uint32_t n) { return ~0ull >> (64u - n); }  // 1 <= n <= 64
__device__ __forceinline__ void pend_load16(gu8* src, uint32_t n16, uint32_t lane16) {
  asm volatile("s_mov_b64 exec, %0\n\tglobal_load_dwordx4 v[120:123], %1, %2\n\ts_mov_b64 exec, -1" :: "s"(lanes_below(n16)), "v"(lane16), "s"(src) : "memory", PEND_REGS);
}
__device__ __forceinline__ void pend_load8(gu8* src, uint32_t n, uint32_t off) {  // byte src[off] of lanes below n
  asm volatile("s_mov_b64 exec, %0\n\tglobal_load_ubyte v124, %1, %2\n\ts_mov_b64 exec, -1" :: "s"(lanes_below(n)), "v"(off), "s"(src) : "memory", PEND_REGS);
}
__device__ __forceinline__ void pend_store16(gu8* dst, uint32_t n16, uint32_t lane16) {
  asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dwordx4 %1, v[120:123], %2\n\ts_mov_b64 exec, -1" :: "s"(lanes_below(n16)), "v"(lane16), "s"(dst) : "memory");
}
__device__ __forceinline__ void pend_store8(gu8* dst, uint32_t n, uint32_t lane) {
  asm volatile("s_mov_b64 exec, %0\n\tglobal_store_byte %1, v124, %2\n\ts_mov_b64 exec, -1" :: "s"(lanes_below(n)), "v"(lane), "s"(dst) : "memory");
}
__device__ __forceinline__ uint32_t pend_byte(uint32_t k) {  // the byte lane k holds
  uint32_t r;
  asm volatile("s_waitcnt vmcnt(0)\n\tv_readlane_b32 %0, v124, %1" : "=s"(r) : "s"(k) : "memory");
  return r;
}

template <bool CTX_NEVER>
// Placement of the lean loop, in 4-byte steps from a 256-byte boundary, per instance (context-free / context-modelled):
// the loops of a lone wave are sensitive to where they lie relative to the 32-byte instruction-fetch lines (C3: 18.9 to
// 20.1 GB/s over the eight placements).  Measured on MI355X with tools/tune_lean_placement.sh; to be measured again after every edit of this function.
#ifndef BROTLI_AMD_LEAN_PAD_NEVER
#define BROTLI_AMD_LEAN_PAD_NEVER 3
#endif
#ifndef BROTLI_AMD_LEAN_PAD_CTX
#define BROTLI_AMD_LEAN_PAD_CTX 3
#endif
__device__ __noinline__ __attribute__((aligned(256))) uint32_t lean_commands(uint32_t lut_vgpr, uint32_t ctx_tree_v) {
  // (the function starts on a 256-byte boundary so that the placement of its loops relative to instruction-fetch
  // lines does not change with the code in front of it)
  asm volatile(".rept %0\n\ts_nop 0\n\t.endr" :: "n"(CTX_NEVER ? BROTLI_AMD_LEAN_PAD_NEVER : BROTLI_AMD_LEAN_PAD_CTX));
  const uint32_t lane = lane_id();
  const Arena a = {nullptr, 0xFFFFFFFFu, 0u};  // every table of the metablock is in LDS: the arena fields are not looked at
  BitReader br;
  br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
  br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED); br.end_dw = LEAN_LD(L_END_DW);
  br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);  // (the caller's window, as it is)
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)LEAN_LD(L_OUT_LO) | ((uint64_t)LEAN_LD(L_OUT_HI) << 32));
  uint64_t P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
  uint32_t quota = LEAN_LD(L_QUOTA);
  int32_t mlen = (int32_t)LEAN_LD(L_MLEN);
  uint32_t bl0 = LEAN_LD(L_BL0), bl1 = LEAN_LD(L_BL1), bl2 = LEAN_LD(L_BL2);
  int32_t d0 = (int32_t)LEAN_LD(L_D0), d1 = (int32_t)LEAN_LD(L_D1), d2 = (int32_t)LEAN_LD(L_D2), d3 = (int32_t)LEAN_LD(L_D3);
  uint32_t ncmd = 0;
  const uint32_t cmd_tree = LEAN_LD(L_CMD_TREE), lit_tree = LEAN_LD(L_LIT_TREE);
  const uint32_t dt0 = LEAN_LD(L_DT0), dt1 = LEAN_LD(L_DT1), dt2 = LEAN_LD(L_DT2), dt3 = LEAN_LD(L_DT3);
  const int32_t max_backward = (int32_t)LEAN_LD(L_MAX_BACKWARD);
  const uint32_t postfix_bits = LEAN_LD(L_POSTFIX), num_direct = LEAN_LD(L_NUM_DIRECT);
  // NPOSTFIX = NDIRECT = 0 (what encoders use by default): distance codes 16 .. 63 map to (extra bits, first distance)
  // through a table held one entry per lane instead of the arithmetic of decode.rs:2099-2128
  const bool dlut_ok = postfix_bits == 0u && num_direct == 16u;
  uint32_t dlut;
  {
    const uint32_t dv = (lane - 16u) & 63u, nb = (dv >> 1) + 1u;
    dlut = nb | ((((2u + (dv & 1u)) << nb) - 3u) << 5);  // distance = ((2 + (dv & 1)) << nb) - 4 + bits + 1
  }
  const uint32_t safe_dw = br.end_dw > 72u ? br.end_dw - 72u : 0u;
  const uint32_t lomask = lane < 32 ? 0xFFFFFFFFu : 0u;
  const uint32_t tree_addr = LDS_FIXED + lit_tree;
  // copy whose bytes are in registers but not stored yet (16 bytes per lane + a byte tail)
  uint32_t pendv_n16 = 0, pend_n = 0; uint64_t pend_pos = 0;  // (the bytes: see PEND_REGS)
  const uint32_t lane16 = lane << 4;
  // context-modelled metablocks: literals collected one per lane (stored 64 at a time or before the next copy), the
  // two bytes before P in p1/p2 when ctx_regs, else in the pending short copy (ctx_pend) or in memory
  uint32_t lit_reg = 0, lit_n = 0; uint64_t lit_pos = P;
  uint32_t p1 = CTX_NEVER ? 0u : LEAN_LD(L_P1), p2 = CTX_NEVER ? 0u : LEAN_LD(L_P2);
  bool ctx_regs = CTX_NEVER ? true : LEAN_LD(L_CTX_REGS) != 0u, ctx_pend = false;
  const uint32_t trivial = CTX_NEVER ? 1u : LEAN_LD(L_TRIVIAL), ctx_lut = CTX_NEVER ? 0u : LEAN_LD(L_CTX_LUT);
  const uint32_t lut0v = CTX_NEVER ? 0u : lds_ld32(ctx_lut + 4u * lane), lut1v = CTX_NEVER ? 0u : lds_ld32(ctx_lut + 256u + 4u * lane);
#define LEAN_FLUSH() do { if (pendv_n16 | pend_n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  /* the pending bytes have arrived */ \
                          if (pendv_n16) pend_store16(out + pend_pos, pendv_n16, lane16); \
                          if (pend_n) pend_store8(out + pend_pos + ((uint64_t)pendv_n16 << 4), pend_n, lane); \
                          if (!CTX_NEVER && lit_n) { if (lane < lit_n) out[lit_pos + lane] = (uint8_t)lit_reg; lit_n = 0; } \
                          pendv_n16 = 0; pend_n = 0; } while (0)
  uint32_t stage = LS_BEGIN;
  int32_t insert_len = 0, copy_len = 0, distance_code = 0;
  uint32_t distance_context = 0, lits_left = 0;

  // The root entry of the next command's prefix code is requested as soon as its bit position is known (before the
  // copy of the current command is issued) and picked up at the top of the loop: one LDS round trip off the chain.
  br.need32();
  uint32_t next_root = lds_ld16(LDS_FIXED + cmd_tree + (((uint32_t)br.buf & 0xFFu) << 1));

#ifdef BROTLI_AMD_PROFILE_LEAN
  uint64_t lp_acc[5] = {0, 0, 0, 0, 0}; uint64_t lp_t = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    LEAN_PROF(3);
    if (bl1 == 0 || br.next_dw >= safe_dw) { stage = LS_BEGIN; break; }  // block switch due, or close to the end of the input
    uint32_t cmd;
    {  // read_symbol() with the root lookup already under way (cnt >= 32 here)
      uint32_t e = rfl(next_root);
      uint32_t len = e & 15u;
      if (len > ROOT_BITS) {
        uint32_t idx = (e >> 4) + (((uint32_t)br.buf >> ROOT_BITS) & mask_bits(len - ROOT_BITS));
        e = rfl(lds_ld16(LDS_FIXED + cmd_tree + (idx << 1)));
        len = ROOT_BITS + (e & 15u);
      }
      br.drop(len);
      cmd = e >> 4;
    }
    uint32_t cell = cmd >> 6;  // RFC 7932 section 5 (see process_commands)
    uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u);
    uint32_t copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);
    uint32_t ie = rdlane(lut_vgpr, ins_code), ce = rdlane(lut_vgpr, 32u + copy_code);
    distance_code = cmd < 128 ? 0 : -1;
    distance_context = copy_code > 2 ? 3u : copy_code;
    insert_len = (int32_t)((ie & 0xFFFFu) + br.read24(ie >> 16));
    copy_len = (int32_t)((ce & 0xFFFFu) + br.read24(ce >> 16));
    bl1--;
    ncmd++;
    lits_left = (uint32_t)insert_len;
    LEAN_PROF(0);
    if (insert_len != 0) {
      if ((uint32_t)insert_len > quota || (uint32_t)insert_len > bl0) { stage = LS_AFTER_HEAD; break; }
      mlen -= insert_len;
      uint32_t i = (uint32_t)insert_len;
      if (!CTX_NEVER) {
        // ---- context-modelled: one at a time, the tree depends on the two bytes before ----
        if (!ctx_regs) {
          if (ctx_pend) {  // they are the tail of the short copy still in pend_reg (copy lengths start at 2)
            p1 = pend_byte(pend_n - 1u); p2 = pend_byte(pend_n - 2u);
          } else {
            LEAN_FLUSH();
            p1 = P >= 1 ? (uint32_t)rfl(out[P - 1]) : 0u;
            p2 = P >= 2 ? (uint32_t)rfl(out[P - 2]) : 0u;
          }
          ctx_regs = true;
        }
        if (lit_n == 0) lit_pos = P;
        while (i > 0 && br.next_dw < safe_dw) {
          uint32_t tree = lit_tree;
          if (!trivial) {
            // (the two 256-entry lookup tables of the block type's context mode, four entries per lane: two v_readlane
            // instead of an LDS round trip per literal)
            const uint32_t context = ((rdlane(lut0v, p1 >> 2) >> ((p1 & 3u) << 3)) | (rdlane(lut1v, p2 >> 2) >> ((p2 & 3u) << 3))) & 0xFFu;
            tree = rdlane(ctx_tree_v, context);
          }
          uint32_t lit = read_symbol<true>(br, a, tree);
          p2 = p1; p1 = lit;
          lit_reg = (lane == lit_n) ? lit : lit_reg;
          lit_n++;
          if (lit_n == 64) { if (lane < 64) out[lit_pos + lane] = (uint8_t)lit_reg; lit_pos += 64; lit_n = 0; }
          i--;
        }
      }
      if (CTX_NEVER && i >= SPEC_ROUND_MIN && br.next_dw + spec_input_dwords(SPEC_MAX_WAVES) < safe_dw && hc_ld(HC_KIND) != 3u) {
        // ---- long run: the caller runs rounds of one chunk per wave, all but the first decoded speculatively by the helper waves
        // (spec_rounds; not called from here: a call in this function costs the loop its SGPRs) ----
        stage = LS_LITERAL_ROUNDS; break;
      }
      const uint32_t run_rest = i;  // what the batches below still have to decode
      gu8* wp = out + P;
      if (CTX_NEVER && i > 2 && br.next_dw < safe_dw) {
        uint32_t woff = 0;
        do {
          br.ensure_dwords(3);
          uint32_t v0, v1, v2, v3, v4, t0, t1, off, n;
          asm volatile(LITERAL_BATCH_ASM
              : [buf] "+s"(br.buf), [cnt] "+s"(br.cnt), [ndw] "+s"(br.next_dw), [i] "+s"(i), [woff] "+s"(woff),
                [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [v4] "=&v"(v4),
                [t0] "=&s"(t0), [t1] "=&s"(t1), [off] "=&s"(off), [n] "=&s"(n)
              : [cur] "v"(br.cur), [lane] "v"(lane), [lomask] "v"(lomask), [tree] "s"(tree_addr), [cb] "s"(br.chunk_base), [wp] "s"(wp)
              : "memory", "vcc", "scc", "s90", "s91", "s92", "s93", "s94", "s95");
        } while (i > 2 && br.next_dw < safe_dw);
        wp += woff;
      }
      while (CTX_NEVER && i > 0 && i <= 2 && br.next_dw < safe_dw) {  // one or two literals: cheaper one by one
        uint32_t lit = read_symbol<true>(br, a, lit_tree);
        if (lane == 0) *wp = (uint8_t)lit;
        wp++; i--;
      }
      const uint32_t done = (CTX_NEVER ? run_rest : (uint32_t)insert_len) - i;
      P += done; bl0 -= done; quota -= done;
      lits_left = i;
      if (i != 0) { stage = LS_LITERALS_REST; break; }
      if (quota == 0) { stage = LS_LITERALS_AT_LIMIT; break; }
    }
    LEAN_PROF(1);
    // ---- distance (ReadDistanceInternal, decode.rs:2066-2131; see process_commands) ----
    if (distance_code >= 0) {
      distance_context = 1;
      distance_code = d0;
    } else {
      if (bl2 == 0) { stage = LS_DISTANCE; break; }
      uint32_t dtree = distance_context == 0 ? dt0 : distance_context == 1 ? dt1 : distance_context == 2 ? dt2 : dt3;
      uint32_t code = read_symbol<true>(br, a, dtree);
      distance_context = 0;
      if (code < 16) {
        if (code == 0) {
          distance_code = d0;
          distance_context = 1;
        } else {
          uint32_t sh = code << 1;
          uint32_t back = 3u - ((0xaaafff1bu >> sh) & 3u);
          int32_t v = back == 0 ? d0 : back == 1 ? d1 : back == 2 ? d2 : d3;
          int32_t mag = (int32_t)((0xfa5fa500u >> sh) & 3u);
          if (code & 1u) v += mag;
          else { v -= mag; if (v <= 0) v = 0x7fffffff; }
          distance_code = v;
        }
      } else if (dlut_ok && code < 64u) {
        const uint32_t de = rdlane(dlut, code);
        distance_code = (int32_t)((de >> 5) + br.read(de & 31u));
      } else {
        int32_t distval = (int32_t)code - (int32_t)num_direct;
        int32_t dc = (int32_t)code;
        if (distval >= 0) {
          int32_t postfix = distval & (int32_t)mask_bits(postfix_bits);
          distval >>= postfix_bits;
          uint32_t nbits = ((uint32_t)distval >> 1) + 1;
          uint32_t bits = br.read(nbits);
          int64_t offset = (int64_t)(int32_t)((((uint32_t)(distval & 1) + 2u) << nbits) - 4u);
          dc = (int32_t)(((offset + (int64_t)bits) << postfix_bits) + postfix + (int64_t)num_direct);
        }
        distance_code = (int32_t)((uint32_t)dc - 16u + 1u);
      }
      bl2--;
      if (br.next_dw > br.end_dw) { stage = LS_NEEDS_INPUT; break; }  // (cannot happen below safe_dw; literal runs are what moves far)
    }
    br.need32();
    next_root = lds_ld16(LDS_FIXED + cmd_tree + (((uint32_t)br.buf & 0xFFu) << 1));
    LEAN_PROF(2);
    // ---- copy: an LZ77 reference (not the dictionary) inside the quota that does not overlap itself ----
    {
      const uint32_t n = (uint32_t)copy_len, dist = (uint32_t)distance_code;
      const int32_t max_distance = (P < (uint64_t)(uint32_t)max_backward) ? (int32_t)P : max_backward;
      if (distance_code > max_distance || distance_code <= 0 || n > quota) { stage = LS_POST_DISTANCE; break; }
      // (one path through here for every kind of copy: two that each updated P and the quota and went back to the top
      // of the loop made the compiler shuffle a dozen scalars at the back edge)
      const bool tiny = !CTX_NEVER && n <= 64u;  // short copy where literal context matters: one byte per lane
      if (!tiny && dist < n) { stage = LS_POST_DISTANCE; break; }
      if (distance_context == 0) { d3 = d2; d2 = d1; d1 = d0; d0 = distance_code; }
      mlen -= copy_len;
#ifdef BROTLI_AMD_PROFILE_LEAN
      { uint64_t t0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); uint64_t t1_ = __builtin_amdgcn_s_memtime();
        LEAN_FLUSH();
        uint64_t t2_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); uint64_t t3_ = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0) { g_lean_prof[5] += t1_ - t0_; g_lean_prof[6] += t3_ - t2_; } }
#else
      LEAN_FLUSH();
#endif
      gu8* src = out + P - dist;
      uint32_t n16 = 0, rem = n;
      if (tiny) {
        // (pattern when it overlaps itself) so that the next literal can take the two bytes before it straight from
        // the register
        pend_load8(src, n, dist >= n ? lane : lane % dist);
        ctx_regs = false; ctx_pend = true;
      } else {
        if (!CTX_NEVER) { ctx_regs = false; ctx_pend = false; }
        n16 = n >> 4; rem = n & 15u;
        if (n > 1024u) {  // long: all but the last (partial) KiB right away, 16 bytes per lane and step
          gu8* dst = out + P;
          const uint32_t whole = (n16 - 1u) & ~63u;  // 16-byte pieces in whole steps, at least one piece is left for below
          for (uint32_t c = lane; c < whole; c += 64) {
            u32x4 t = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16);
            *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = t;
          }
          src += (uint64_t)whole * 16; P += (uint64_t)whole * 16; quota -= whole * 16; n16 -= whole;
        }
        // the load is issued now, the store when the next command gets here (its source may be what this one writes)
        if (n16) pend_load16(src, n16, lane16);
        if (rem) pend_load8(src, rem, (n16 << 4) + lane);
      }
      pendv_n16 = n16; pend_n = rem; pend_pos = P;
      P += (n16 << 4) + rem;
      quota -= (n16 << 4) + rem;
      if (quota == 0) { stage = LS_COMMAND_DONE; break; }
    }
  }
  if (!CTX_NEVER && !ctx_regs && ctx_pend) { p1 = pend_byte(pend_n - 1u); p2 = pend_byte(pend_n - 2u); ctx_regs = true; }
  LEAN_FLUSH();
#ifdef BROTLI_AMD_PROFILE_LEAN
  if (blockIdx.x == 0 && lane == 0) { for (int k = 0; k < 4; k++) g_lean_prof[k] += lp_acc[k]; g_lean_prof[4] += ncmd; }
#endif
#undef LEAN_FLUSH
  lds_st32(LDS_LEANWIN + 4u * lane, br.cur);
  if (lane == 0) {
    LEAN_ST(L_CHUNK_BASE, br.chunk_base);
    if (!CTX_NEVER) { LEAN_ST(L_P1, p1); LEAN_ST(L_P2, p2); LEAN_ST(L_CTX_REGS, ctx_regs ? 1u : 0u); }
    LEAN_ST(L_BUF_LO, (uint32_t)br.buf); LEAN_ST(L_BUF_HI, (uint32_t)(br.buf >> 32)); LEAN_ST(L_CNT, br.cnt); LEAN_ST(L_NEXT_DW, br.next_dw);
    LEAN_ST(L_ISSUED, br.issued_half); LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32)); LEAN_ST(L_QUOTA, quota);
    LEAN_ST(L_MLEN, mlen); LEAN_ST(L_BL0, bl0); LEAN_ST(L_BL1, bl1); LEAN_ST(L_BL2, bl2);
    LEAN_ST(L_D0, d0); LEAN_ST(L_D1, d1); LEAN_ST(L_D2, d2); LEAN_ST(L_D3, d3); LEAN_ST(L_NCMD_LO, ncmd);
    LEAN_ST(L_INSERT, insert_len); LEAN_ST(L_COPY, copy_len); LEAN_ST(L_DCODE, distance_code); LEAN_ST(L_DCTX, distance_context);
    LEAN_ST(L_LITS_LEFT, lits_left);
  }
  lds_sync();
  return rfl(stage);
}


// Wave 2 of the block while the decoding wave parses a context-modelled metablock: command records ahead of it (see above).
__device__ __noinline__ void rec_wave() {
  const uint32_t lane = lane_id();
  const uint32_t xb = hc_ld(HC_EXT_BASE), ring = xb + SPX_CTL_BYTES;
  gcu32* const in = BitReader::base();
  uint32_t lut = 0;  // per-lane image of the insert / copy code tables (as the decoding wave's)
  if (lane < 24) lut = (uint32_t)kInsBase[lane] | ((uint32_t)kInsExtra[lane] << 16);
  else if (lane >= 32 && lane < 56) lut = (uint32_t)kCopyBase[lane - 32] | ((uint32_t)kCopyExtra[lane - 32] << 16);
  const uint32_t origin_dw = sp_ld(xb, XW_ORIGIN_DW), limit = sp_ld(xb, XW_LIMIT), ring_mask = sp_ld(xb, XW_RING_MASK);
  uint32_t epoch = 0, F = 0, cmd_tree = 0, dt0 = 0, dt1 = 0, dt2 = 0, dt3 = 0, postfix_bits = 0, num_direct = 0, idle = 0;
  for (;;) {
    if (sp_ld(xb, XW_STOP) != 0u) break;
    const uint32_t ep = sp_ld(xb, XW_EPOCH);
    const uint32_t pos = sp_ld(xb, XW_POS);
    if (ep != epoch) {  // new prefix codes (a block switch): everything from the parser's position on again
      lds_acquire();
      epoch = ep;
      cmd_tree = sp_ld(xb, XW_CMD_TREE); dt0 = sp_ld(xb, XW_DT0); dt1 = sp_ld(xb, XW_DT0 + 1); dt2 = sp_ld(xb, XW_DT0 + 2); dt3 = sp_ld(xb, XW_DT0 + 3);
      postfix_bits = sp_ld(xb, XW_POSTFIX); num_direct = sp_ld(xb, XW_NUM_DIRECT);
      F = sp_ld(xb, XW_POS) & ~63u;
    }
    if (epoch == 0u) { __builtin_amdgcn_s_sleep(8); continue; }
    if (F < (pos & ~63u)) F = pos & ~63u;  // the parser went ahead (literals, commands by hand)
    if (F + 64u > limit || F + 64u > pos + (ring_mask - 63u)) {  // the end of what may be read, or a lap ahead of the parser
      idle++;  // (a lap is some forty commands: the parser is a long way off; polling costs the parser on this SIMD its issue slots)
      if (idle < 4u) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(64);
      continue;
    }
    idle = 0;
    // the 64 stream bits from this lane's position
    const uint32_t p = F + lane, sh = p & 31u;
    gcu32* const q = in + origin_dw + (p >> 5);
    const uint32_t a0 = q[0], a1 = q[1], a2 = q[2];
    const uint32_t lo = __builtin_amdgcn_alignbit(a1, a0, sh), hi = __builtin_amdgcn_alignbit(a2, a1, sh);
    const ScHead h = sc_head(lo, hi, cmd_tree, lut);
    uint32_t w0 = (h.copy & 0xFFFFu) | (h.dctx << XR_DCTX_SHIFT), w1 = h.insert, bits = h.bits;
    bool valid = h.copy < 65536u;
    if (h.insert != 0u) w0 |= XR_LITERALS | (h.implicit ? XR_IMPLICIT : 0u);
    else if (h.implicit) w0 |= XR_IMPLICIT;
    else {
      const uint64_t w = (((uint64_t)hi << 32) | lo) >> h.bits;
      const uint32_t dtree = h.dctx == 0u ? dt0 : h.dctx == 1u ? dt1 : h.dctx == 2u ? dt2 : dt3;
      const ScDist d = sc_dist((uint32_t)w, (uint32_t)(w >> 32), dtree, postfix_bits, num_direct);
      bits += d.bits;
      w1 = d.val;
      if (d.kind == SCK_SHORT) w0 |= XR_SHORT;
    }
    valid = valid && bits <= 63u;  // (all of the command inside the lane's 64 bits; 63: the parser shifts its 64-bit buffer by that much)
    w0 |= (bits & 127u) << 16;
    if (valid) w0 |= XR_VALID;
    if (!valid || h.copy > 63u) w0 |= XR_NOT_RUN;
    *reinterpret_cast<__attribute__((address_space(3))) uint64_t*>(&g_smem[ring + ((p & ring_mask) << 3)]) = ((uint64_t)w1 << 32) | w0;
    F += 64u;
    lds_release();
    if (lane == 0) *reinterpret_cast<volatile __attribute__((address_space(3))) uint64_t*>(&g_smem[xb + 4u * XW_FRONT]) = ((uint64_t)epoch << 32) | F;
  }
  lds_release();
  sp_st(xb, XW_STOP, 2u);
}

// ===================================== record lean loop =====================================
// The plain commands of a context-modelled metablock out of their records, in hand-written scalar code: LEAN_REC_RUN_ASM, generated by
// tools/gen_rec_asm.py into brotli_rec_run_asm.h (that file says what the loop does and why it is laid out as it is: a lone wave pays 13
// clocks for a branch it does not take and 26 - 29 for one it takes, tools/ubench/branch.hip).  One iteration = one command of the plain kind:
//   * WITHOUT literals: a short LZ77 copy that does not repeat itself, clear of every limit, all of it out of the record at the reader's
//     position -- distance (explicit, implicit, or a ring code: decode.rs:2017-2049), counts, the reader moved on by the record's bits;
//   * WITH literals (round 6; decode.rs:2463-2551): at most sixteen, the tree of each by the two bytes before it (the context tables and
//     the map in registers: three v_readlane), then the distance code behind them (decode.rs:2066-2131: ring codes and the codes of the
//     plain alphabet with their extra bits out of a lane table) and the same plain copy.  Nothing of such a command is COMMITTED before
//     all of it is known to be plain: where anything else turns up -- a word of the static dictionary, a copy that repeats itself, a
//     distance code beyond the lane table, a count that has run out -- the reader's position has not moved, the context bytes are put
//     back, and the compiled loop below takes that command from its first bit.
// The copy in flight (its bytes on their way into v124) is stored when the next command comes by, a command's literals with it in one
// store: lanes below pn the copy's bytes, the literals behind them.  The record of the next command is asked for before the memory
// pipe is waited for.  Leaves in front of the first command that is anything else, with nothing of it touched; bit 0 of `ok` then says
// whether rx / ry hold that command's record, bit 1 whether p1 / p2 are the two bytes before P.  P < 2^32 in here; copies of at most 63 bytes.
#ifdef BROTLI_AMD_PROFILE_RUN_WAIT   // (tools/gen_rec_asm.py --profile-wait: s_memtime around the run's waits for the memory pipe; with -DBROTLI_AMD_PROFILE_SPLIT)
#include "brotli_rec_run_asm_wait.h"
#define LRA_WAIT_OPERAND , [wacc] "+s"(run_wait)
#define LRA_WAIT_CLOBBERS "s74", "s75", "s76", "s77", "s78", "s79",
#else
#include "brotli_rec_run_asm.h"
#define LRA_WAIT_OPERAND
#define LRA_WAIT_CLOBBERS
#endif
static_assert(XR_VALID == 0x800000u && XR_LITERALS == 0x1000000u && XR_IMPLICIT == (1u << 25) && XR_DCTX_SHIFT == 26 && XR_SHORT == (1u << 28) && XR_NOT_RUN == (1u << 31), "LEAN_REC_RUN_ASM spells these out");

// The lean loop of a context-modelled metablock whose command records are there (rec_wave): nothing of a command's head is
// parsed here, and a command without literals is not parsed at all -- copy length, distance and the bits to skip come out of
// the record at the reader's position.  Kept small on purpose (the compiler's scalar code for a loop with many ways out is
// mostly bookkeeping of which way it went): one plain path -- literals of at most 64, then a short LZ77 copy that does not
// repeat itself, clear of every limit and of the end of the register window -- and everything else is left, at the command
// boundary (LS_BEGIN) or in front of the distance (LS_DISTANCE / LS_POST_DISTANCE), to the checked stages of process_commands.
// State and hand-over as lean_commands<false> (decode.rs:2359-2726 for the path it takes).
__device__ __noinline__ __attribute__((aligned(256))) uint32_t lean_rec_commands(uint32_t ctx_tree_v) {
  const uint32_t lane = lane_id();
  BitReader br;
  br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
  br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED); br.end_dw = LEAN_LD(L_END_DW);
  br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)LEAN_LD(L_OUT_LO) | ((uint64_t)LEAN_LD(L_OUT_HI) << 32));
  uint64_t P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
  uint32_t quota = LEAN_LD(L_QUOTA);
  int32_t mlen = (int32_t)LEAN_LD(L_MLEN);
  uint32_t bl0 = LEAN_LD(L_BL0), bl1 = LEAN_LD(L_BL1), bl2 = LEAN_LD(L_BL2);
  int32_t d0 = (int32_t)LEAN_LD(L_D0), d1 = (int32_t)LEAN_LD(L_D1), d2 = (int32_t)LEAN_LD(L_D2), d3 = (int32_t)LEAN_LD(L_D3);
  uint32_t ncmd = 0;
  const uint32_t cmd_tree = LEAN_LD(L_CMD_TREE), lit_tree = LEAN_LD(L_LIT_TREE);
  const uint32_t dt0 = LEAN_LD(L_DT0), dt1 = LEAN_LD(L_DT1), dt2 = LEAN_LD(L_DT2), dt3 = LEAN_LD(L_DT3);
  const uint32_t max_backward = LEAN_LD(L_MAX_BACKWARD);
  const uint32_t postfix_bits = LEAN_LD(L_POSTFIX), num_direct = LEAN_LD(L_NUM_DIRECT);
  const bool dlut_ok = postfix_bits == 0u && num_direct == 16u;
  uint32_t dlut;
  {
    const uint32_t dv = (lane - 16u) & 63u, nb = (dv >> 1) + 1u;
    dlut = dlut_ok ? nb | ((((2u + (dv & 1u)) << nb) - 3u) << 5) : 0u;  // distance = ((2 + (dv & 1)) << nb) - 4 + bits + 1   (all zero where the alphabet is another: the run looks at the entry)
  }
  // (a command is begun at most fifty dwords into the register window, and no command of the plain path reads more than thirteen:
  // the window only moves between two commands)
  const uint32_t safe_dw = br.end_dw > 72u ? br.end_dw - 72u : 0u;
  uint32_t win_end = br.chunk_base + 50u;
  // the reader inside the window: a command of the plain path takes at most 2 (head) + 8 (sixteen literals) + 2 (distance) + 1 dwords
  auto pull = [&]() { const uint32_t dw = rdlane(br.cur, br.next_dw - br.chunk_base); br.buf |= (uint64_t)dw << br.cnt; br.cnt += 32u; br.next_dw++; };
  auto need32 = [&]() { if (br.cnt < 32u) pull(); };
  auto advance = [&](uint32_t n) {  // n <= 64
    if (n > br.cnt) { n -= br.cnt; br.buf = 0; br.cnt = 0; if (n >= 32u) { br.next_dw++; n -= 32u; } pull(); }
    br.buf >>= n; br.cnt -= n;
  };
  auto read24 = [&](uint32_t n) -> uint32_t { need32(); const uint32_t v = (uint32_t)br.buf & ((1u << n) - 1u); br.buf >>= n; br.cnt -= n; return v; };
  auto symbol = [&](uint32_t tree) -> uint32_t {  // read_symbol<true>
    need32();
    const uint32_t bits = (uint32_t)br.buf;
    uint32_t e = rfl(lds_ld16(LDS_FIXED + tree + ((bits & 0xFFu) << 1)));
    uint32_t len = e & 15u;
    if (len > ROOT_BITS) {
      const uint32_t idx = (e >> 4) + ((bits >> ROOT_BITS) & mask_bits(len - ROOT_BITS));
      e = rfl(lds_ld16(LDS_FIXED + tree + (idx << 1)));
      len = ROOT_BITS + (e & 15u);
    }
    br.buf >>= len; br.cnt -= len;
    return e >> 4;
  };
  uint32_t p1 = LEAN_LD(L_P1), p2 = LEAN_LD(L_P2);
  bool ctx_regs = LEAN_LD(L_CTX_REGS) != 0u;  // else: the tail of the short copy in flight (pend_n != 0), or memory
  const uint32_t trivial = LEAN_LD(L_TRIVIAL), ctx_lut = LEAN_LD(L_CTX_LUT);
  const uint32_t lut0v = lds_ld32(ctx_lut + 4u * lane), lut1v = lds_ld32(ctx_lut + 256u + 4u * lane);
  uint32_t pend_n = 0; uint64_t pend_pos = 0;  // short copy whose bytes are on their way into v124 (see PEND_REGS)
  uint32_t stage = LS_BEGIN;
  int32_t insert_len = 0, copy_len = 0, distance_code = 0;
  uint32_t distance_context = 0;

  const uint32_t xb = hc_ld(HC_EXT_BASE), xring = xb + SPX_CTL_BYTES;
  const uint32_t origin_bits = LEAN_LD(L_SP_ORIGIN) << 5, ring_mask = sp_ld(xb, XW_RING_MASK);
  uint32_t front_c = 0;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 rec_v = {0u, 0u};
  uint32_t my_epoch = sp_ld(xb, XW_EPOCH);
  bool new_epoch = false;   // (wave 2 starts over at the reader's position: its first records are a thousand clocks away, not there yet)
  {
    // the prefix codes the records are parsed with: new ones (the first time, after a block switch) start a new epoch
    const bool same = my_epoch != 0u && sp_ld(xb, XW_CMD_TREE) == LDS_FIXED + cmd_tree && sp_ld(xb, XW_DT0) == LDS_FIXED + dt0 && sp_ld(xb, XW_DT0 + 1) == LDS_FIXED + dt1 &&
                      sp_ld(xb, XW_DT0 + 2) == LDS_FIXED + dt2 && sp_ld(xb, XW_DT0 + 3) == LDS_FIXED + dt3;
    sp_st(xb, XW_POS, br.next_dw * 32u - br.cnt - origin_bits);
    if (!same) {
      sp_st(xb, XW_CMD_TREE, LDS_FIXED + cmd_tree); sp_st(xb, XW_DT0, LDS_FIXED + dt0); sp_st(xb, XW_DT0 + 1, LDS_FIXED + dt1);
      sp_st(xb, XW_DT0 + 2, LDS_FIXED + dt2); sp_st(xb, XW_DT0 + 3, LDS_FIXED + dt3);
      sp_st(xb, XW_POSTFIX, postfix_bits); sp_st(xb, XW_NUM_DIRECT, num_direct);
      my_epoch++;
      lds_release();
      sp_st(xb, XW_EPOCH, my_epoch);
      new_epoch = true;
    }
  }
  // the record at the reader's position, once wave 2 has written it (rel: stream bits from the ring's origin)
  uint32_t pos_said = br.next_dw * 32u - br.cnt - origin_bits;
  auto request = [&]() -> bool {
    const uint32_t rel = br.next_dw * 32u - br.cnt - origin_bits;
    if (rel - pos_said >= 128u) { sp_st(xb, XW_POS, rel); pos_said = rel; }  // (wave 2 stays less than a lap ahead of what it was told)
    if (rel >= 0x40000000u) return false;
    if (rel >= front_c) {
      const uint64_t f = *reinterpret_cast<volatile __attribute__((address_space(3))) uint64_t*>(&g_smem[xb + 4u * XW_FRONT]);
      front_c = rfl((uint32_t)(f >> 32)) == my_epoch ? rfl((uint32_t)f) : 0u;
      if (rel >= front_c) return false;
    }
    rec_v = *reinterpret_cast<__attribute__((address_space(3))) const u32x2*>(&g_smem[xring + ((rel & ring_mask) << 3)]);
    return true;
  };
  // the copy in flight goes to memory (its bytes have arrived)
  auto flush = [&]() {
    if (pend_n) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pend_store8(out + pend_pos, pend_n, lane); pend_n = 0; }
  };
  bool rec_ok = request();
  // (behind a block switch the records are parsed anew: waiting for the first of them here -- a thousand clocks -- instead of leaving, which sends a command through the checked
  // loop at ten times that: twenty block switches a stream of alice29)
  if (new_epoch) for (uint32_t spins = 0; !rec_ok && spins < 96u; spins++) { __builtin_amdgcn_s_sleep(4); rec_ok = request(); }
  bool waited_out = false;   // (the loop is left because the records are not there: not a verdict on the stream's commands, see process_commands)
  bool long_cmd = false;     // (... because of a command longer than the run takes: that is one)
  static const bool no_run_asm = false;
  const uint32_t ctx_tree_abs = ctx_tree_v + LDS_FIXED;   // (the run's literals: a context's tree as an address)
#ifdef BROTLI_AMD_PROFILE_RUN_WAIT
  uint64_t run_wait = 0;
#endif
  // what the run reads once a command at most, a lane each: the four distance contexts' tables (lanes 0 .. 3), XW_POS (lane 5), the static dictionary's address (lanes 6, 7), XW_FRONT and the codes' epoch (lanes 8, 9)
  const uint32_t run_params = lane == 0u ? LDS_FIXED + dt0 : lane == 1u ? LDS_FIXED + dt1 : lane == 2u ? LDS_FIXED + dt2 : lane == 3u ? LDS_FIXED + dt3 :
                              lane == 6u ? sp_ld(xb, XW_DICT_LO) : lane == 7u ? sp_ld(xb, XW_DICT_HI) : lane == 8u ? xb + 4u * (uint32_t)XW_FRONT : lane == 9u ? my_epoch :
                              xb + 4u * (uint32_t)XW_POS;
  // ... and of a word of the static dictionary as it stands (transform 0), by its length: where the words of that length begin | the bits of their index << 24
  const uint32_t run_wtab = lane >= 4u && lane <= 24u ? kDictOffsetsByLength[lane] | ((uint32_t)kDictSizeBitsByLength[lane] << 24) : 0u;
  // ... and of a copy that repeats itself, by its distance d: 65536 / d + 1 (lane * that >> 16 = lane / d: the copy's byte i is byte i mod d of its source)
  const uint32_t run_mtab = lane != 0u ? 65536u / lane + 1u : 0u;

#ifdef BROTLI_AMD_PROFILE_SPLIT
  uint64_t lap_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t lap_t = __builtin_amdgcn_s_memtime(); const uint64_t lap_t0 = lap_t; uint32_t n_run = 0, n_lit = 0, n_nolit = 0, n_word = 0;
#endif
  for (;;) {
    if (bl1 == 0 || br.next_dw >= safe_dw) break;
    // (the only place the window moves: between two commands.  The run goes by the reader's position in window bits: the window starts
    // in front of whatever the bit buffer still holds)
    if (br.next_dw >= win_end || (br.next_dw - br.chunk_base) * 32u < br.cnt) { br.rebase_back2(); win_end = br.chunk_base + 50u; }
    SPLIT_LAP(0);
    if (rec_ok && !no_run_asm && P + (uint64_t)quota <= 0xFFFFFFFFull && front_c <= 0x40000000u) {   // (the run keeps the output position in 32 bits and moves it by at most `quota`: ADVICE round 3)
      // ---- a run of commands without literals (see LEAN_REC_RUN_ASM) ----
      uint32_t ok = rfl(1u | (ctx_regs ? 2u : 0u)), rx = rec_v.x, ry = rec_v.y, P32 = rfl((uint32_t)P);   // (the copy in flight lies at P - pend_n)
      uint32_t p1s = rfl(p1), p2s = rfl(p2), saids = rfl(pos_said), bl0s = rfl(bl0), fronts = rfl(front_c);
      const uint32_t lim = rfl(safe_dw < win_end ? safe_dw : win_end);  // (rfl: scalar registers for the "s" operands)
      const uint32_t ncmd0 = ncmd, bl1_0 = bl1, quota0 = quota; (void)ncmd0;
      asm volatile(LEAN_REC_RUN_ASM
          : [buf] "+s"(br.buf), [cnt] "+s"(br.cnt), [ndw] "+s"(br.next_dw), [bl0] "+s"(bl0s), [bl1] "+s"(bl1), [bl2] "+s"(bl2),
            [d0] "+s"(d0), [d1] "+s"(d1), [d2] "+s"(d2), [d3] "+s"(d3), [P] "+s"(P32), [quota] "+s"(quota), [pn] "+s"(pend_n),
            [ok] "+s"(ok), [p1] "+s"(p1s), [p2] "+s"(p2s), [said] "+s"(saids), [front] "+s"(fronts), [rx] "+v"(rx), [ry] "+v"(ry) LRA_WAIT_OPERAND
          : [cur] "v"(br.cur), [lane] "v"(lane), [lut0] "v"(lut0v), [lut1] "v"(lut1v), [ctxtree] "v"(ctx_tree_abs), [dlut] "v"(dlut), [params] "v"(run_params), [wtab] "v"(run_wtab), [mtab] "v"(run_mtab),
            [cb] "s"(br.chunk_base), [outlo] "s"((uint32_t)(uintptr_t)out), [outhi] "s"((uint32_t)((uint64_t)(uintptr_t)out >> 32)),
            [maxb] "s"(max_backward), [lim] "s"(lim), [xring] "s"(xring), [rmask] "s"(ring_mask), [org] "s"(origin_bits),
            [littree] "s"(rfl(LDS_FIXED + lit_tree)), [trivial] "s"(rfl(trivial))
          : "memory", "vcc", "scc", "m0", LRA_WAIT_CLOBBERS "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
            "v115", "v116", "v117", "v118", "v119", "v124");
      ncmd += bl1_0 - bl1; mlen -= (int32_t)(quota0 - quota);   // (a command of the run takes one of the command block's count, and from the quota what it takes from the metablock)
      SPLIT_LAP(1);
#ifdef BROTLI_AMD_PROFILE_SPLIT
      n_run += ncmd - ncmd0;
#endif
      P = P32; pend_pos = (uint64_t)rfl(P32 - pend_n);
      p1 = p1s; p2 = p2s; pos_said = saids; bl0 = bl0s; front_c = fronts;
      rec_v.x = rx; rec_v.y = ry; rec_ok = (ok & 1u) != 0u;
      ctx_regs = (ok & 2u) != 0u;  // (after a command of the run: the two bytes before P are the tail of the copy in flight)
      if (bl1 == 0 || br.next_dw >= safe_dw) break;
      if (br.next_dw >= win_end) continue;
    }
    for (uint32_t spins = 0; !rec_ok && spins < 8u; spins++) { if (spins) __builtin_amdgcn_s_sleep(4); rec_ok = request(); }  // (wave 2 is rarely behind)
    if (!rec_ok) { waited_out = true; break; }
    const uint32_t rec_lo = rfl(rec_v.x), rec_hi = rfl(rec_v.y);
    if (!(rec_lo & XR_VALID)) break;
    const uint32_t n = rec_lo & 0xFFFFu, nb = (rec_lo >> 16) & 127u;
    uint32_t lit_n = 0, lit_reg = 0;
    int32_t dist = 0;
    uint32_t push = 0;
    bool committed = false;  // head and literals taken (a command with literals), or nothing yet
    if (rec_lo & XR_LITERALS) {
      // ---- literals: the tree depends on the two bytes before (decode.rs:2463-2551) ----
      const uint32_t ins = rec_hi;
      if (ins > 63u) long_cmd = true;   // (more literals than the run takes: what a stream of long literal runs does at every command)
      if (ins > 16u || ins >= quota || ins > bl0) break;
      advance(nb);
      if (!ctx_regs) {
        if (pend_n >= 2u) { p1 = pend_byte(pend_n - 1u); p2 = pend_byte(pend_n - 2u); }  // (the tail of the short copy in flight)
        else { flush(); p1 = P >= 1 ? (uint32_t)rfl(out[P - 1]) : 0u; p2 = P >= 2 ? (uint32_t)rfl(out[P - 2]) : 0u; }
        ctx_regs = true;
      }
      SPLIT_LAP(2);
      for (uint32_t i = 0; i < ins; i++) {
        uint32_t tree = lit_tree;
        if (!trivial) {
          const uint32_t context = ((rdlane(lut0v, p1 >> 2) >> ((p1 & 3u) << 3)) | (rdlane(lut1v, p2 >> 2) >> ((p2 & 3u) << 3))) & 0xFFu;
          tree = rdlane(ctx_tree_v, context);
        }
        const uint32_t lit = symbol(tree);
        p2 = p1; p1 = lit;
        lit_reg = (lane == i) ? lit : lit_reg;
      }
      SPLIT_LAP(3);
#ifdef BROTLI_AMD_PROFILE_SPLIT
      n_lit++;
#endif
      lit_n = ins;
      insert_len = (int32_t)ins; copy_len = (int32_t)n;
      // ---- the distance behind them (ReadDistanceInternal, decode.rs:2066-2131) ----
      if (rec_lo & XR_IMPLICIT) { dist = d0; push = 0u; distance_context = 1; distance_code = d0; }
      else {
        distance_context = (rec_lo >> XR_DCTX_SHIFT) & 3u;
        distance_code = -1;
        if (bl2 == 0) stage = LS_DISTANCE;
        else {
          const uint32_t dtree = distance_context == 0 ? dt0 : distance_context == 1 ? dt1 : distance_context == 2 ? dt2 : dt3;
          const uint32_t code = symbol(dtree);
          distance_context = 0;
          bl2--;
          if (code == 0u) { dist = d0; push = 0u; distance_context = 1; }
          else if (code < 16u) {
            const uint32_t sh = code << 1;
            const uint32_t back = 3u - ((0xaaafff1bu >> sh) & 3u);
            int32_t v = back == 0 ? d0 : back == 1 ? d1 : back == 2 ? d2 : d3;
            const int32_t mag = (int32_t)((0xfa5fa500u >> sh) & 3u);
            if (code & 1u) v += mag;
            else { v -= mag; if (v <= 0) v = 0x7fffffff; }
            dist = v; push = 1u;
          } else if (dlut_ok && code < 64u) {
            const uint32_t de = rdlane(dlut, code);
            dist = (int32_t)((de >> 5) + read24(de & 31u)); push = 1u;
          } else {
            int32_t distval = (int32_t)code - (int32_t)num_direct;
            int32_t dc = (int32_t)code;
            if (distval >= 0) {
              const int32_t postfix = distval & (int32_t)mask_bits(postfix_bits);
              distval >>= postfix_bits;
              const uint32_t nbits = ((uint32_t)distval >> 1) + 1;
              const uint32_t bits = read24(nbits);
              const int64_t offset = (int64_t)(int32_t)((((uint32_t)(distval & 1) + 2u) << nbits) - 4u);
              dc = (int32_t)(((offset + (int64_t)bits) << postfix_bits) + postfix + (int64_t)num_direct);
            }
            dist = (int32_t)((uint32_t)dc - 16u + 1u); push = 1u;
          }
          distance_code = dist;
        }
      }
      SPLIT_LAP(4);
      bl1--; ncmd++;
      mlen -= (int32_t)ins;
      // the literals go out with (in front of) the copy in flight: one store each
      flush();
      if (lane < lit_n) out[P + lane] = (uint8_t)lit_reg;
      P += ins; bl0 -= ins; quota -= ins;
      if (stage != LS_BEGIN) break;
      committed = true;
    } else {
      // ---- a command without literals: all of it is in the record ----
      if (rec_lo & XR_IMPLICIT) { dist = d0; push = 0u; }
      else {
        if (bl2 == 0) break;
        push = 1u;
        if (!(rec_lo & XR_SHORT)) dist = (int32_t)rec_hi;
        else if (rec_hi == 0u) { dist = d0; push = 0u; }
        else {  // TakeDistanceFromRingBuffer, decode.rs:2017-2049
          const uint32_t sh = rec_hi << 1;
          const uint32_t back = 3u - ((0xaaafff1bu >> sh) & 3u);
          int32_t v = back == 0 ? d0 : back == 1 ? d1 : back == 2 ? d2 : d3;
          const int32_t mag = (int32_t)((0xfa5fa500u >> sh) & 3u);
          if (rec_hi & 1u) v += mag;
          else { v -= mag; if (v <= 0) v = 0x7fffffff; }
          dist = v;
        }
      }
    }
    // ---- the plain copy (a short LZ77 reference that does not repeat itself, clear of every limit), or a word of the static dictionary
    // that is (decode.rs:2593-2640); anything else goes to the checked stages -- whole, if nothing of it has been taken yet, else they
    // finish it from the distance on (the ring and the copy's counts are untouched) ----
    SPLIT_LAP(5);
    const uint32_t max_distance = P < (uint64_t)max_backward ? (uint32_t)P : max_backward;
    const bool plain = dist > 0 && (uint32_t)dist <= max_distance && n <= 63u && (uint32_t)dist >= n && n < quota;   // (63: the hand-written run makes a copy's lanes with s_bfm_b64)
    bool word = false;
    WordShape w = {};
    uint32_t word_offset = 0;
    if (!plain) {
      if ((uint32_t)dist > max_distance && dist > 0 && dist <= 0x7FFFFFFC && n >= 4u && n <= 24u) {
        const uint32_t shift = kDictSizeBitsByLength[n];
        const uint32_t word_id = (uint32_t)dist - max_distance - 1u;
        const uint32_t transform_idx = word_id >> shift;
        if (transform_idx < (uint32_t)BROTLI_NUM_TRANSFORMS) {
          word_offset = kDictOffsetsByLength[n] + (word_id & mask_bits(shift)) * n;
          w = word_shape(n, transform_idx);
          word = w.total != 0u && w.total < quota && (int32_t)w.total <= mlen;
        }
      }
      if (!word) { if (n > 63u) long_cmd = true; if (committed) stage = LS_POST_DISTANCE; break; }
    }
    if (!committed) {
      if (!(rec_lo & XR_IMPLICIT)) bl2--;
      bl1--; ncmd++;
      advance(nb);
      flush();
    }
    if (word) {  // (the ring is not touched: decode.rs:2643-2644)
      gcu8* const dict = (gcu8*)(uintptr_t)((uint64_t)sp_ld(xb, XW_DICT_LO) | ((uint64_t)sp_ld(xb, XW_DICT_HI) << 32));
      const uint32_t ob = dictionary_word_bytes(dict, word_offset, w);
      if (lane < w.total) out[P + lane] = (uint8_t)ob;
      if (w.total >= 2u) { p1 = rdlane(ob, w.total - 1u); p2 = rdlane(ob, w.total - 2u); ctx_regs = true; }
      else if (ctx_regs) { p2 = p1; p1 = rdlane(ob, 0); }  // (else: both come out of memory when a literal asks for them)
      mlen -= (int32_t)w.total;
      P += w.total; quota -= w.total;
    } else {  // its load is issued now, its store when the next command gets here (its source may be what this one writes)
      if (push) { d3 = d2; d2 = d1; d1 = d0; d0 = dist; }
      mlen -= (int32_t)n;
      pend_load8(out + P - (uint32_t)dist, n, lane);
      pend_n = n; pend_pos = P;
      ctx_regs = false;
      P += n; quota -= n;
    }
    SPLIT_LAP(6);
#ifdef BROTLI_AMD_PROFILE_SPLIT
    if (word) n_word++; else if (!committed) n_nolit++;
#endif
    need32();
    rec_ok = request();
  }
#ifdef BROTLI_AMD_PROFILE_SPLIT
  if (blockIdx.x == 0 && lane == 0) { for (int k = 0; k < 7; k++) g_split_prof[24 + k] += lap_acc[k]; g_split_prof[32] += n_run; g_split_prof[33] += n_lit; g_split_prof[34] += n_nolit; g_split_prof[35] += n_word; g_split_prof[36] += ncmd; g_split_prof[37] += 1; g_split_prof[38] += __builtin_amdgcn_s_memtime() - lap_t0; }
#ifdef BROTLI_AMD_PROFILE_RUN_WAIT
  if (blockIdx.x == 0 && lane == 0) g_split_prof[39] += run_wait;
#endif
#endif
  if (!ctx_regs && pend_n >= 2u) { p1 = pend_byte(pend_n - 1u); p2 = pend_byte(pend_n - 2u); ctx_regs = true; }
  flush();
  sp_st(xb, XW_POS, br.next_dw * 32u - br.cnt - origin_bits);
  lds_st32(LDS_LEANWIN + 4u * lane, br.cur);
  if (lane == 0) {
    LEAN_ST(L_CHUNK_BASE, br.chunk_base);
    LEAN_ST(L_P1, p1); LEAN_ST(L_P2, p2); LEAN_ST(L_CTX_REGS, ctx_regs ? 1u : 0u);
    LEAN_ST(L_BUF_LO, (uint32_t)br.buf); LEAN_ST(L_BUF_HI, (uint32_t)(br.buf >> 32)); LEAN_ST(L_CNT, br.cnt); LEAN_ST(L_NEXT_DW, br.next_dw);
    LEAN_ST(L_ISSUED, br.issued_half); LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32)); LEAN_ST(L_QUOTA, quota);
    LEAN_ST(L_MLEN, mlen); LEAN_ST(L_BL0, bl0); LEAN_ST(L_BL1, bl1); LEAN_ST(L_BL2, bl2);
    LEAN_ST(L_D0, d0); LEAN_ST(L_D1, d1); LEAN_ST(L_D2, d2); LEAN_ST(L_D3, d3); LEAN_ST(L_NCMD_LO, ncmd);
    LEAN_ST(L_INSERT, insert_len); LEAN_ST(L_COPY, copy_len); LEAN_ST(L_DCODE, distance_code); LEAN_ST(L_DCTX, distance_context);
    LEAN_ST(L_LITS_LEFT, 0u);
    LEAN_ST(L_SP_WAITED, waited_out ? 1u : long_cmd ? 2u : 0u);
  }
  lds_sync();
  return rfl(stage);
}

// ===================================== split lean loop: the parser =====================================
// lean_commands<false> without the part that moves bytes (see "parse / copy split" above): the same stages, the same
// hand-over to process_commands (L_STAGE ...), but a command that stays clear of every limit becomes a record for the
// copier wave.  Returns with the ring drained: whatever the records produced is in memory.
#define SP_CTX_REG "v125"
__device__ __noinline__ __attribute__((aligned(256))) uint32_t lean_split_commands(uint32_t lut_vgpr, uint32_t ctx_tree_v) {
  const uint32_t lane = lane_id();
  const Arena a = {nullptr, 0xFFFFFFFFu, 0u};
  BitReader br;
  br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
  br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED); br.end_dw = LEAN_LD(L_END_DW);
  br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)LEAN_LD(L_OUT_LO) | ((uint64_t)LEAN_LD(L_OUT_HI) << 32));
  uint64_t P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
  uint32_t quota = LEAN_LD(L_QUOTA);
  int32_t mlen = (int32_t)LEAN_LD(L_MLEN);
  uint32_t bl0 = LEAN_LD(L_BL0), bl1 = LEAN_LD(L_BL1), bl2 = LEAN_LD(L_BL2);
  int32_t d0 = (int32_t)LEAN_LD(L_D0), d1 = (int32_t)LEAN_LD(L_D1), d2 = (int32_t)LEAN_LD(L_D2), d3 = (int32_t)LEAN_LD(L_D3);
  uint32_t ncmd = 0;
  const uint32_t cmd_tree = LEAN_LD(L_CMD_TREE), lit_tree = LEAN_LD(L_LIT_TREE);
  const uint32_t dt0 = LEAN_LD(L_DT0), dt1 = LEAN_LD(L_DT1), dt2 = LEAN_LD(L_DT2), dt3 = LEAN_LD(L_DT3);
  const int32_t max_backward = (int32_t)LEAN_LD(L_MAX_BACKWARD);
  const uint32_t postfix_bits = LEAN_LD(L_POSTFIX), num_direct = LEAN_LD(L_NUM_DIRECT);
  const bool dlut_ok = postfix_bits == 0u && num_direct == 16u;
  uint32_t dlut;
  {
    const uint32_t dv = (lane - 16u) & 63u, nb = (dv >> 1) + 1u;
    dlut = nb | ((((2u + (dv & 1u)) << nb) - 3u) << 5);
  }
  const uint32_t safe_dw = br.end_dw > 72u ? br.end_dw - 72u : 0u;
  uint32_t lit_reg = 0, lit_n = 0;
  uint32_t p1 = LEAN_LD(L_P1), p2 = LEAN_LD(L_P2);
  // where the two bytes before P are: p1/p2 (ctx_regs), on their way from the last copy's source into SP_CTX_REG (ctx_pend),
  // or in memory once the copier has caught up
  bool ctx_regs = LEAN_LD(L_CTX_REGS) != 0u, ctx_pend = false;
  const uint32_t trivial = LEAN_LD(L_TRIVIAL), ctx_lut = LEAN_LD(L_CTX_LUT);
  const uint32_t lut0v = lds_ld32(ctx_lut + 4u * lane), lut1v = lds_ld32(ctx_lut + 256u + 4u * lane);
  const uint32_t sp_rec = sp_rec_base(), sp_lit = sp_lit_base(), sp_ctl = sp_ctl_base();
  uint32_t sp_head = LEAN_LD(L_SP_HEAD), lit_head = LEAN_LD(L_SP_LIT);
  uint32_t sp_tail_c = sp_ld(sp_ctl, CW_TAIL), lit_tail_c = sp_ld(sp_ctl, CW_LIT_TAIL);
  uint32_t failed = 0;
  uint32_t stage = LS_BEGIN;
  int32_t insert_len = 0, copy_len = 0, distance_code = 0;
  uint32_t distance_context = 0, lits_left = 0;
  uint32_t rec_ins = 0, rec_lit = lit_head;  // literals decoded but not posted yet, and where their bytes start in the ring
  // command records (rec_wave): positions are stream bits from dword origin_dw; front_c: positions below have a record
  const uint32_t xb = hc_ld(HC_EXT_BASE), xring = xb + SPX_CTL_BYTES;
  const bool rec_on = LEAN_LD(L_SP_REC) != 0u;
  const uint32_t origin_bits = LEAN_LD(L_SP_ORIGIN) << 5;
  uint32_t my_epoch = 0, front_c = 0;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 rec_v = {0u, 0u};
  bool rec_ok = false;
  if (rec_on) {
    // the prefix codes the records are parsed with: new ones (the first time, after a block switch) start a new epoch
    my_epoch = sp_ld(xb, XW_EPOCH);
    const bool same = my_epoch != 0u && sp_ld(xb, XW_CMD_TREE) == LDS_FIXED + cmd_tree && sp_ld(xb, XW_DT0) == LDS_FIXED + dt0 && sp_ld(xb, XW_DT0 + 1) == LDS_FIXED + dt1 &&
                      sp_ld(xb, XW_DT0 + 2) == LDS_FIXED + dt2 && sp_ld(xb, XW_DT0 + 3) == LDS_FIXED + dt3;
    sp_st(xb, XW_POS, br.next_dw * 32u - br.cnt - origin_bits);
    if (!same) {
      sp_st(xb, XW_CMD_TREE, LDS_FIXED + cmd_tree); sp_st(xb, XW_DT0, LDS_FIXED + dt0); sp_st(xb, XW_DT0 + 1, LDS_FIXED + dt1);
      sp_st(xb, XW_DT0 + 2, LDS_FIXED + dt2); sp_st(xb, XW_DT0 + 3, LDS_FIXED + dt3);
      sp_st(xb, XW_POSTFIX, postfix_bits); sp_st(xb, XW_NUM_DIRECT, num_direct);
      my_epoch++;
      lds_release();
      sp_st(xb, XW_EPOCH, my_epoch);
    }
  }
  // ask for the record of the command that starts where the reader stands (looked at when the command is begun)
#define SP_REC_REQUEST() do { rec_ok = false; \
    if (rec_on) { const uint32_t rel_ = br.next_dw * 32u - br.cnt - origin_bits; \
      if ((ncmd & 3u) == 0u) sp_st(xb, XW_POS, rel_); \
      if (rel_ < 0x40000000u) { \
        if (rel_ >= front_c) { const uint64_t f_ = *reinterpret_cast<volatile __attribute__((address_space(3))) uint64_t*>(&g_smem[xb + 4u * XW_FRONT]); \
          front_c = rfl((uint32_t)(f_ >> 32)) == my_epoch ? rfl((uint32_t)f_) : 0u; lds_acquire(); SPLIT_COUNT(14, 1); } \
        if (rel_ < front_c) { rec_v = *reinterpret_cast<__attribute__((address_space(3))) const u32x2*>(&g_smem[xring + ((rel_ & sp_ld(xb, XW_RING_MASK)) << 3)]); rec_ok = true; } } } } while (0)

  // one record: lane 0 writes its four words, then the head (LDS operations of a wave execute in order)
#define SP_POST(W0_, W1_, W2_, W3_) do { \
    if (sp_head - sp_tail_c >= SP_RECS - 1u) { const uint64_t t_ = SPLIT_T(); uint32_t spins_ = 0; \
      do { __builtin_amdgcn_s_sleep(1); sp_tail_c = sp_ld(sp_ctl, CW_TAIL); if (++spins_ > SP_SPIN_CAP) { failed = 1; break; } } while (sp_head - sp_tail_c >= SP_RECS - 1u); \
      SPLIT_PROF(0, t_); } \
    const u32x4 wv_ = {(uint32_t)(W0_), (uint32_t)(W1_), (uint32_t)(W2_), (uint32_t)(W3_)}; \
    const uint32_t ra_ = sp_rec + ((sp_head & (SP_RECS - 1u)) << 4); \
    sp_head++; \
    asm volatile("" ::: "memory");  /* (the literal bytes are written; no hardware wait: DS operations stay in order) */ \
    if (lane == 0) { *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(&g_smem[ra_]) = wv_; \
                     asm volatile("" ::: "memory"); \
                     *reinterpret_cast<lds_vu32*>(&g_smem[sp_ctl + 4u * CW_HEAD]) = sp_head; } } while (0)
  // the copier has executed every record and its stores have completed
#define SP_DRAIN() do { const uint64_t t_ = SPLIT_T(); uint32_t spins_ = 0; \
    while (sp_ld(sp_ctl, CW_DONE_SEQ) != sp_head) { __builtin_amdgcn_s_sleep(1); if (++spins_ > SP_SPIN_CAP) { failed = 1; break; } } \
    lds_acquire(); SPLIT_PROF(1, t_); } while (0)
  // literals collected in lit_reg go to the literal ring
#define SP_LIT_FLUSH() do { if (lit_n) { if (lane < lit_n) lds_st8(sp_lit + ((lit_head + lane) & (SP_LIT_BYTES - 1u)), lit_reg); lit_head += lit_n; lit_n = 0; } } while (0)

  const uint64_t split_t0 = SPLIT_T();
  SP_POST(SPR_SETP, (uint32_t)P, (uint32_t)(P >> 32), 0u);

  br.need32();
  uint32_t next_root = lds_ld16(LDS_FIXED + cmd_tree + (((uint32_t)br.buf & 0xFFu) << 1));
  SP_REC_REQUEST();

  // a distance symbol 1..15: one of the last four distances, +- up to 3 (TakeDistanceFromRingBuffer, decode.rs:2017-2049)
  auto ring_distance = [&](uint32_t code) -> int32_t {
    const uint32_t sh = code << 1;
    const uint32_t back = 3u - ((0xaaafff1bu >> sh) & 3u);
    int32_t v = back == 0 ? d0 : back == 1 ? d1 : back == 2 ? d2 : d3;
    const int32_t mag = (int32_t)((0xfa5fa500u >> sh) & 3u);
    if (code & 1u) v += mag;
    else { v -= mag; if (v <= 0) v = 0x7fffffff; }
    return v;
  };

#ifdef BROTLI_AMD_PROFILE_SPLIT
  uint64_t lap_acc[6] = {0, 0, 0, 0, 0, 0}; uint64_t lap_t = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    SPLIT_LAP(5);
    if (bl1 == 0 || br.next_dw >= safe_dw || failed) { stage = LS_BEGIN; break; }
    // how far the copier's stores have got (asked for now, looked at when this command's distance is known)
    const uint32_t done_v = *reinterpret_cast<lds_vu32*>(&g_smem[sp_ctl + 4u * CW_DONE_P]);
    bool have_dist = false;
    uint32_t rec_lo = 0, rec_hi = 0;
    if (rec_ok) { rec_lo = rfl(rec_v.x); rec_hi = rfl(rec_v.y); }
    // a record that is the exact parse of this position (an explicit distance needs its block count)
    if ((rec_lo & XR_VALID) != 0u && ((rec_lo & (XR_LITERALS | XR_IMPLICIT)) != 0u || bl2 != 0u)) {
      SPLIT_COUNT(15, 1);
      copy_len = (int32_t)(rec_lo & 0xFFFFu);
      distance_context = (rec_lo >> XR_DCTX_SHIFT) & 3u;
      distance_code = (rec_lo & XR_IMPLICIT) ? 0 : -1;
      insert_len = 0;
      if (rec_lo & XR_LITERALS) insert_len = (int32_t)rec_hi;
      else if (!(rec_lo & XR_IMPLICIT)) {
        have_dist = true;
        distance_context = 0;
        if (rec_lo & XR_SHORT) {
          if (rec_hi == 0u) { distance_code = d0; distance_context = 1; }
          else distance_code = ring_distance(rec_hi);
        } else distance_code = (int32_t)rec_hi;
        bl2--;
      }
      br.advance((rec_lo >> 16) & 127u);
    } else {
      uint32_t cmd;
      {
        uint32_t e = rfl(next_root);
        uint32_t len = e & 15u;
        if (len > ROOT_BITS) {
          uint32_t idx = (e >> 4) + (((uint32_t)br.buf >> ROOT_BITS) & mask_bits(len - ROOT_BITS));
          e = rfl(lds_ld16(LDS_FIXED + cmd_tree + (idx << 1)));
          len = ROOT_BITS + (e & 15u);
        }
        br.drop(len);
        cmd = e >> 4;
      }
      uint32_t cell = cmd >> 6;
      uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u);
      uint32_t copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);
      uint32_t ie = rdlane(lut_vgpr, ins_code), ce = rdlane(lut_vgpr, 32u + copy_code);
      distance_code = cmd < 128 ? 0 : -1;
      distance_context = copy_code > 2 ? 3u : copy_code;
      insert_len = (int32_t)((ie & 0xFFFFu) + br.read24(ie >> 16));
      copy_len = (int32_t)((ce & 0xFFFFu) + br.read24(ce >> 16));
    }
    SPLIT_LAP(0);
    bl1--;
    ncmd++;
    lits_left = (uint32_t)insert_len;
    rec_ins = 0; rec_lit = lit_head;
    if (insert_len != 0) {
      if ((uint32_t)insert_len > quota || (uint32_t)insert_len > bl0 || (uint32_t)insert_len > SP_MAX_INSERT) { stage = LS_AFTER_HEAD; break; }
      mlen -= insert_len;
      uint32_t i = (uint32_t)insert_len;
      if (!ctx_regs) {
        if (ctx_pend) {  // the last copy's source bytes, asked for when its distance was known
          const uint64_t t_ = SPLIT_T();
          asm volatile("s_waitcnt vmcnt(0)\n\tv_readlane_b32 %0, " SP_CTX_REG ", 0\n\tv_readlane_b32 %1, " SP_CTX_REG ", 1" : "=s"(p1), "=s"(p2) :: "memory");
          SPLIT_PROF(2, t_); SPLIT_COUNT(8, 1);
        } else {  // they were not in memory yet when the copy was posted (or the copy repeats itself): wait for the copier
          SP_DRAIN();
          SPLIT_COUNT(9, 1);
          const uint64_t t_ = SPLIT_T();
          p1 = P >= 1 ? (uint32_t)rfl(out[P - 1]) : 0u;
          p2 = P >= 2 ? (uint32_t)rfl(out[P - 2]) : 0u;
          SPLIT_PROF(3, t_);
        }
        ctx_regs = true; ctx_pend = false;
      }
      if (lit_head + i - lit_tail_c > SP_LIT_BYTES) {  // room in the literal ring
        const uint64_t t_ = SPLIT_T(); uint32_t spins_ = 0;
        do { __builtin_amdgcn_s_sleep(1); lit_tail_c = sp_ld(sp_ctl, CW_LIT_TAIL); if (++spins_ > SP_SPIN_CAP) { failed = 1; break; } } while (lit_head + i - lit_tail_c > SP_LIT_BYTES);
        SPLIT_PROF(0, t_);
      }
      while (i > 0 && br.next_dw < safe_dw) {
        uint32_t tree = lit_tree;
        if (!trivial) {
          const uint32_t context = ((rdlane(lut0v, p1 >> 2) >> ((p1 & 3u) << 3)) | (rdlane(lut1v, p2 >> 2) >> ((p2 & 3u) << 3))) & 0xFFu;
          tree = rdlane(ctx_tree_v, context);
        }
        uint32_t lit = read_symbol<true>(br, a, tree);
        p2 = p1; p1 = lit;
        lit_reg = (lane == lit_n) ? lit : lit_reg;
        lit_n++;
        if (lit_n == 64) SP_LIT_FLUSH();
        i--;
      }
      SP_LIT_FLUSH();
      const uint32_t done = (uint32_t)insert_len - i;
      P += done; bl0 -= done; quota -= done;
      rec_ins = done;
      lits_left = i;
      if (i != 0) { stage = LS_LITERALS_REST; break; }
      if (quota == 0) { stage = LS_LITERALS_AT_LIMIT; break; }
    }
    SPLIT_LAP(1);
    // ---- distance (ReadDistanceInternal, decode.rs:2066-2131; see process_commands) ----
    if (have_dist) {
    } else if (distance_code >= 0) {
      distance_context = 1;
      distance_code = d0;
    } else {
      if (bl2 == 0) { stage = LS_DISTANCE; break; }
      uint32_t dtree = distance_context == 0 ? dt0 : distance_context == 1 ? dt1 : distance_context == 2 ? dt2 : dt3;
      uint32_t code = read_symbol<true>(br, a, dtree);
      distance_context = 0;
      if (code < 16) {
        if (code == 0) {
          distance_code = d0;
          distance_context = 1;
        } else distance_code = ring_distance(code);
      } else if (dlut_ok && code < 64u) {
        const uint32_t de = rdlane(dlut, code);
        distance_code = (int32_t)((de >> 5) + br.read(de & 31u));
      } else {
        int32_t distval = (int32_t)code - (int32_t)num_direct;
        int32_t dc = (int32_t)code;
        if (distval >= 0) {
          int32_t postfix = distval & (int32_t)mask_bits(postfix_bits);
          distval >>= postfix_bits;
          uint32_t nbits = ((uint32_t)distval >> 1) + 1;
          uint32_t bits = br.read(nbits);
          int64_t offset = (int64_t)(int32_t)((((uint32_t)(distval & 1) + 2u) << nbits) - 4u);
          dc = (int32_t)(((offset + (int64_t)bits) << postfix_bits) + postfix + (int64_t)num_direct);
        }
        distance_code = (int32_t)((uint32_t)dc - 16u + 1u);
      }
      bl2--;
      if (br.next_dw > br.end_dw) { stage = LS_NEEDS_INPUT; break; }
    }
    SPLIT_LAP(2);
    br.need32();
    next_root = lds_ld16(LDS_FIXED + cmd_tree + (((uint32_t)br.buf & 0xFFu) << 1));
    SP_REC_REQUEST();
    // ---- the copy becomes a record: an LZ77 reference (not the dictionary) inside the quota ----
    {
      const uint32_t n = (uint32_t)copy_len, dist = (uint32_t)distance_code;
      const int32_t max_distance = (P < (uint64_t)(uint32_t)max_backward) ? (int32_t)P : max_backward;
      if (distance_code > max_distance || distance_code <= 0 || n > quota) { stage = LS_POST_DISTANCE; break; }
    SPLIT_LAP(3);
      if (distance_context == 0) { d3 = d2; d2 = d1; d1 = d0; d0 = distance_code; }
      mlen -= copy_len;
      SP_POST((uint32_t)SPR_COPY | (rec_ins << 8), n, dist, rec_lit);
      rec_ins = 0;
      // the two bytes before the next command's first literal are the last two of this copy = of its source, if that is
      // in memory already and the copy does not repeat itself (copy lengths start at 2)
      ctx_regs = false; ctx_pend = false;
      const uint64_t s2 = P + n - 2u - dist;  // (>= 0: dist <= P)
      if (dist >= n && (int32_t)(rfl(done_v) - (uint32_t)(s2 + 2u)) >= 0) {
        asm volatile("s_mov_b64 exec, 3\n\tglobal_load_ubyte " SP_CTX_REG ", %0, %1\n\ts_mov_b64 exec, -1" :: "v"(1u - lane), "s"(out + s2) : "memory", SP_CTX_REG);
        ctx_pend = true;
      }
    SPLIT_LAP(4);
      P += n;
      quota -= n;
      if (quota == 0) { stage = LS_COMMAND_DONE; break; }
    }
  }
  // literals of a command that did not get to its copy here: a record of their own
  if (rec_ins != 0u) SP_POST((uint32_t)SPR_COPY | (rec_ins << 8), 0u, 1u, rec_lit);
  if (!ctx_regs && ctx_pend) {
    asm volatile("s_waitcnt vmcnt(0)\n\tv_readlane_b32 %0, " SP_CTX_REG ", 0\n\tv_readlane_b32 %1, " SP_CTX_REG ", 1" : "=s"(p1), "=s"(p2) :: "memory");
    ctx_regs = true;
  }
  SP_DRAIN();
#ifdef BROTLI_AMD_PROFILE_SPLIT
  if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 6; k++) g_split_prof[16 + k] += lap_acc[k];
#endif
  SPLIT_PROF(4, split_t0); SPLIT_COUNT(10, 1); SPLIT_COUNT(11, ncmd); SPLIT_COUNT(7, stage == LS_POST_DISTANCE ? 1 : 0);
#undef SP_POST
#undef SP_DRAIN
#undef SP_LIT_FLUSH
#undef SP_REC_REQUEST
  if (rec_on) sp_st(xb, XW_POS, br.next_dw * 32u - br.cnt - origin_bits);
  if (failed) hc_st(HC_FAILED, 1);
  lds_st32(LDS_LEANWIN + 4u * lane, br.cur);
  if (lane == 0) {
    LEAN_ST(L_CHUNK_BASE, br.chunk_base);
    LEAN_ST(L_P1, p1); LEAN_ST(L_P2, p2); LEAN_ST(L_CTX_REGS, ctx_regs ? 1u : 0u);
    LEAN_ST(L_BUF_LO, (uint32_t)br.buf); LEAN_ST(L_BUF_HI, (uint32_t)(br.buf >> 32)); LEAN_ST(L_CNT, br.cnt); LEAN_ST(L_NEXT_DW, br.next_dw);
    LEAN_ST(L_ISSUED, br.issued_half); LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32)); LEAN_ST(L_QUOTA, quota);
    LEAN_ST(L_MLEN, mlen); LEAN_ST(L_BL0, bl0); LEAN_ST(L_BL1, bl1); LEAN_ST(L_BL2, bl2);
    LEAN_ST(L_D0, d0); LEAN_ST(L_D1, d1); LEAN_ST(L_D2, d2); LEAN_ST(L_D3, d3); LEAN_ST(L_NCMD_LO, ncmd);
    LEAN_ST(L_INSERT, insert_len); LEAN_ST(L_COPY, copy_len); LEAN_ST(L_DCODE, distance_code); LEAN_ST(L_DCTX, distance_context);
    LEAN_ST(L_LITS_LEFT, lits_left);
    LEAN_ST(L_SP_HEAD, sp_head); LEAN_ST(L_SP_LIT, lit_head);
  }
  lds_sync();
  return rfl(stage);
}

// (tree cache, context-modelled literals: does every literal block type's context map name at most `slots` trees?)
__device__ __forceinline__ bool lit_types_fit_cache(const Arena& ar, const uint32_t nbt0, const uint32_t ctx_map, const uint32_t slots) {
  for (uint32_t bt = 0; bt < nbt0; bt++) {
    const uint32_t mine = ar.ld8_lane<false>(ctx_map + (bt << 6) + lane_id());
    uint64_t todo = ~0ull; uint32_t n = 0;
    while (todo != 0ull) { const uint32_t idx = rdlane(mine, (uint32_t)__builtin_ctzll(todo)); todo &= ~__ballot(mine == idx); n++; }
    if (n > slots) return false;
  }
  return true;
}

// ===================================== the command loop (hot path) =====================================
// Argument block of the command loop.  The loop is a real function (one per table placement) so that it gets a
// register allocation of its own: everything uniform lives in SGPRs for the whole metablock and nothing of the
// (large, cold) header code competes for them.
struct HotArgs {
  BitReader br;
  Arena ar;
  gu8* out; gcu8* dict;
  uint64_t out_cap, P, next_boundary, rb_size;
  uint32_t window_bits, large_window;
  int32_t mlen, max_backward;
  int32_t d0, d1, d2, d3;
  uint32_t bl0, bl1, bl2;
  uint32_t postfix_bits, num_direct;
  uint32_t lut_vgpr, bl_vgpr;
  uint64_t spec_scratch;   // global address of the helper waves' literal scratch
  uint64_t num_commands;
  uint32_t engine_commands, reserved_;  // commands a command engine (scan or path) took
  uint32_t general_engine;  // the stream has words of the static dictionary: the path engine's general form from here on (see PE_CFG_DICT)
  uint64_t resume_out;     // global address of the status' BrotliAmdResume: command boundaries close to the end of the input are noted there
  uint64_t prof[6];
};

#ifdef BROTLI_AMD_GANG_KERNEL
// What the gang made of an invocation: it sits out one invocation where it took nothing because it met a literal run that wants regions of its own first
// thing or because its regions filled their closure's room, twice as many every time that happens again before it has taken anything, up to 64
__device__ __noinline__ void gang_took(HotArgs* args, const uint32_t took, const uint32_t form_raw) {
  const uint32_t counts = rfl(args->general_engine);
  uint32_t pen = (counts >> 8) & 0xFFu, hold = 0u;
  if (took >= 64u) pen = 0u;
  if (((form_raw >> 12) & 1u) != 0u) { }   // (a pool that has sent this stream nobody yet: nothing to learn from)
  else if (((form_raw >> 11) & 1u) != 0u || (((form_raw >> 8) & 1u) != 0u && took == 0u)) {
    pen = pen == 0u ? 1u : pen >= 32u ? 64u : pen * 2u;
    hold = pen;
  }
  args->general_engine = (counts & 0xFFu) | (pen << 8) | (hold << 16);
}
#endif

// src/decode.rs:2330-2744 with a flat output buffer.
//  * literals are collected one per lane (v_writelane) and stored 64 at a time;
//  * a copy of <= 64 bytes is split in two: its load is issued when the command is decoded, its store when the
//    next command needs the memory pipe -- the load latency overlaps the decode of the next command;
//  * long copies move 16 bytes per lane per step when source and destination are at least a step apart;
//  * the last two output bytes (literal context) are taken from registers where they still are, from memory
//    only after a long copy.
// (CACHED: the LDS part of the arena as a cache of the trees in use, see run_commands -- an instantiation of its own, so that the
// loop for tables that fit LDS stays the code it was: with the cache as a run-time switch the metric lost 1.3 %)
template <bool LDS_ONLY, bool CTX_NEVER, bool CACHED = false>
__device__ __noinline__ int process_commands(HotArgs* args) {
  BitReader br = args->br; br.uniformize();
  Arena a_ = args->ar; a_.uniformize();
  const Arena a = a_;
  const uint32_t lane = lane_id();
  gu8* const out = rfl_ptr(args->out);
  gcu8* const dict = rfl_ptr(args->dict);
  const uint64_t out_cap = rfl(args->out_cap);
  uint64_t P = rfl(args->P), next_boundary = rfl(args->next_boundary);
  const uint64_t rb_size = rfl(args->rb_size);
  const bool full_ring = rb_size == (1ull << rfl(args->window_bits));
  int32_t mlen = rfl(args->mlen);
  const int32_t max_backward = rfl(args->max_backward);
  // last four distances, most recent first (the reference's dist_rb/dist_rb_idx ring as a shift register: short
  // code 0 and dictionary references leave it untouched, every other LZ77 distance is pushed; state.rs:295-296)
  int32_t d0 = rfl(args->d0), d1 = rfl(args->d1), d2 = rfl(args->d2), d3 = rfl(args->d3);
  uint32_t bl0 = rfl(args->bl0), bl1 = rfl(args->bl1), bl2 = rfl(args->bl2);
  // What only block switches need (block-type and block-length trees, number of block types, the block-type rings
  // of state.rs:429-435, where the tree groups and maps are) stays in LDS, put there by run_commands():
  enum { H_BT_TREE = 0, H_BL_TREE = 3, H_NBT = 6, H_CTX_MODES = 9, H_CTX_MAP = 10, H_DIST_CTX_MAP = 11, H_LIT_TREES = 12, H_CMD_TREES = 13,
         H_DIST_TREES = 14, H_RING = 15 /* + 2 * category: second last, last block type */ };
#define HOTC(k) rfl(lds_ld32(LDS_HOT + 4u * (uint32_t)(k)))
#define BLOCK_SWITCH(cat, bl, res) do { \
    uint32_t t0_ = HOTC(H_RING + 2 * (cat)), t1_ = HOTC(H_RING + 2 * (cat) + 1); \
    (res) = block_switch<false>(br, a, bl_vgpr, HOTC(H_BT_TREE + (cat)), HOTC(H_BL_TREE + (cat)), HOTC(H_NBT + (cat)), bl, t0_, t1_); \
    if ((res) == BS_SWITCHED) { if (lane == 0) { lds_st32(LDS_HOT + 4u * (H_RING + 2 * (cat)), t0_); lds_st32(LDS_HOT + 4u * (H_RING + 2 * (cat) + 1), t1_); } lds_sync(); } \
  } while (0)
  const uint32_t postfix_bits = rfl(args->postfix_bits), num_direct = rfl(args->num_direct);
  const uint32_t lut_vgpr = args->lut_vgpr, bl_vgpr = args->bl_vgpr;  // per-lane LUT images
  uint64_t num_commands = rfl(args->num_commands);
  uint32_t engine_commands = rfl(args->engine_commands);
  int result = E_SUCCESS;
  uint64_t prof_cmd = 0, prof_lit = 0, prof_dist = 0, prof_copy = 0, prof_t = PROF_T();
  (void)prof_cmd; (void)prof_lit; (void)prof_dist; (void)prof_copy; (void)prof_t;
  uint64_t prof_fast_batches = 0, prof_fast_syms = 0; (void)prof_fast_batches; (void)prof_fast_syms;
  uint32_t prof_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)prof_stage;

  // (the LDS part as a cache of the trees in use: see run_commands)
  uint32_t rec_base = 0;   // (the command records' ring, where wave 2 parses ahead: see below)
  constexpr bool tree_cache = CACHED;
  // (cached tables: a tree that changes keeps its address -- the records' parser is told by a word of the ring's header that the
  // codes it parses with are no longer the ones, and lean_rec_commands starts a new epoch: it compares addresses)
  auto records_stale = [&]() { if (tree_cache && rec_base != 0u) { if (lane == 0) lds_st32(rec_base + 4u * (uint32_t)XW_CMD_TREE, 0u); lds_sync(); } };
  static_assert(!CACHED || LDS_ONLY, "the cache is for the loops that read their tables out of LDS");
  auto cached_tree = [&](const uint32_t tree, const uint32_t slot, const uint32_t bytes) -> uint32_t {
    if (!tree_cache) return tree;
    // (a dword a lane: the arena aligns a tree to four bytes, no more -- ADVICE round 4)
    for (uint32_t off = lane * 4u; off < bytes; off += 256u)
      lds_st32(LDS_FIXED + slot + off, *reinterpret_cast<gu32*>(a.glb + tree + off));
    lds_sync();
    return slot;
  };
  uint32_t cmd_tree = cached_tree(a.ld32<false>(HOTC(H_CMD_TREES) + HOTC(H_RING + 3) * 4), TREE_CACHE_CMD, TREE_CACHE_CMD_BYTES);
  uint32_t ctx_slice = 0, lit_tree = 0, trivial = 0, ctx_lut = LDS_CTX_LUT, lit_zero = 0, ctx_tree_v = 0;
  // PrepareLiteralDecoding, decode.rs:1554-1570
  auto prepare_literal = [&]() {
    uint32_t bt = HOTC(H_RING + 1);
    ctx_slice = bt << 6;
    // trivial <=> all 64 map entries of the block type are equal (DetectTrivialLiteralBlockTypes, 1525-1553)
    uint32_t mine = a.ld8_lane<false>(HOTC(H_CTX_MAP) + ctx_slice + lane);
    uint32_t first = rdlane(mine, 0);
    trivial = (__ballot(mine != first) == 0ull) ? 1u : 0u;
    lit_tree = cached_tree(a.ld32<false>(HOTC(H_LIT_TREES) + first * 4), TREE_CACHE_LIT, TREE_CACHE_LIT_BYTES);
    // lane c: the tree of literal context c in this block type (context map and tree group folded into one readlane)
    if (!CTX_NEVER) {
      uint32_t toff = HOTC(H_LIT_TREES) + mine * 4;
      ctx_tree_v = toff < a.lds_limit ? lds_ld32(LDS_FIXED + toff) : (uint32_t)*reinterpret_cast<gu32*>(a.glb + toff);
      if (tree_cache) {   // (the block type's trees, each once, into the cache's literal slots: the caller has counted them)
        uint64_t todo = ~0ull; uint32_t slot = TREE_CACHE_BYTES;
        while (todo != 0ull) {
          const uint32_t l = (uint32_t)__builtin_ctzll(todo);
          const uint32_t idx = rdlane(mine, l), t = rdlane(ctx_tree_v, l);
          const uint64_t same = __ballot(mine == idx) & todo;
          (void)cached_tree(t, slot, TREE_CACHE_LIT_BYTES);
          if (mine == idx) ctx_tree_v = slot;
          todo &= ~same; slot += TREE_CACHE_LIT_STRIDE;
        }
        lit_tree = rdlane(ctx_tree_v, 0);
      }
    }
    ctx_lut = LDS_CTX_LUT + 512u * (a.ld8<false>(HOTC(H_CTX_MODES) + bt) & 3u);
    if (LDS_ONLY) lit_zero = (rfl(lds_ld16(LDS_FIXED + lit_tree)) & 15u) == 0u ? 1u : 0u;  // one-symbol code: zero bits per literal
  };
  prepare_literal();
  // the four distance trees of the current distance block type, one per distance context (decode.rs:2566-2570)
  uint32_t dt0 = 0, dt1 = 0, dt2 = 0, dt3 = 0;
  auto prepare_distance = [&]() {
    uint32_t m = HOTC(H_DIST_CTX_MAP) + (HOTC(H_RING + 5) << 2), g = HOTC(H_DIST_TREES);
    dt0 = a.ld32<false>(g + a.ld8<false>(m + 0) * 4); dt1 = a.ld32<false>(g + a.ld8<false>(m + 1) * 4);
    dt2 = a.ld32<false>(g + a.ld8<false>(m + 2) * 4); dt3 = a.ld32<false>(g + a.ld8<false>(m + 3) * 4);
    if (tree_cache) {   // (contexts that share a tree share its copy: the path engine asks whether the four are one)
      const uint32_t g0 = dt0, g1 = dt1, g2 = dt2, g3 = dt3;
      dt0 = cached_tree(g0, TREE_CACHE_DIST, TREE_CACHE_DIST_BYTES);
      dt1 = g1 == g0 ? dt0 : cached_tree(g1, TREE_CACHE_DIST + TREE_CACHE_DIST_BYTES, TREE_CACHE_DIST_BYTES);
      dt2 = g2 == g0 ? dt0 : g2 == g1 ? dt1 : cached_tree(g2, TREE_CACHE_DIST + 2u * TREE_CACHE_DIST_BYTES, TREE_CACHE_DIST_BYTES);
      dt3 = g3 == g0 ? dt0 : g3 == g1 ? dt1 : g3 == g2 ? dt2 : cached_tree(g3, TREE_CACHE_DIST + 3u * TREE_CACHE_DIST_BYTES, TREE_CACHE_DIST_BYTES);
      records_stale();
    }
  };
  prepare_distance();
  // Literal context never matters in this metablock when every literal block type has a trivial context map
  // (CTX_NEVER, found by the caller): then p1/p2 need not be tracked at all, which saves reading back the tail of
  // every long copy, and the lean stages below apply.
  constexpr bool ctx_never = CTX_NEVER;

  // ---- lean commands ----
  // A command whose whole effect stays clear of every limit (declared metablock length, end of the output buffer,
  // next ring-buffer flush point, current literal block, end of the input) runs through stages that do not check
  // any of them again: `quota` is the number of bytes that can be produced before the first of the output-side
  // limits, recomputed after every command that went through the checked stages.
  constexpr bool quota_mb = LDS_ONLY;              // lean copies and lean context-modelled literals
  const uint32_t safe_dw = br.end_dw > 72u ? br.end_dw - 72u : 0u;  // a 64-dword register window that starts below lies inside the stream
  uint32_t quota = 0;
#define RECOMPUTE_QUOTA() do { \
    uint64_t room_ = out_cap - P; \
    uint64_t q_ = room_; \
    uint64_t rb_ = next_boundary > P ? next_boundary - P : 0; \
    if (rb_ < q_) q_ = rb_; \
    uint32_t m_ = mlen > 0 ? (uint32_t)mlen : 0u; \
    quota = q_ < (uint64_t)m_ ? (uint32_t)q_ : m_; } while (0)
  if (quota_mb) RECOMPUTE_QUOTA();

  // ---- output side state ----
  // literal run being collected: lane k holds literal k, lit_n of them, first one goes to out[lit_pos]
  uint32_t lit_reg = 0, lit_n = 0; uint64_t lit_pos = P;
  // short copy / dictionary word whose bytes are in registers (lane k = byte k) but not stored yet
  uint32_t pend_reg = 0, pend_n = 0; uint64_t pend_pos = 0;
  // the same for a copy of up to 1 KiB: 16 bytes per lane (its sub-16-byte tail goes through pend_reg)
  u32x4 pendv = {0, 0, 0, 0}; uint32_t pendv_n16 = 0; uint64_t pendv_pos = 0;
  // where the two bytes before P (literal context) currently are
  enum { CTX_REGS = 0, CTX_PEND = 1, CTX_MEMORY = 2 };
  uint32_t ctx_src = CTX_MEMORY, ctx_len = 0;  // CTX_PEND: last ctx_len bytes of output are pend_reg[0..ctx_len)
  uint32_t p1 = 0, p2 = 0;  // stream start counts as two zero bytes (decode.rs:1859-1860)
  if (P == 0) ctx_src = CTX_REGS;

#define FLUSH_LITERALS() do { if (lit_n) { if (lane < lit_n) out[lit_pos + lane] = (uint8_t)lit_reg; lit_pos += lit_n; lit_n = 0; } } while (0)
#define FLUSH_PENDING() do { if (pendv_n16) { if (lane < pendv_n16) *reinterpret_cast<gu32x4*>(out + pendv_pos + (uint64_t)lane * 16) = pendv; pendv_n16 = 0; } \
                             if (pend_n) { if (lane < pend_n) out[pend_pos + lane] = (uint8_t)pend_reg; pend_n = 0; } } while (0)
#define STOP(e) do { result = (e); goto done; } while (0)
  // ring-buffer flush points (decode.rs:1693-1738, 3299-3344): crossing one with a negative remaining length is
  // BLOCK_LENGTH_1; the last crossed one is what the caller has received when an error is reported
#define RING_CROSS() do { while (P >= next_boundary) { if (mlen < 0) STOP(E_BLOCK_LENGTH_1); if (!full_ring) STOP(E_UNREACHABLE); next_boundary += rb_size; } } while (0)

  // per-command values (declared out here so that the lean loop's hand-over can jump to any stage below)
  int32_t insert_len = 0, copy_len = 0, distance_code = 0, lits_left = 0, max_distance = 0;
  uint32_t distance_context = 0;

  if (LDS_ONLY && lane == 0) {  // what lean_commands needs and never changes in this metablock
    LEAN_ST(L_END_DW, br.end_dw); LEAN_ST(L_MAX_BACKWARD, max_backward); LEAN_ST(L_POSTFIX, postfix_bits); LEAN_ST(L_NUM_DIRECT, num_direct);
    LEAN_ST(L_OUT_LO, (uint32_t)(uintptr_t)out); LEAN_ST(L_OUT_HI, (uint32_t)((uint64_t)(uintptr_t)out >> 32));
    LEAN_ST(L_SPEC_LO, (uint32_t)rfl(args->spec_scratch)); LEAN_ST(L_SPEC_HI, (uint32_t)(rfl(args->spec_scratch) >> 32));
  }

  // ---- the command engine (brotli_scan_engine.h): blocks of sixteen waves, metablocks whose literals never depend on
  // context (large-window streams: the path engine only).  It takes commands until one needs the checked code below (or the input runs short) and
  // hands the stream back in front of that command; an invocation that got nowhere makes the next ones rarer.
  const bool scan_block = LDS_ONLY && CTX_NEVER && hc_ld(HC_SCAN_BASE) != 0u;
  const bool large_window = rfl(args->large_window) != 0u;   // (the path engine takes such streams since round 4, the scan engine does not)
  uint32_t scan_fails = 0;
#ifdef BROTLI_AMD_PROFILE_SCAN
  uint64_t pp_exit = 0, pp_enter = 0; (void)pp_enter; const uint64_t pp_start = __builtin_amdgcn_s_memtime(); bool pp_first = true; (void)pp_first;
#endif
  const uint32_t engine_hints = rfl(args->general_engine);   // (bit 0: the general form; bits 8 .. 15, 16 .. 23: the gang's penalty and the invocations it still sits out, see remote_off)
  bool prefer_one_engine = false;   // the next invocation of the path engine: its one-engine form (see `declined` below)
  // ... the one-block form though the stream has a gang of blocks: for the next invocations where the gang met a literal run that wants regions
  // of its own first thing (it takes nothing then, and a region's tables of every block are lost: 8.1 against 5.8 ms on 4 MiB of high-entropy
  // literals, all runs) -- one invocation the first time, twice as many every time it happens again before the gang has taken anything, up to 64;
  // the same where its regions fill their closure's room (few literals: the one-block form halves its regions there, or hands the stream to the
  // scan engine; sitting out the rest of the metablock instead left 18 % of a 1 GiB stream's commands to one block, three quarters of its time).
  // Both counts go with the stream from metablock to metablock (HotArgs::general_engine).
  // (the counts and the two functions that keep them -- gang_wanted, gang_took -- are not this function's: what the gang asks of it lives in
  // HotArgs, in words of the mailbox and in code of its own, because this function has no register to spare: with the counts in its registers and
  // the decision in its body its frame was 432 bytes a lane where it is 128 without, and text through the gangs' kernel 8 % slower)
  bool prefer_scan = false;         // ... or the scan engine: the path engine found its regions bound by their closure (see there)
  bool prefer_general = (engine_hints & 1u) != 0u;   // ... or the path engine's general form: the lean one has stopped in front of a dictionary reference in this stream
  // ---- helper waves of a context-modelled metablock (LDS tables, a block of four or more waves): wave 2 parses command records
  // ahead of this wave (rec_wave; the lean loop takes commands out of them); on request (BROTLI_AMD_ENGINE=split) wave 1 executes
  // what this wave parses (copier_wave, lean_split_commands).  They stay engaged, idle while the checked stages run, until the
  // metablock is done.
  bool split_on = false, helpers_on = false;
  // (round 6) ... and of a metablock WITHOUT context too, in blocks that have helper waves and no command engine: the records and the hand-written run take
  // text -- a dozen bytes a command -- at 2.4 times what the one-wave loop does (1024 x lcet10 at -q 5: 10.4 -> 25 GB/s); a stream of long copies and long
  // literal runs leaves the run at every command, which is noticed after three calls that took next to nothing (rec_off; BROTLI_AMD_ENGINE=norecall: never)
  const bool rec_plain = CTX_NEVER && (g_engine_mode & 32u) == 0u && !scan_block;
  if (LDS_ONLY && (!CTX_NEVER || rec_plain) && hc_ld(HC_NW_ALL) >= 4u && hc_ld(HC_KIND) != (uint32_t)HK_NO_ROUNDS && mlen >= (int32_t)SP_MIN_MLEN && (g_engine_mode & 6u) != 6u) {
    split_on = !CTX_NEVER && (g_engine_mode & 2u) == 0u;
    // the records' ring: what is left of the LDS arena now that the tables of this metablock are built (no large window: a
    // record's distance has at most 24 extra bits); positions count from the dword the reader is in now
    const uint32_t free_at = tree_cache ? TREE_CACHE_BYTES + tree_cache_lit_slots(rfl(args->reserved_)) * TREE_CACHE_LIT_STRIDE : (a.top + 15u) & ~15u;   // (cached tables: behind the cache; `reserved_` is the LDS part's real size)
    const uint32_t lds_room = tree_cache ? rfl(args->reserved_) : a.lds_limit;
    const uint32_t origin_dw = (br.next_dw - ((br.cnt + 31u) >> 5)) & ~1u;
    const uint32_t ring_pos = lds_room >= free_at + SPX_BYTES ? SPX_POS : SPX_POS_MIN;   // (round 6: half a ring where the tables leave no room for a whole one -- text at -q 11 in blocks of four waves)
    if ((g_engine_mode & 4u) == 0u && lds_room >= free_at + SPX_CTL_BYTES + ring_pos * 8u && rfl(args->large_window) == 0u && br.end_dw > origin_dw + 80u) rec_base = LDS_FIXED + free_at;
#ifdef BROTLI_AMD_REC_DEBUG
    if (blockIdx.x == 0 && lane == 0) printf("records: room %u free_at %u need %u cache %d -> %s\n", lds_room, free_at, (uint32_t)SPX_BYTES, (int)tree_cache, rec_base != 0u ? "yes" : "no");
#endif
    if (split_on || rec_base != 0u) {
      helpers_on = true;
      lds_sync();
      hc_st(HC_EXT_BASE, rec_base);
      if (rec_base != 0u && lane < (uint32_t)XW_WORDS)
        lds_st32(rec_base + 4u * lane, lane == (uint32_t)XW_ORIGIN_DW ? origin_dw : lane == (uint32_t)XW_LIMIT ? ((br.end_dw - origin_dw - 8u) << 5) & ~63u :
                                       lane == (uint32_t)XW_DICT_LO ? (uint32_t)(uintptr_t)dict : lane == (uint32_t)XW_DICT_HI ? (uint32_t)((uint64_t)(uintptr_t)dict >> 32) :
                                       lane == (uint32_t)XW_RING_MASK ? ring_pos - 1u : 0u);
      if (split_on && lane < (uint32_t)CW_WORDS)
        lds_st32(sp_ctl_base() + 4u * lane, lane == (uint32_t)CW_OUT_LO ? (uint32_t)(uintptr_t)out : lane == (uint32_t)CW_OUT_HI ? (uint32_t)((uint64_t)(uintptr_t)out >> 32) : 0u);
      if (lane == 0) { LEAN_ST(L_SP_HEAD, 0u); LEAN_ST(L_SP_LIT, 0u); LEAN_ST(L_SP_ORIGIN, origin_dw); LEAN_ST(L_SP_REC, rec_base != 0u ? 1u : 0u); }
      hc_st(HC_KIND, (uint32_t)HK_SPLIT);
      lds_release();
      hc_st(HC_SEQ, hc_ld(HC_SEQ) + 1u);
    }
  }
  if (LDS_ONLY && !helpers_on && lane == 0) LEAN_ST(L_SP_REC, 0u);
  uint32_t rec_poor = 0u;   // (a metablock without context) calls of the record loop in a row that took little and ended at a command too long for the run
  auto rec_off = [&]() {   // ... and the way back to the one-wave loop and its rounds: wave 2 goes back to sleep, as at the metablock's end
    sp_st(rec_base, XW_STOP, 1u);
    for (uint32_t spins = 0; sp_ld(rec_base, XW_STOP) != 2u; spins++) {
      if (spins > SP_SPIN_CAP) { hc_st(HC_KIND, (uint32_t)HK_NO_ROUNDS); break; }
      __builtin_amdgcn_s_sleep(2);
    }
#ifdef BROTLI_AMD_REC_DEBUG
    if (blockIdx.x == 0 && lane == 0) printf("rec_off: stop word %u kind %u ncmd %llu mlen %d\n", sp_ld(rec_base, XW_STOP), hc_ld(HC_KIND), (unsigned long long)num_commands, mlen);
#endif
    rec_base = 0u; helpers_on = split_on;
    hc_st(HC_EXT_BASE, 0u);
    if (lane == 0) LEAN_ST(L_SP_REC, 0u);
    lds_sync();
  };
  uint32_t force_checked = 0;  // commands that go through the checked stages before the engine (or the lean loop) is tried again:
                               // the command the engine stopped at, more of them after invocations that got nowhere

  for (;;) {
    if (LDS_ONLY && CTX_NEVER) {
     if (scan_block && force_checked == 0u && bl1 != 0 && quota >= SC_MIN_QUOTA && !lit_zero && hc_ld(HC_KIND) != (uint32_t)HK_NO_ROUNDS) {
      const uint64_t abs_bit = br.pos() + BitReader::skip_bits();
      const uint64_t origin = abs_bit & ~63ull;
      const uint64_t avail = BitReader::total_bits() + BitReader::skip_bits() - origin;
      // the path engine where the four distance contexts share one prefix code (its states do not carry the context)
      const bool use_path = dt0 == dt1 && dt0 == dt2 && dt0 == dt3 && (g_engine_mode & 1u) == 0u && (!prefer_scan || large_window);
      if ((use_path || !large_window) && avail >= (use_path ? 2u * PE_MIN_INPUT : 8u * SC_N) && (origin >> 5) < 0xFFFFFFFFull) {
        FLUSH_LITERALS();
        FLUSH_PENDING();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in memory before the other waves read the output as copy sources
        const uint32_t sb = hc_ld(HC_SCAN_BASE);
        lds_sync();
        if (lane == 0) {
          LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32)); LEAN_ST(L_QUOTA, quota);
          LEAN_ST(L_MLEN, mlen); LEAN_ST(L_BL0, bl0); LEAN_ST(L_BL1, bl1); LEAN_ST(L_BL2, bl2);
          LEAN_ST(L_D0, d0); LEAN_ST(L_D1, d1); LEAN_ST(L_D2, d2); LEAN_ST(L_D3, d3);
        }
        sc_ctl_st(sb, SCC_BASE_DW, (uint32_t)(origin >> 5)); sc_ctl_st(sb, SCC_IN_LIMIT, avail > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)avail);
        sc_ctl_st(sb, SCC_ENTRY, (uint32_t)(abs_bit - origin));
        sc_ctl_st(sb, SCC_LIT_TREE, LDS_FIXED + lit_tree); sc_ctl_st(sb, SCC_CMD_TREE, LDS_FIXED + cmd_tree);
        sc_ctl_st(sb, SCC_DT0, LDS_FIXED + dt0); sc_ctl_st(sb, SCC_DT0 + 1, LDS_FIXED + dt1); sc_ctl_st(sb, SCC_DT0 + 2, LDS_FIXED + dt2); sc_ctl_st(sb, SCC_DT0 + 3, LDS_FIXED + dt3);
        sc_ctl_st(sb, SCC_POSTFIX, postfix_bits); sc_ctl_st(sb, SCC_NUM_DIRECT, num_direct);
        sc_ctl_st(sb, SCC_OUT_LO, (uint32_t)(uintptr_t)out); sc_ctl_st(sb, SCC_OUT_HI, (uint32_t)((uint64_t)(uintptr_t)out >> 32));
        sc_ctl_st(sb, SCC_DICT_LO, (uint32_t)(uintptr_t)dict); sc_ctl_st(sb, SCC_DICT_HI, (uint32_t)((uint64_t)(uintptr_t)dict >> 32));
        const bool use_pipe = use_path && (g_engine_mode & 8u) != 0u && !prefer_one_engine;   // (two engines of eight waves, regions in turns: BROTLI_AMD_ENGINE=path2 -- measured slower than one of sixteen, see DESIGN)
        const bool use_general = use_path && !use_pipe && prefer_general;
        // (the stream's owner in a gang of blocks: the gang's form of the engine -- not for a literal run that wants regions of its own, nor for
        // words of the static dictionary: those are the one-block forms')
#ifndef BROTLI_AMD_GANG_KERNEL   // (the kernel of the launches without gangs: see the end of this file)
        const bool use_remote = false;
#else
        // (the counts looked at here, whether the helpers have all started by the engine itself the first time: a call in this place costs this
        // function three hundred bytes of frame, and the spills around it every invocation)
        bool use_remote = false;
        if (use_path && !use_pipe && !use_general && hc_ld(HC_GANG_M) > 1u) {
          const uint32_t counts = rfl(args->general_engine);
          if (((counts >> 16) & 0xFFu) != 0u) args->general_engine = counts - (1u << 16);
          else if (!prefer_one_engine) use_remote = true;
        }
#endif
        hc_st(HC_KIND, use_remote ? (uint32_t)HK_PATHR : use_pipe ? (uint32_t)HK_PATH2 : use_general ? (uint32_t)HK_PATHG : use_path ? (uint32_t)HK_PATH : (uint32_t)HK_SCAN);
        lds_release();
        hc_st(HC_SEQ, hc_ld(HC_SEQ) + 1u);  // the other waves of the block join (helper_wave)
#ifdef BROTLI_AMD_PROFILE_SCAN
        { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && lane == 0 && pp_exit != 0) { g_path_prof[36] += t_ - pp_exit; g_path_prof[37] += 1; } if (blockIdx.x == 0 && lane == 0 && pp_exit == 0) g_path_prof[29] += t_ - pp_start; pp_enter = t_; }
#endif
#ifdef BROTLI_AMD_PE_DEBUG
        if (blockIdx.x == 0 && lane == 0) printf("engine in: P %llu bl1 %u quota %u mlen %d commands so far %llu\n", (unsigned long long)P, bl1, quota, mlen, (unsigned long long)num_commands);
#endif
#ifndef BROTLI_AMD_GANG_KERNEL
        const uint32_t took = use_pipe ? rfl(pe8::path_engine(0)) : use_general ? rfl(pe16g::path_engine(0)) : use_path ? rfl(pe16::path_engine(0)) : rfl(scan_engine(0));
#else
        const uint32_t took = use_remote ? rfl(pe16r::path_engine(0)) : use_pipe ? rfl(pe8::path_engine(0)) : use_general ? rfl(pe16g::path_engine(0)) : use_path ? rfl(pe16::path_engine(0)) : rfl(scan_engine(0));
#endif
#ifdef BROTLI_AMD_PE_DEBUG
        if (blockIdx.x == 0 && lane == 0) printf("engine out: tick %llu took %u, P %llu form %u\n", (unsigned long long)__builtin_amdgcn_s_memtime(), took, (unsigned long long)(LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32)), LEAN_LD(L_SC_POS_HI));
#endif
#ifdef BROTLI_AMD_PROFILE_SCAN
        pp_exit = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && lane == 0) g_path_prof[38] += pp_exit - pp_enter;
#endif
        prefer_one_engine = false;
        engine_commands += took;
        {  // the literal rounds' mailbox words lie in the engine's rings: back to their idle state
          const uint32_t hb_ = hc_ld(HC_BASE);
          for (uint32_t t = lane; t < SC_WAVES * 16u; t += 64u) lds_st32(hb_ + (t >> 4) * HL_SLOT + HL_CTL + 4u * (t & 15u), 0u);
        }
        const uint32_t form_raw = LEAN_LD(L_SC_POS_HI), form = form_raw & 0xFFu;
        const bool declined = ((form_raw >> 8) & 1u) != 0u;
        if (((form_raw >> 10) & 1u) != 0u) prefer_general = true;   // (the lean form stopped in front of a dictionary reference)
#ifdef BROTLI_AMD_GANG_KERNEL
        if (use_remote) gang_took(args, took, form_raw);
#endif
        if (((form_raw >> 9) & 1u) != 0u) prefer_scan = true;   // (the path engine's regions were bound by their closure: a stream of few literals -- the scan engine's from here on)   // (the two engines stopped in front of a literal run that wants regions of its own: the one-engine form's, at once)
        const uint64_t pos = origin + LEAN_LD(L_SC_POS_LO) - BitReader::skip_bits();
        if (lane == 0) { LEAN_ST(L_SPEC_LO, (uint32_t)rfl(args->spec_scratch)); LEAN_ST(L_SPEC_HI, (uint32_t)(rfl(args->spec_scratch) >> 32)); }
        br.seek(pos);
#ifdef BROTLI_AMD_PE_DEBUG
        if (blockIdx.x == 0 && lane == 0) printf("  after seek: tick %llu\n", (unsigned long long)__builtin_amdgcn_s_memtime());
#endif
        const uint64_t P_before = P;
        P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
        quota = LEAN_LD(L_QUOTA); mlen = (int32_t)LEAN_LD(L_MLEN);
        bl0 = LEAN_LD(L_BL0); bl1 = LEAN_LD(L_BL1); bl2 = LEAN_LD(L_BL2);
        d0 = (int32_t)LEAN_LD(L_D0); d1 = (int32_t)LEAN_LD(L_D1); d2 = (int32_t)LEAN_LD(L_D2); d3 = (int32_t)LEAN_LD(L_D3);
        num_commands += took;
        lit_pos = P;
        force_checked = form == SCX_BEGIN ? 1u : 0u;  // (the command the engine stopped IN FRONT OF goes through the checked stages; one it stopped inside is on its way through them already)
        // (an invocation that got nowhere -- few commands AND few bytes: a long literal run is one command -- makes the next ones rarer)
        if (declined) { prefer_one_engine = true; force_checked = 0u; }
        else if (took < 64u && P - P_before < 4096u) { scan_fails = scan_fails < 6u ? scan_fails + 1u : 6u; force_checked = 8u << scan_fails; } else scan_fails = 0;
        insert_len = (int32_t)LEAN_LD(L_INSERT); copy_len = (int32_t)LEAN_LD(L_COPY);
        distance_code = (int32_t)LEAN_LD(L_DCODE); distance_context = LEAN_LD(L_DCTX); lits_left = (int32_t)LEAN_LD(L_LITS_LEFT);
        lds_sync();
        if (form == SCX_LITERALS_REST) { if (lits_left != 0) goto general_literals_rest; goto general_distance; }
        if (form == SCX_POST_DISTANCE) goto general_post_distance;
        if (declined && use_remote) continue;   // (a gang stopped in front of a literal run that wants regions of its own: the one-block form's, at once -- not the one-wave loop's)
      }
     }
    }
    if (LDS_ONLY && bl1 != 0 && quota != 0 && !(CTX_NEVER && lit_zero) && br.next_dw < safe_dw && force_checked == 0u) {
      // ---- the lean loop takes over until a stage needs the checked code below ----
      if (!CTX_NEVER && ctx_src == CTX_PEND) {  // the context bytes leave the pending copy before it is stored
        uint32_t q1 = rdlane(pend_reg, ctx_len - 1);
        p2 = ctx_len >= 2 ? rdlane(pend_reg, ctx_len - 2) : p1;
        p1 = q1;
        ctx_src = CTX_REGS;
      }
      FLUSH_LITERALS();
      FLUSH_PENDING();
      lds_sync();
      lds_st32(LDS_LEANWIN + 4u * lane, br.cur);
      if (lane == 0) {
        LEAN_ST(L_CHUNK_BASE, br.chunk_base);
        LEAN_ST(L_BUF_LO, (uint32_t)br.buf); LEAN_ST(L_BUF_HI, (uint32_t)(br.buf >> 32)); LEAN_ST(L_CNT, br.cnt); LEAN_ST(L_NEXT_DW, br.next_dw);
        LEAN_ST(L_ISSUED, br.issued_half); LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32)); LEAN_ST(L_QUOTA, quota);
        LEAN_ST(L_MLEN, mlen); LEAN_ST(L_BL0, bl0); LEAN_ST(L_BL1, bl1); LEAN_ST(L_BL2, bl2);
        LEAN_ST(L_D0, d0); LEAN_ST(L_D1, d1); LEAN_ST(L_D2, d2); LEAN_ST(L_D3, d3);
        LEAN_ST(L_CMD_TREE, cmd_tree); LEAN_ST(L_LIT_TREE, lit_tree);
        LEAN_ST(L_DT0, dt0); LEAN_ST(L_DT1, dt1); LEAN_ST(L_DT2, dt2); LEAN_ST(L_DT3, dt3);
        if (!CTX_NEVER) {
          LEAN_ST(L_P1, p1); LEAN_ST(L_P2, p2); LEAN_ST(L_CTX_REGS, ctx_src == CTX_REGS ? 1u : 0u); LEAN_ST(L_TRIVIAL, trivial);
          LEAN_ST(L_CTX_LUT, ctx_lut);
        } else if (rec_base != 0u) {   // (the record loop's words for a metablock without context: one tree, the context bytes whatever they are)
          LEAN_ST(L_P1, 0u); LEAN_ST(L_P2, 0u); LEAN_ST(L_CTX_REGS, 1u); LEAN_ST(L_TRIVIAL, 1u); LEAN_ST(L_CTX_LUT, (uint32_t)LDS_CTX_LUT);
        }
      }
      lds_sync();
      uint32_t stage;
      if (!CTX_NEVER && split_on) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // what this wave has stored is in memory before the copier reads it as a copy's source
        sp_st(sp_ctl_base(), CW_DONE_P, (uint32_t)P);       // (the copier is idle: it says so again only after the next record)
        stage = rfl(lean_split_commands(lut_vgpr, ctx_tree_v));
        if (hc_ld(HC_FAILED) != 0u) STOP(E_UNREACHABLE);    // the copier did not answer (never seen)
        engine_commands += LEAN_LD(L_NCMD_LO);
      } else {
        stage = rec_base != 0u ? rfl(lean_rec_commands(ctx_tree_v)) : rfl(lean_commands<CTX_NEVER>(lut_vgpr, ctx_tree_v));
        if (rec_base != 0u) {
          const uint32_t took_ = LEAN_LD(L_NCMD_LO);
          engine_commands += took_;
          // (a metablock without context has another road: what a call of the record loop takes, halved and added up -- a stream of long copies or long
          // literal runs leaves the run at every command, and every way out costs what forty of its commands do)
#ifdef BROTLI_AMD_REC_DEBUG
          if (CTX_NEVER && blockIdx.x == 0 && lane == 0) printf("rec call: took %u waited %u poor %u stage %u ncmd %llu\n", took_, LEAN_LD(L_SP_WAITED), rec_poor, stage, (unsigned long long)num_commands);
#endif
          // (a metablock without context has another road.  A stream of long copies or long literal runs leaves the run at every such command, and every way
          // out costs what forty of the run's commands do: six calls in a row that ended at a LONG command having taken less than sixty-four, and the
          // one-wave loop and its rounds have the rest of the metablock.  The first commands of a text -- words of the dictionary with their transforms while
          // the window is empty -- end calls too, but not like that.)
          if (CTX_NEVER) { if (took_ >= 64u) rec_poor = 0u; else if (LEAN_LD(L_SP_WAITED) == 2u && ++rec_poor >= 6u) rec_off(); }
        }
      }
      br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
      br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED);
      br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);
      P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
      quota = LEAN_LD(L_QUOTA); mlen = (int32_t)LEAN_LD(L_MLEN);
      bl0 = LEAN_LD(L_BL0); bl1 = LEAN_LD(L_BL1); bl2 = LEAN_LD(L_BL2);
      d0 = (int32_t)LEAN_LD(L_D0); d1 = (int32_t)LEAN_LD(L_D1); d2 = (int32_t)LEAN_LD(L_D2); d3 = (int32_t)LEAN_LD(L_D3);
      num_commands += LEAN_LD(L_NCMD_LO);
      if (!CTX_NEVER) {
        if (LEAN_LD(L_CTX_REGS)) { p1 = LEAN_LD(L_P1); p2 = LEAN_LD(L_P2); ctx_src = CTX_REGS; }
        else ctx_src = CTX_MEMORY;
      }
#ifdef BROTLI_AMD_PROFILE
      prof_fast_batches++; prof_fast_syms += LEAN_LD(L_NCMD_LO);  // (lean entries, commands run lean)
      prof_stage[stage & 7]++;
#endif
      insert_len = (int32_t)LEAN_LD(L_INSERT); copy_len = (int32_t)LEAN_LD(L_COPY);
      distance_code = (int32_t)LEAN_LD(L_DCODE); distance_context = LEAN_LD(L_DCTX); lits_left = (int32_t)LEAN_LD(L_LITS_LEFT);
      if (CTX_NEVER && stage == LS_LITERAL_ROUNDS) {
        spec_rounds(LDS_FIXED + lit_tree);
        if (hc_ld(HC_FAILED) != 0u) STOP(E_UNREACHABLE);  // a helper wave did not finish moving its literals (never seen)
        br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
        br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED);
        br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);
        P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
        quota = LEAN_LD(L_QUOTA); bl0 = LEAN_LD(L_BL0); lits_left = (int32_t)LEAN_LD(L_LITS_LEFT);
        // what is left of the run (fewer than a round pays for, or close to the end of the input) is decoded below;
        // a run that is complete goes on like one the lean loop completed
        stage = lits_left != 0 ? LS_LITERALS_REST : quota == 0 ? LS_LITERALS_AT_LIMIT : LS_DISTANCE;
      }
      if (stage == LS_AFTER_HEAD) goto after_head;
      if (stage == LS_LITERALS_REST) goto general_literals_rest;
      if (stage == LS_LITERALS_AT_LIMIT) {  // exactly at a limit: end of the metablock, flush point, or full output buffer
        if (P >= next_boundary) RING_CROSS();
        if (mlen <= 0) STOP(E_SUCCESS);  // METABLOCK_DONE, copy part ignored (decode.rs:2552-2556)
        RECOMPUTE_QUOTA();
        goto general_distance;
      }
      if (stage == LS_DISTANCE) goto general_distance;
      if (stage == LS_POST_DISTANCE) goto general_post_distance;
      if (stage == LS_COMMAND_DONE) goto command_done;
      if (stage == LS_NEEDS_INPUT) STOP(E_NEEDS_MORE_INPUT);
    }
    // ---- COMMAND_BEGIN ----
    if (br.next_dw >= safe_dw) {
      // Close to the end of the input: should it run out inside this metablock, the next launch goes on from the last
      // boundary noted here instead of from the metablock's first command (BrotliAmdResume: mid_*).  Any boundary of
      // this metablock would do; store_resume() forgets them at the next metablock boundary.
      typedef __attribute__((address_space(1))) BrotliAmdResume gresume;
      gresume* const r = (gresume*)(uintptr_t)rfl(args->resume_out);
      const uint64_t bit = br.pos();
      if (lane < 6u) r->mid_types[lane] = lds_ld32(LDS_HOT + 4u * (H_RING + lane));
      if (lane == 0) {
        r->mid_mlen = mlen; r->mid_bit_pos = bit; r->mid_out_pos = P;
        r->mid_bl[0] = bl0; r->mid_bl[1] = bl1; r->mid_bl[2] = bl2;
        r->mid_dist_rb[0] = d0; r->mid_dist_rb[1] = d1; r->mid_dist_rb[2] = d2; r->mid_dist_rb[3] = d3;
        r->mid_valid = 1u;
      }
    }
    if (force_checked != 0u) force_checked--;
    if (bl1 == 0) {
      int r;
      BLOCK_SWITCH(1, bl1, r);
      if (r == BS_NEEDS_INPUT) STOP(E_NEEDS_MORE_INPUT);
      if (r == BS_SWITCHED) { cmd_tree = cached_tree(a.ld32<false>(HOTC(H_CMD_TREES) + HOTC(H_RING + 3) * 4), TREE_CACHE_CMD, TREE_CACHE_CMD_BYTES); records_stale(); continue; }
    }
    {
      uint32_t cmd = read_symbol<LDS_ONLY>(br, a, cmd_tree);
      // kCmdLut regenerated arithmetically (RFC 7932 section 5; replaces src/prefix.rs:115-5755); 2-byte tree entries:
      // 8-byte entries carrying these fields were measured 1 % faster but cost 3.3 KB of LDS per command tree
      uint32_t cell = cmd >> 6;
      uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u);   // {0,0,0,0,1,1,0,2,1,2,2}
      uint32_t copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);         // {0,1,0,1,0,1,2,0,2,1,2}
      uint32_t ie = rdlane(lut_vgpr, ins_code), ce = rdlane(lut_vgpr, 32u + copy_code);
      distance_code = cmd < 128 ? 0 : -1;
      distance_context = copy_code > 2 ? 3u : copy_code;
      insert_len = (int32_t)((ie & 0xFFFFu) + br.read(ie >> 16));
      copy_len = (int32_t)((ce & 0xFFFFu) + br.read(ce >> 16));
    }
    if (br.over()) STOP(E_NEEDS_MORE_INPUT);
    bl1--;
    num_commands++;
    PROF_ADD(prof_cmd, prof_t);
    lits_left = insert_len;  // literals of this command that are still to be decoded
after_head:
    // p1/p2 must be right whenever a literal's context can matter: not at all in a metablock whose literal block
    // types are all trivial, otherwise always (a block switch inside the run may make the very next literal
    // context-modelled)
    if (!ctx_never && insert_len != 0) {
      if (ctx_src != CTX_REGS) {
        if (ctx_src == CTX_PEND) {
          uint32_t q1 = rdlane(pend_reg, ctx_len - 1);
          p2 = ctx_len >= 2 ? rdlane(pend_reg, ctx_len - 2) : p1;
          p1 = q1;
        } else {
          FLUSH_PENDING();
          p1 = P >= 1 ? (uint32_t)rfl(out[P - 1]) : 0u;
          p2 = P >= 2 ? (uint32_t)rfl(out[P - 2]) : 0u;
        }
      }
      ctx_src = CTX_REGS;
    }

    if (insert_len != 0 && quota_mb && !CTX_NEVER && (uint32_t)insert_len <= quota && (uint32_t)insert_len <= bl0) {
      // ---- context-modelled literals of a command that stays clear of every output-side limit and of the end of
      // its literal block: one at a time (the tree depends on the two bytes before), nothing to check but the input
      mlen -= insert_len;
      uint32_t i = (uint32_t)insert_len;
      if (lit_n == 0) lit_pos = P;
      while (i > 0 && br.next_dw < safe_dw) {
        uint32_t tree = lit_tree;
        if (!trivial) {
          uint32_t context = rfl(lds_ld8(ctx_lut + p1) | lds_ld8(ctx_lut + 256 + p2));
          tree = rdlane(ctx_tree_v, context);
        }
        uint32_t lit = read_symbol<true>(br, a, tree);
        p2 = p1; p1 = lit;
        lit_reg = (lane == lit_n) ? lit : lit_reg;
        lit_n++;
        if (lit_n == 64) FLUSH_LITERALS();
        i--;
      }
      const uint32_t done = (uint32_t)insert_len - i;
      P += done; bl0 -= done; quota -= done;
      lits_left = (int32_t)i;
      if (quota == 0 && i == 0) {
        if (P >= next_boundary) RING_CROSS();
        if (mlen <= 0) STOP(E_SUCCESS);  // METABLOCK_DONE, copy part ignored (decode.rs:2552-2556)
        RECOMPUTE_QUOTA();
      }
    } else if (insert_len != 0) {
      mlen -= insert_len;
    }
general_literals_rest:
    if (lits_left != 0) {
      // ---- COMMAND_INNER: literals, every limit checked ----
      int32_t i = lits_left;
      // ---- wave-parallel literal decode (trivial context: one prefix code for the whole run) ----
      // Every lane decodes the symbol that would start at bit offset `lane` of a 64-bit window (one gathered table
      // lookup for all 64 candidates); a short scalar walk over the code lengths then picks the offsets that really
      // are symbol boundaries, and the surviving lanes store their bytes with one coalesced instruction.
#ifdef BROTLI_AMD_PROFILE
      if (prof_fast_syms == 0) prof_fast_syms = 0x100000u | (LDS_ONLY ? 1u : 0u) | (trivial ? 2u : 0u) | (mlen >= 0 ? 4u : 0u) | (i >= 8 ? 8u : 0u) | (bl0 >= 8 ? 16u : 0u) | (out_cap - P >= 8 ? 32u : 0u);
#endif
      PROF_LIT(prof_lit, prof_t);
rounds_again:
      // A long run that the lean loop could not take whole (it crosses a flush point of the ring buffer, the end of a
      // literal block or of the output buffer): rounds of the helper waves for the part in front of that limit, the
      // checked loops below for what is left of that part, and again behind the limit.
      if (CTX_NEVER && LDS_ONLY && trivial && mlen >= 0 && i >= (int32_t)SPEC_ROUND_MIN && !lit_zero &&
          br.next_dw + spec_input_dwords(SPEC_MAX_WAVES) < safe_dw && hc_ld(HC_KIND) != 3u) {
        uint64_t lim = out_cap - P;
        const uint64_t rb_ = next_boundary > P ? next_boundary - P : 0;
        if (rb_ < lim) lim = rb_;
        uint32_t part = (uint32_t)i < bl0 ? (uint32_t)i : bl0;
        if (lim < (uint64_t)part) part = (uint32_t)lim;
        if (part >= SPEC_ROUND_MIN) {
          if (rec_base != 0u) rec_off();   // (a literal run for the helper waves' rounds: wave 2 is one of them -- and a stream of such runs is not the record loop's)
          FLUSH_LITERALS();
          FLUSH_PENDING();
          lds_sync();
          lds_st32(LDS_LEANWIN + 4u * lane, br.cur);
          if (lane == 0) {
            LEAN_ST(L_CHUNK_BASE, br.chunk_base);
            LEAN_ST(L_BUF_LO, (uint32_t)br.buf); LEAN_ST(L_BUF_HI, (uint32_t)(br.buf >> 32)); LEAN_ST(L_CNT, br.cnt); LEAN_ST(L_NEXT_DW, br.next_dw);
            LEAN_ST(L_ISSUED, br.issued_half); LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32));
            LEAN_ST(L_LITS_LEFT, part); LEAN_ST(L_BL0, bl0); LEAN_ST(L_QUOTA, part);
          }
          lds_sync();
          spec_rounds(LDS_FIXED + lit_tree);
          if (hc_ld(HC_FAILED) != 0u) STOP(E_UNREACHABLE);
          br.buf = (uint64_t)LEAN_LD(L_BUF_LO) | ((uint64_t)LEAN_LD(L_BUF_HI) << 32);
          br.cnt = LEAN_LD(L_CNT); br.next_dw = LEAN_LD(L_NEXT_DW); br.issued_half = LEAN_LD(L_ISSUED);
          br.chunk_base = LEAN_LD(L_CHUNK_BASE); br.cur = lds_ld32(LDS_LEANWIN + 4u * lane);
          P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
          const uint32_t got = part - LEAN_LD(L_LITS_LEFT);
          i -= (int32_t)got; bl0 -= got;
          ctx_src = CTX_MEMORY;
          if (P >= next_boundary) RING_CROSS();
          if (got != 0 && part == got) goto rounds_again;  // the part is done: behind the limit there may be another
        }
      }
      if (LDS_ONLY && trivial && mlen >= 0) {
        while (i >= 8 && bl0 >= 8) {
          uint64_t room = out_cap - P;
          uint32_t lim = (uint32_t)i < bl0 ? (uint32_t)i : bl0;
          if (lim > 64) lim = 64;
          if ((uint64_t)lim > room) lim = (uint32_t)room;
          if (lim < 8) break;
          br.need32();
          uint64_t wlo, whi;
          br.window128(wlo, whi);
          // bits that really exist from here on (pos <= total here), saturated to 32 bits: the walk below compares
          // in 32 bits (hipcc 7.2 miscompiles a select fed by a uniform 64-bit unsigned compare in this loop)
          uint64_t avail64 = br.total_bits() - br.pos();
          uint32_t avail = (avail64 >> 16) ? 0xFFFFu : (uint32_t)avail64;
          uint32_t w0 = (uint32_t)wlo, w1 = (uint32_t)(wlo >> 32), w2 = (uint32_t)whi;
          uint32_t x = lane < 32 ? __builtin_amdgcn_alignbit(w1, w0, lane) : __builtin_amdgcn_alignbit(w2, w1, lane - 32);
          uint32_t e = lds_ld16(LDS_FIXED + lit_tree + ((x & 0xFFu) << 1));
          uint32_t L = e & 15u;
          if (L > ROOT_BITS) {
            uint32_t idx = (e >> 4) + ((x >> ROOT_BITS) & mask_bits(L - ROOT_BITS));
            e = lds_ld16(LDS_FIXED + lit_tree + (idx << 1));
            L = ROOT_BITS + (e & 15u);
          }
          uint32_t sym = e >> 4;
          if (rdlane(L, 0) == 0) break;  // single-symbol code (zero bits per literal): the scalar loop handles it
          // Candidates that would run past the end of the input end the walk: give them a length that jumps out of
          // the window, and cut the result at the first such start afterwards.
          uint64_t vmask = __ballot(lane + L <= avail);
          uint32_t Lw = ((vmask >> lane) & 1ull) ? L : 64u;
          uint64_t starts; uint32_t off, tmp;
          // off = 0; do { starts |= 1 << off; off += Lw[off]; } while (off < 64);   -- 5 instructions per symbol
          asm volatile("s_mov_b64 %0, 0\n\ts_mov_b32 %1, 0\n"
                       "1:\n\ts_nop 3\n\tv_readlane_b32 %2, %3, %1\n\ts_bitset1_b64 %0, %1\n\ts_add_u32 %1, %1, %2\n\ts_cmp_lt_u32 %1, 64\n\ts_cbranch_scc1 1b\n"
                       : "=&s"(starts), "=&s"(off), "=&s"(tmp) : "v"(Lw) : "scc");
          uint64_t bad = starts & ~vmask;
          if (bad) { off = (uint32_t)__builtin_ctzll(bad); starts &= (1ull << off) - 1ull; }
          // no more symbols than the run (and the block) has left
          for (uint32_t cnt_ = (uint32_t)__popcll(starts); cnt_ > lim; cnt_--) {
            off = 63u - (uint32_t)__clzll((long long)starts);
            starts &= ~(1ull << off);
          }
          if (starts == 0) break;
          uint32_t n = (uint32_t)__popcll(starts);
          FLUSH_LITERALS();
          if ((starts >> lane) & 1ull) out[P + (uint32_t)__popcll(starts & ((1ull << lane) - 1ull))] = (uint8_t)sym;
          uint32_t last = 63u - (uint32_t)__clzll((long long)starts);
          uint64_t rest = starts & ~(1ull << last);
          uint32_t q1 = rdlane(sym, last);
          p2 = rest ? rdlane(sym, 63u - (uint32_t)__clzll((long long)rest)) : p1;
          p1 = q1;
          br.advance(off);
          P += n; i -= (int32_t)n; bl0 -= n;
#ifdef BROTLI_AMD_PROFILE
          prof_fast_batches++; prof_fast_syms += n;
#endif
          if (P >= next_boundary) { RING_CROSS(); if (i >= (int32_t)SPEC_ROUND_MIN) goto rounds_again; }
        }
      }
      PROF_LIT(prof_dist, prof_t);
      while (i > 0) {
        if (lit_n == 0) lit_pos = P;
        if (bl0 == 0) {
          int r;
          BLOCK_SWITCH(0, bl0, r);
          if (r == BS_NEEDS_INPUT) STOP(mlen < 0 ? E_BLOCK_LENGTH_1 : E_NEEDS_MORE_INPUT);
          if (r == BS_SWITCHED) prepare_literal();
          if (i >= (int32_t)SPEC_ROUND_MIN) goto rounds_again;
        }
        uint32_t tree = lit_tree;
        if (!CTX_NEVER && !trivial) {
          uint32_t context = rfl(lds_ld8(ctx_lut + p1) | lds_ld8(ctx_lut + 256 + p2));
          tree = rdlane(ctx_tree_v, context);
        }
        uint32_t lit = read_symbol<LDS_ONLY>(br, a, tree);
        if (br.over()) STOP(mlen < 0 ? E_BLOCK_LENGTH_1 : E_NEEDS_MORE_INPUT);  // decode.rs:2835-2846 + 1709-1711
        // a full output buffer is only an error while the metablock is still within its declared length; past it
        // the stream is already invalid and the remaining literals are decoded without being stored
        if (P >= out_cap && mlen >= 0) STOP(E_NEEDS_MORE_OUTPUT);
        p2 = p1; p1 = lit;
        if (P < out_cap) {  // (past the end only while the stream is already invalid: decoded, not stored)
          lit_reg = (lane == lit_n) ? lit : lit_reg;
          lit_n++;
          if (lit_n == 64) FLUSH_LITERALS();
        }
        P++;
        if (bl0 == 0) STOP(E_WINDOW_BITS);  // decode.rs:2434-2439
        bl0--;
        i--;
        if (P >= next_boundary) { RING_CROSS(); if (i >= (int32_t)SPEC_ROUND_MIN) goto rounds_again; }
      }
      PROF_LIT(prof_copy, prof_t);
      if (mlen <= 0) STOP(E_SUCCESS);  // METABLOCK_DONE, copy part ignored (decode.rs:2552-2556)
      if (quota_mb) RECOMPUTE_QUOTA();
    }
    PROF_ADD(prof_lit, prof_t);
    // ---- COMMAND_POST_DECODE_LITERALS ----
general_distance:
    if (distance_code >= 0) {
      distance_context = 1;  // implicit distance: the last one, not pushed again (decode.rs:2560-2565 + 2643-2644)
      distance_code = d0;
    } else {
      if (bl2 == 0) {
        int r;
        BLOCK_SWITCH(2, bl2, r);
        if (r == BS_NEEDS_INPUT) STOP(E_NEEDS_MORE_INPUT);
        if (r == BS_SWITCHED) prepare_distance();
      }
      uint32_t dtree = distance_context == 0 ? dt0 : distance_context == 1 ? dt1 : distance_context == 2 ? dt2 : dt3;
      // ReadDistanceInternal, decode.rs:2066-2131
      uint32_t code = read_symbol<LDS_ONLY>(br, a, dtree);
      distance_context = 0;
      if (code < 16) {
        if (br.over()) STOP(E_NEEDS_MORE_INPUT);
        // TakeDistanceFromRingBuffer, decode.rs:2017-2049
        if (code == 0) {
          distance_code = d0;
          distance_context = 1;
        } else {
          uint32_t sh = code << 1;
          uint32_t back = 3u - ((0xaaafff1bu >> sh) & 3u);  // 0 = last distance ... 3 = fourth last
          int32_t v = back == 0 ? d0 : back == 1 ? d1 : back == 2 ? d2 : d3;
          int32_t mag = (int32_t)((0xfa5fa500u >> sh) & 3u);
          if (code & 1u) v += mag;
          else { v -= mag; if (v <= 0) v = 0x7fffffff; }
          distance_code = v;
        }
      } else {
        int32_t distval = (int32_t)code - (int32_t)num_direct;
        int32_t dc = (int32_t)code;
        if (distval >= 0) {
          int32_t postfix = distval & (int32_t)mask_bits(postfix_bits);
          distval >>= postfix_bits;
          uint32_t nbits = ((uint32_t)distval >> 1) + 1;
          uint32_t bits = br.read(nbits);
          int64_t offset = (int64_t)(int32_t)((((uint32_t)(distval & 1) + 2u) << nbits) - 4u);
          dc = (int32_t)(((offset + (int64_t)bits) << postfix_bits) + postfix + (int64_t)num_direct);
        }
        if (br.over()) STOP(E_NEEDS_MORE_INPUT);
        distance_code = (int32_t)((uint32_t)dc - 16u + 1u);
      }
      bl2--;
    }
    PROF_REST(prof_dist, prof_t);
    // postReadDistance, decode.rs:2583-2589
general_post_distance:
    max_distance = (P < (uint64_t)(uint32_t)max_backward) ? (int32_t)P : max_backward;
    if (distance_code > max_distance) {
      if (distance_code > 0x7FFFFFFC) STOP(E_DISTANCE);
      if (copy_len < 4 || copy_len > 24) STOP(E_DICTIONARY);
      uint32_t shift = kDictSizeBitsByLength[copy_len];
      int32_t word_id = distance_code - max_distance - 1;
      uint32_t word_idx = (uint32_t)word_id & mask_bits(shift);
      uint32_t transform_idx = (uint32_t)word_id >> shift;
      if (transform_idx >= BROTLI_NUM_TRANSFORMS) STOP(E_TRANSFORM);
      uint32_t offset = kDictOffsetsByLength[copy_len] + word_idx * (uint32_t)copy_len;
      WordShape w = word_shape((uint32_t)copy_len, transform_idx);
      mlen -= (int32_t)w.total;
      if (mlen < 0) STOP(P + w.total >= next_boundary ? E_BLOCK_LENGTH_1 : E_BLOCK_LENGTH_2);  // decode.rs:2621-2625, 3356-3359
      if (w.total != 0) {
        if (P + w.total > out_cap) {  // clip: deliver what fits, then report the full buffer
          FLUSH_LITERALS(); FLUSH_PENDING();
          uint32_t ob = dictionary_word_bytes(dict, offset, w);
          uint64_t q = P + lane;
          if (lane < w.total && q < out_cap) out[q] = (uint8_t)ob;
          P = out_cap;
          STOP(E_NEEDS_MORE_OUTPUT);
        }
        FLUSH_LITERALS();
        if (ctx_src == CTX_PEND && w.total == 1) { p1 = rdlane(pend_reg, ctx_len - 1); ctx_src = CTX_REGS; }
        FLUSH_PENDING();
        uint32_t ob = dictionary_word_bytes(dict, offset, w);
        if (w.total == 1) {  // context = this byte and the one before it
          if (ctx_src == CTX_MEMORY) { p1 = P >= 1 ? (uint32_t)rfl(out[P - 1]) : 0u; }
          p2 = p1; p1 = rdlane(ob, 0); ctx_src = CTX_REGS;
        } else { ctx_src = CTX_PEND; ctx_len = w.total; }
        pend_reg = ob; pend_n = w.total; pend_pos = P;
        P += w.total;
      }
    } else {
      if (distance_context == 0) { d3 = d2; d2 = d1; d1 = d0; d0 = distance_code; }
      mlen -= copy_len;
      // a copy that overshoots MLEN ends the stream whatever it copies (decode.rs:2690-2720 + 1709-1711 / 3356-3359)
      if (mlen < 0) STOP(P + (uint32_t)copy_len >= next_boundary ? E_BLOCK_LENGTH_1 : E_BLOCK_LENGTH_2);
      if (distance_code <= 0) STOP(E_UNREACHABLE);  // wrapped large-window arithmetic, never on valid streams
      const uint32_t dist = (uint32_t)distance_code;
      if (quota_mb && !CTX_NEVER && (uint32_t)copy_len <= 64u && (uint32_t)copy_len <= quota && dist >= (uint32_t)copy_len) {
        // lean short copy where literal context matters: one byte per lane, so that the next literal can take the two
        // bytes before it straight from the register (copy lengths start at 2)
        FLUSH_LITERALS();
        FLUSH_PENDING();
        const uint32_t n = (uint32_t)copy_len;
        uint32_t b = 0;
        if (lane < n) b = (out + P - dist)[lane];
        pend_reg = b; pend_n = n; pend_pos = P;
        ctx_src = CTX_PEND; ctx_len = n;
        P += n;
        quota -= n;
        PROF_REST(prof_copy, prof_t);
        if (quota != 0) continue;
        goto command_done;
      }
      if (quota_mb && (uint32_t)copy_len <= quota && dist >= (uint32_t)copy_len && (uint32_t)copy_len <= 1024u &&
          (CTX_NEVER || (uint32_t)copy_len > 64u)) {
        // lean copy: fits, does not overlap itself; 16 bytes per lane plus a byte tail, stored when the next
        // command gets here (its source may be what this one writes)
        FLUSH_LITERALS();
        FLUSH_PENDING();
        const uint32_t n = (uint32_t)copy_len;
        gu8* src = out + P - dist;
        uint32_t n16 = n >> 4, rem = n & 15u;
        u32x4 v = {0, 0, 0, 0};
        if (lane < n16) v = *reinterpret_cast<gu32x4*>(src + (uint64_t)lane * 16);
        uint32_t b = 0;
        if (lane < rem) b = src[(n16 << 4) + lane];
        pendv = v; pendv_n16 = n16; pendv_pos = P;
        pend_reg = b; pend_n = rem; pend_pos = P + (n16 << 4);
        if (!CTX_NEVER) ctx_src = CTX_MEMORY;
        P += n;
        quota -= n;
        PROF_REST(prof_copy, prof_t);
        if (quota != 0) continue;
        goto command_done;
      }
      uint64_t room = out_cap - P;
      uint32_t n = (uint32_t)copy_len;
      const bool clipped = (uint64_t)n > room;
      if (clipped) n = (uint32_t)room;
      // every earlier byte must be in memory (or at least ordered before the loads below in the wave's in-order
      // vector memory pipeline)
      FLUSH_LITERALS();
      FLUSH_PENDING();
      gu8* dst = out + P;
      if (n <= 64 && dist >= n) {
        // short, non-overlapping: load now, store when the next command comes here (or at exit)
        uint32_t b = 0;
        if (lane < n) b = (dst - dist)[lane];
        pend_reg = b; pend_n = n; pend_pos = P;
        if (n >= 2) { ctx_src = CTX_PEND; ctx_len = n; }
        else if (n == 1) { if (ctx_src == CTX_MEMORY) p1 = P >= 1 ? (uint32_t)rfl(out[P - 1]) : 0u; p2 = p1; p1 = rdlane(b, 0); ctx_src = CTX_REGS; }
      } else if (dist >= n && n <= 1024) {
        // up to 1 KiB, not overlapping itself: same split as above with 16 bytes per lane (+ a byte tail)
        gu8* src = dst - dist;
        uint32_t n16 = n >> 4, rem = n & 15u;
        u32x4 v = {0, 0, 0, 0};
        if (lane < n16) v = *reinterpret_cast<gu32x4*>(src + (uint64_t)lane * 16);
        uint32_t b = 0;
        if (lane < rem) b = src[(n16 << 4) + lane];
        pendv = v; pendv_n16 = n16; pendv_pos = P;
        pend_reg = b; pend_n = rem; pend_pos = P + (n16 << 4);
        ctx_src = CTX_MEMORY;
      } else if (dist >= n || dist >= 1024) {
        // long: 16 bytes per lane and step; steps are >= 1 KiB apart from their source or do not overlap at all
        gu8* src = dst - dist;
        uint32_t n16 = n >> 4;
        for (uint32_t c = lane; c < n16; c += 64) {
          u32x4 v = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16);  // any alignment: global accesses are byte-addressed
          *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = v;
        }
        uint32_t tail = n16 << 4;
        if (tail + lane < n) dst[tail + lane] = src[tail + lane];
        ctx_src = CTX_MEMORY;
      } else if (dist >= 64) {
        // overlapping at a distance of 64..1023: 64 bytes per step, a step reads what earlier steps wrote
        gu8* src = dst - dist;
        for (uint32_t k = 0; k < n; k += 64) {
          uint32_t i = k + lane;
          if (i < n) dst[i] = src[i];
        }
        ctx_src = CTX_MEMORY;
      } else {
        // overlapping with a short period: pattern fill from the `dist` bytes before P
        gu8* pat = dst - dist;
        uint32_t m = lane % dist;
        uint32_t step = 64 % dist;
        for (uint32_t k = 0; k < n; k += 64) {
          uint32_t i = k + lane;
          if (i < n) dst[i] = pat[m];
          m += step; if (m >= dist) m -= dist;
        }
        ctx_src = CTX_MEMORY;
      }
      P += n;
      if (clipped) STOP(E_NEEDS_MORE_OUTPUT);
    }
    PROF_REST(prof_copy, prof_t);
command_done:
    if (P >= next_boundary) RING_CROSS();
    if (mlen <= 0) STOP(E_SUCCESS);  // METABLOCK_DONE
    if (quota_mb) RECOMPUTE_QUOTA();
  }
#undef RECOMPUTE_QUOTA
#undef BLOCK_SWITCH
#undef HOTC
#undef STOP
#undef RING_CROSS
done:
#ifdef BROTLI_AMD_PE_DEBUG
  if (blockIdx.x == 0 && lane == 0) printf("done: tick %llu P %llu mlen %d result %d\n", (unsigned long long)__builtin_amdgcn_s_memtime(), (unsigned long long)P, mlen, result);
#endif
#ifdef BROTLI_AMD_PROFILE_SCAN
  if (blockIdx.x == 0 && lane == 0) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (pp_exit != 0) g_path_prof[39] += t_ - pp_exit; else g_path_prof[39] += 0; g_path_prof[31] += t_ - pp_start; }
#endif
  FLUSH_LITERALS();
  FLUSH_PENDING();
  if (helpers_on) {  // the helpers go back to sleep (HC_KIND stays: a helper that looks late must find nothing to do)
    const uint32_t ctl = sp_ctl_base();
    if (split_on) sp_st(ctl, CW_STOP, 1u);
    if (rec_base != 0u) sp_st(rec_base, XW_STOP, 1u);
    for (uint32_t spins = 0; (split_on && sp_ld(ctl, CW_STOP) != 2u) || (rec_base != 0u && sp_ld(rec_base, XW_STOP) != 2u); spins++) {
      if (spins > SP_SPIN_CAP) { hc_st(HC_KIND, (uint32_t)HK_NO_ROUNDS); result = E_UNREACHABLE; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
#undef FLUSH_LITERALS
#undef FLUSH_PENDING
  args->br = br;
  args->P = P; args->next_boundary = next_boundary; args->mlen = mlen;
  args->d0 = d0; args->d1 = d1; args->d2 = d2; args->d3 = d3;
  args->num_commands = num_commands;
  args->engine_commands = engine_commands; args->general_engine = (rfl(args->general_engine) & 0xFFFF00u) | (prefer_general ? 1u : 0u);
#ifdef BROTLI_AMD_PROFILE
  if (lane == 0 && blockIdx.x == 0) printf("lean exits by stage: %u %u %u %u %u %u %u %u\n", prof_stage[0], prof_stage[1], prof_stage[2], prof_stage[3], prof_stage[4], prof_stage[5], prof_stage[6], prof_stage[7]);
  args->prof[0] = prof_cmd; args->prof[1] = prof_lit; args->prof[2] = prof_dist; args->prof[3] = prof_copy;
#ifdef BROTLI_AMD_PROFILE_HDR
  args->prof[4] = 0; args->prof[5] = prof_fast_syms;
#else
  args->prof[4] = prof_fast_batches; args->prof[5] = prof_fast_syms;
#endif
#endif
  return result;
}

// Marshals the stream state into the argument block, runs the loop, takes the results back.
constexpr uint32_t ENGINE_ONLY_MIN_MLEN = 32768;  // (BROTLI_AMD_FLAG_ENGINE_ONLY: smaller metablocks are decoded where they are)
__device__ __forceinline__ int run_commands(Stream& s, const BrotliAmdResume* mid_, uint64_t mb_out_pos, BrotliAmdStreamStatus* st) {
  HotArgs h;
  h.br = s.br; h.ar = s.ar; h.out = s.out; h.dict = s.dict;
  h.out_cap = s.out_cap; h.P = s.P; h.next_boundary = s.next_boundary; h.rb_size = s.rb_size;
  h.window_bits = s.window_bits; h.large_window = s.large_window; h.mlen = s.mlen; h.max_backward = s.max_backward;
  h.d0 = s.dist_rb0; h.d1 = s.dist_rb1; h.d2 = s.dist_rb2; h.d3 = s.dist_rb3;
  h.bl0 = s.bl0; h.bl1 = s.bl1; h.bl2 = s.bl2;
  h.postfix_bits = s.postfix_bits; h.num_direct = s.num_direct;
  lds_sync();
  if (lane_id() == 0) {  // the cold part of the loop's state (see HOTC in process_commands)
    const uint32_t cold[21] = {s.bt_tree0, s.bt_tree1, s.bt_tree2, s.bl_tree0, s.bl_tree1, s.bl_tree2, s.nbt0, s.nbt1, s.nbt2,
                               s.ctx_modes, s.ctx_map, s.dist_ctx_map, s.lit_trees, s.cmd_trees, s.dist_trees, 1, 0, 1, 0, 1, 0};
    for (int k = 0; k < 21; k++) lds_st32(LDS_HOT + 4u * (uint32_t)k, cold[k]);
  }
  lds_sync();
  h.resume_out = (uint64_t)(uintptr_t)&st->resume;
  if (mid_) {
    typedef __attribute__((address_space(1))) const BrotliAmdResume gcresume;
    gcresume* const mid = (gcresume*)(uintptr_t)mid_;
    // This launch continues inside the metablock whose header it has just parsed again: block counts and types, the
    // distance ring, what is left of MLEN and the bit position are those of the command boundary an earlier launch got
    // to.  What was produced up to there plus what is left must be the MLEN just read, the block types must exist.
    const uint64_t mp = rfl(mid->mid_out_pos), mb = rfl(mid->mid_bit_pos);
    const int32_t mm = (int32_t)rfl((uint32_t)mid->mid_mlen);
    uint32_t ty[6];
    bool sane = mm > 0 && mp >= mb_out_pos && (mp - mb_out_pos) + (uint64_t)(uint32_t)mm == (uint64_t)(uint32_t)rfl((uint32_t)s.mlen) && mb >= s.br.pos();
    for (int k = 0; k < 6; k++) {  // (a category's ring starts as {1, 0} whatever its number of types, state.rs:429-435)
      ty[k] = rfl(mid->mid_types[k]);
      sane = sane && (ty[k] <= 1u || ty[k] < (k < 2 ? rfl(s.nbt0) : k < 4 ? rfl(s.nbt1) : rfl(s.nbt2)));
    }
    if (!sane) return E_UNREACHABLE;
    h.P = mp; h.mlen = mm;
    h.bl0 = rfl(mid->mid_bl[0]); h.bl1 = rfl(mid->mid_bl[1]); h.bl2 = rfl(mid->mid_bl[2]);
    h.d0 = (int32_t)rfl((uint32_t)mid->mid_dist_rb[0]); h.d1 = (int32_t)rfl((uint32_t)mid->mid_dist_rb[1]);
    h.d2 = (int32_t)rfl((uint32_t)mid->mid_dist_rb[2]); h.d3 = (int32_t)rfl((uint32_t)mid->mid_dist_rb[3]);
    if (h.rb_size) h.next_boundary = (mp / h.rb_size + 1) * h.rb_size;
    if (lane_id() == 0) for (int k = 0; k < 6; k++) lds_st32(LDS_HOT + 4u * (uint32_t)(15 + k), ty[k]);
    lds_sync();
    h.br.seek(mb);
  }
  h.lut_vgpr = s.lut_vgpr; h.bl_vgpr = s.bl_vgpr;
  h.spec_scratch = (uint64_t)(uintptr_t)(s.ar.glb + s.ar_end);
  h.num_commands = s.num_commands;
  h.engine_commands = s.engine_commands; h.reserved_ = 0u; h.general_engine = s.general_engine;
  h.prof[0] = h.prof[1] = h.prof[2] = h.prof[3] = h.prof[4] = h.prof[5] = 0;
  // tables entirely in the LDS part of the arena (the common case) take the ds_read-only instantiation
  int e;
  // every literal block type with a constant context map (DetectTrivialLiteralBlockTypes, decode.rs:1525-1553)?
  bool ctx_never = true;
  Arena ar_ = s.ar; ar_.uniformize();
  {
    const uint32_t nbt0 = rfl(s.nbt0), ctx_map = rfl(s.ctx_map);
    for (uint32_t bt = 0; bt < nbt0; bt++) {
      uint32_t mine = ar_.ld8_lane<false>(ctx_map + (bt << 6) + lane_id());
      if (__ballot(mine != rdlane(mine, 0)) != 0ull) { ctx_never = false; break; }
    }
  }
  if (rfl(s.ar.top) <= rfl(s.ar.lds_limit)) {
    if (!ctx_never && (rfl(s.flags) & BROTLI_AMD_FLAG_ENGINE_ONLY) && rfl((uint32_t)s.mlen) >= ENGINE_ONLY_MIN_MLEN) {
      s.num_metablocks--; return E_RETRY_ARENA;  // (see the flag; nothing of this metablock has been output yet)
    }
    e = ctx_never ? process_commands<true, true>(&h) : process_commands<true, false>(&h);
  } else if (ctx_never && ar_.lds_limit >= TREE_CACHE_BYTES && (g_engine_mode & 16u) == 0u && rfl(s.large_window) == 0u) {
    // (not for large-window streams: a cache slot for a distance tree holds the 928 entries the 520-symbol alphabet takes at
    // most, a large-window alphabet of up to 1128 symbols builds tables of up to 1528 -- ADVICE round 4; such a metablock spills)
    // More tables than the LDS part holds (binaries at -q 5 and up: dozens of block types, a tree each), but literals that do not
    // depend on context: at any time the loop reads ONE literal tree, one command tree and the distance trees of one block type.
    // What lies in the LDS part goes to its place in the block's global scratch (the arena addresses both with the same offsets),
    // the LDS part becomes a cache of the trees in use -- refilled at the block switches, every few hundred commands --, and the
    // loop is the one for tables in LDS, command engine and all.  (Round 4; before: every table lookup of such a metablock a round
    // trip to memory, 1.1 GB/s for 256 copies of libc.so.6 at -q 5.)
    for (uint32_t off = lane_id() * 16u; off < ar_.lds_limit; off += 1024u)
      *reinterpret_cast<gu32x4*>(ar_.glb + off) = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[LDS_FIXED + off]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    h.ar.lds_limit = 0u;
    e = process_commands<true, true, true>(&h);
  } else if (!ctx_never && ar_.lds_limit >= TREE_CACHE_CTX_BYTES && (g_engine_mode & 16u) == 0u && rfl(s.large_window) == 0u && lit_types_fit_cache(ar_, rfl(s.nbt0), rfl(s.ctx_map), tree_cache_lit_slots(ar_.lds_limit)) &&
             !((rfl(s.flags) & BROTLI_AMD_FLAG_ENGINE_ONLY) && rfl((uint32_t)s.mlen) >= ENGINE_ONLY_MIN_MLEN)) {   // (engine blocks hand such a metablock back, as above)
    // ... and the same where literals do depend on context, as long as no literal block type names more trees than the LDS part has
    // slots for: the loop for context-modelled metablocks out of LDS, command records included (their ring lies behind the cache)
    for (uint32_t off = lane_id() * 16u; off < ar_.lds_limit; off += 1024u)
      *reinterpret_cast<gu32x4*>(ar_.glb + off) = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[LDS_FIXED + off]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    h.reserved_ = ar_.lds_limit; h.ar.lds_limit = 0u;
    e = process_commands<true, false, true>(&h);
  } else {
    if (rfl(s.flags) & BROTLI_AMD_FLAG_NO_SPILL) { s.num_metablocks--; return E_RETRY_ARENA; }  // nothing of this metablock has been output yet (the next pass counts it)
    s.num_spilled++;
    e = process_commands<false, false>(&h);
  }
  e = rfl(e);
  s.br = h.br; s.br.uniformize();
  s.P = rfl(h.P); s.next_boundary = rfl(h.next_boundary); s.mlen = rfl(h.mlen);
  s.dist_rb0 = rfl(h.d0); s.dist_rb1 = rfl(h.d1); s.dist_rb2 = rfl(h.d2); s.dist_rb3 = rfl(h.d3);
  s.num_commands = rfl(h.num_commands);
  s.engine_commands = rfl(h.engine_commands);
  s.general_engine = rfl(h.general_engine);
#ifdef BROTLI_AMD_PROFILE
  s.prof[0] += h.prof[0]; s.prof[1] += h.prof[1]; s.prof[2] += h.prof[2]; s.prof[3] += h.prof[3]; s.prof[4] += h.prof[4]; s.prof[5] += h.prof[5];
#endif
  return e;
}

// decode.rs:1754-1806: stored metablock = byte-aligned memcpy of MLEN bytes, all lanes
__device__ __noinline__ int copy_uncompressed(Stream& s) {
  BitReader br = s.br; br.uniformize();
  const uint32_t lane = lane_id();
  uint64_t byte = br.pos() >> 3;
  uint64_t in_size = br.total_bits() >> 3;
  uint64_t avail = in_size > byte ? in_size - byte : 0;
  const uint64_t mlen0 = (uint64_t)rfl((uint32_t)s.mlen), P0 = rfl(s.P);
  uint64_t n = mlen0 < avail ? mlen0 : avail;
  uint64_t room = rfl(s.out_cap) - P0;
  bool clipped = n > room;
  if (clipped) n = room;
  gcu8* src = rfl_ptr(s.in_bytes) + byte;
  gu8* dst = rfl_ptr(s.out) + P0;
  for (uint64_t k = lane; k < n; k += 64) dst[k] = src[k];
  s.P += n;
  s.mlen -= (int32_t)n;
  if (clipped) return E_NEEDS_MORE_OUTPUT;
  // flush points inside the copy (ring wraps; a canny ring holds the whole block)
  if (s.rb_size == (1ull << s.window_bits)) while (s.P >= s.next_boundary) s.next_boundary += s.rb_size;
  br.seek((byte + n) * 8);
  s.br = br;
  if (s.mlen != 0) return E_NEEDS_MORE_INPUT;
  return E_SUCCESS;
}

// ring buffer size chosen at the first non-empty data metablock (decode.rs:1808-1871)
__device__ __forceinline__ void allocate_ring(Stream& s, const BitReader& br) {
  uint32_t is_last = s.is_last;
  uint64_t rb = 1ull << s.window_bits;
  if (s.is_uncompressed) {
    uint64_t byte = (br.pos() >> 3) + (uint64_t)(uint32_t)s.mlen;
    if (byte < (br.total_bits() >> 3)) {
      uint32_t b = rfl(s.in_bytes[byte]);
      if ((b & 3u) == 3u) is_last = 1;
    }
  }
  if (is_last && !(s.flags & BROTLI_AMD_FLAG_NO_CANNY)) {
    while ((int64_t)rb >= ((int64_t)s.mlen + 16) * 2 && rb > 32) rb >>= 1;
  }
  s.rb_size = rb;
  s.next_boundary = (s.P / rb + 1) * rb;
}

__device__ __forceinline__ void store_resume(const Stream& s, const BitReader& br, BrotliAmdResume* r) {
  if (lane_id() == 0) {
    r->bit_pos = br.pos();
    r->out_pos = s.P;
    r->dist_rb[0] = s.dist_rb0; r->dist_rb[1] = s.dist_rb1; r->dist_rb[2] = s.dist_rb2; r->dist_rb[3] = s.dist_rb3;
    r->dist_rb_idx = s.dist_rb_idx;
    r->window_bits = s.window_bits;
    r->large_window = s.large_window;
    r->rb_size_log2 = s.rb_size ? (uint32_t)(63 - __clzll((long long)s.rb_size)) : 0;
    r->is_last_done = 0;
    r->reserved = 0;
    r->mid_valid = 0;  // (command boundaries noted inside the metablock that ends here are history)
  }
}

// src/decode.rs:2779-3403 restated for one whole-input call
__device__ __forceinline__ int decode_stream(Stream& s, bool have_header, uint64_t in_size, BrotliAmdStreamStatus* st, const BrotliAmdResume* mid) {
  if (!have_header) {
    ColdScope c(s);
    BitReader& br = c.br;
    // UNINITED: DecodeWindowBits needs one whole byte (decode.rs:2921-2939, 152-187)
    if (in_size == 0) return E_NEEDS_MORE_INPUT;
    s.large_window = 0;
    if (br.read(1) == 0) s.window_bits = 16;
    else {
      uint32_t n = br.read(3);
      if (n != 0) s.window_bits = 17 + n;
      else {
        n = br.read(3);
        if (n == 1) {
          if (!(s.flags & BROTLI_AMD_FLAG_LARGE_WINDOW)) return E_WINDOW_BITS;
          if (br.read(1) == 1) return E_WINDOW_BITS;
          s.large_window = 1;
        } else if (n != 0) s.window_bits = 8 + n;
        else s.window_bits = 17;
      }
    }
    if (s.large_window) {
      s.window_bits = br.read(6); NEED_INPUT(br);
      if (s.window_bits < 10 || s.window_bits > 30) return E_WINDOW_BITS;
    }
    store_resume(s, br, &st->resume);
  }
  s.max_backward = (int32_t)((1u << s.window_bits) - 16u);

  for (;;) {
    // METABLOCK_BEGIN (state.rs:422-450)
#ifdef BROTLI_AMD_PROFILE_HDR
    const uint64_t hdr_t0 = __builtin_amdgcn_s_memtime();
#endif
    s.bl0 = s.bl1 = s.bl2 = 1u << 24;
    s.nbt0 = s.nbt1 = s.nbt2 = 1;
    s.ar.top = 0;
    s.ar.cold = s.ar_end;
    const uint32_t counted_before = s.num_metablocks;  // (what is counted from here on lies behind the resume point: see E_RETRY_ARENA below)
    {
      ColdScope c(s);
      BitReader& br = c.br;
      // A run of metadata blocks (streams made of tens of thousands of them exist: empty.compressed.17 / .18 of the
      // reference) is skipped here, block after block, without leaving the reader's registers; the resume point moves
      // again at the next metablock that is not one.
      const uint64_t stream_bits = br.total_bits();
      const uint32_t reader_skip = BitReader::skip_bits();
      for (;;) {
        {
          // the common shape in one look: ISLAST = 0, MNIBBLES = 3, reserved = 0, MSKIPBYTES = k, MSKIPLEN - 1 in k bytes
          // (decode.rs:243-372); anything else, and every error, goes through the general parser below
          uint32_t run = 0;
          for (;;) {
            const uint32_t pk = br.peek32();
            const uint32_t kb = (pk >> 4) & 3u;
            const uint32_t len = kb == 0u ? 0u : ((pk >> 6) & 0xFFu) + 1u;
            if ((pk & 0xFu) != 6u || kb > 1u || len > 128u) break;  // (at most 14 header bits and 7 of padding: all inside pk)
            const uint64_t p0 = (uint64_t)br.next_dw * 32 - br.cnt - reader_skip;
            const uint32_t hdr = 6u + 8u * kb, pad = (8u - (uint32_t)((p0 + hdr) & 7u)) & 7u;
            if (p0 + hdr + pad + 8ull * len > stream_bits || ((pk >> hdr) & mask_bits(pad)) != 0u) break;
            if (hdr + pad + 8u * len <= 32u) br.drop(hdr + pad + 8u * len);  // (peek32 left at least 32 bits in the buffer)
            else { br.drop(hdr + pad); br.advance(len * 8u); }
            run++;
          }
          if (run != 0u) { s.num_metablocks += run; s.is_last = 0; s.is_metadata = 1; s.is_uncompressed = 0; s.mlen = 0; }
        }
        TRY(decode_metablock_length(br, s));
        s.num_metablocks++;
        if ((s.is_metadata || s.is_uncompressed) && !jump_to_byte_boundary(br)) return E_PADDING_2;  // decode.rs:2990-2994
        if (!s.is_metadata) break;
        // skip MLEN bytes (decode.rs:3031-3045)
        uint64_t byte = br.pos() >> 3, isz = br.total_bits() >> 3;
        uint64_t avail = isz > byte ? isz - byte : 0;
        if ((uint64_t)(uint32_t)s.mlen > avail) { br.seek(isz * 8 + 8); return E_NEEDS_MORE_INPUT; }
        // (a short skip stays in the reader's window: a seek fetches the input anew, 12 K clocks per metadata block)
        if ((uint32_t)s.mlen <= 128u) { if (s.mlen != 0) br.advance((uint32_t)s.mlen * 8u); }
        else br.seek((byte + (uint32_t)s.mlen) * 8);
        s.mlen = 0;
        if (s.is_last) break;
      }
      if (!s.is_metadata && s.mlen != 0 && s.rb_size == 0) allocate_ring(s, br);
    }
    if (mid && (s.is_metadata || s.mlen == 0 || s.is_uncompressed)) return E_UNREACHABLE;  // (the resume block names a command inside a compressed metablock)
    if (!s.is_metadata && s.mlen != 0) {
      if ((s.flags & BROTLI_AMD_FLAG_PROBE) && s.is_uncompressed) { s.engine_commands = 0u; return E_PROBE; }   // (a stored metablock first: no engine's stream)
      if (s.is_uncompressed) {
        TRY(copy_uncompressed(s));
      } else {
        {
          ColdScope c(s);
          BitReader& br = c.br;
          // HUFFMAN_CODE_0..3 (decode.rs:3046-3140)
          for (int k = 0; k < 3; k++) {
            uint32_t nbt;
            TRY(decode_varlen_uint8(br, &nbt));
            nbt += 1;
            if (k == 0) s.nbt0 = nbt; else if (k == 1) s.nbt1 = nbt; else s.nbt2 = nbt;
            if (nbt < 2) continue;
            uint32_t tt, tl;
            TRY(c.huffman(nbt + 2, nbt + 2, &tt, true));
            TRY(c.huffman(26, 26, &tl, true));
            uint32_t len = read_block_length(br, s.ar, s.bl_vgpr, tl); NEED_INPUT(br);
            if (k == 0) { s.bl0 = len; s.bt_tree0 = tt; s.bl_tree0 = tl; }
            else if (k == 1) { s.bl1 = len; s.bt_tree1 = tt; s.bl_tree1 = tl; }
            else { s.bl2 = len; s.bt_tree2 = tt; s.bl_tree2 = tl; }
          }
          // METABLOCK_HEADER_2 + CONTEXT_MODES (decode.rs:3141-3172)
          uint32_t bits = br.read(6); NEED_INPUT(br);
          s.postfix_bits = bits & 3u;
          s.num_direct = 16 + ((bits >> 2) << s.postfix_bits);
          s.ctx_modes = s.ar.alloc_cold(s.nbt0);
          for (uint32_t k = 0; k < s.nbt0; k++) { uint32_t m = br.read(2); NEED_INPUT(br); s.ar.st8(s.ctx_modes + k, m); }
        }
        TRY(decode_context_map(s, s.nbt0 << 6, &s.num_lit_trees, &s.ctx_map));
        if (s.flags & BROTLI_AMD_FLAG_PROBE) {
          // what kind of stream is this?  (every literal block type with a constant context map: DetectTrivialLiteralBlockTypes,
          // decode.rs:1525-1553 -- the command engines' kind)
          bool ctx_never = true;
          Arena ar_ = s.ar; ar_.uniformize();
          const uint32_t nbt0 = rfl(s.nbt0), ctx_map = rfl(s.ctx_map);
          for (uint32_t bt = 0; bt < nbt0; bt++) {
            uint32_t mine = ar_.ld8_lane<false>(ctx_map + (bt << 6) + lane_id());
            if (__ballot(mine != rdlane(mine, 0)) != 0ull) { ctx_never = false; break; }
          }
          s.engine_commands = 1u | (ctx_never ? 2u : 0u) | (rfl((uint32_t)s.mlen) >= ENGINE_ONLY_MIN_MLEN ? 4u : 0u);
          if (s.engine_commands != 7u) return E_PROBE;
          // (round 6) ... and are its commands SHORT?  The command engines take a stream of long copies and long literal runs at ten
          // times what one wave does; text -- a dozen bytes a command -- they take at a tenth of that, and four streams a CU on a wave each
          // with the command records and the hand-written run (lean_rec_commands) do 2.4 times as much (1024 x lcet10 at -q 5: 10.4 -> 25 GB/s).
          // What the command codes say -- each of them: of the bytes a code's commands stand for (insert base + copy base, each command by the code's own probability: a
          // symbol of length L has 2^(8 - L) of the root table's slots, or its share of a second-level table), which part comes from commands that insert or
          // copy more than 63 bytes -- what the run does not take.  Text: a few per cent; the metric's make-up and the survey's: nine tenths.  Bit 3.
          {
            uint32_t ndirect = s.num_direct - 16;
            uint32_t num_dist_codes = 16 + ndirect + ((s.large_window ? 62u : 24u) << (s.postfix_bits + 1));
            (void)num_dist_codes;
            TRY(decode_context_map(s, s.nbt2 << 2, &s.num_dist_trees, &s.dist_ctx_map));
            TRY(decode_tree_group(s, 256, 256, s.num_lit_trees, &s.lit_trees));
            TRY(decode_tree_group(s, 704, 704, s.nbt1, &s.cmd_trees));
            Arena a2 = s.ar; a2.uniformize();
            bool every_short = true;   // (every command block type's code: the encoder gives a stream's parts codes of their own -- the metric's seed one, its copies another)
            for (uint32_t ct = 0; ct < rfl(s.nbt1); ct++) {
            const uint32_t tree = a2.ld32<false>(rfl(s.cmd_trees) + 4u * ct);
            // (bytes by the code's own probabilities: a slot of the root table weighs 128, a slot of a second-level table of k index bits 128 >> k;
            // a command's bytes: its insert base + its copy base; LONG: either beyond 63)
            uint32_t all = 0, lng = 0;
            auto take = [&](const uint32_t cmd, const uint32_t w) {
              if (cmd >= 704u) return;
              const uint32_t cell = cmd >> 6;
              const uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u), copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);
              const uint32_t ib = kInsBase[ins_code], cb = kCopyBase[copy_code], v = (ib + cb) * w;
              all += v; if (ib > 63u || cb > 63u) lng += v;
            };
            auto entry = [&](const uint32_t off) -> uint32_t { return off < a2.lds_limit ? lds_ld16(LDS_FIXED + off) : (uint32_t)*reinterpret_cast<gu16*>(a2.glb + off); };
            for (uint32_t k = 0; k < 4u; k++) {
              const uint32_t e = entry(tree + ((lane_id() * 4u + k) << 1));
              const uint32_t len = e & 15u;
              if (len == 0u) continue;
              if (len <= (uint32_t)ROOT_BITS) take(e >> 4, 128u);
              else {
                const uint32_t kb = len - (uint32_t)ROOT_BITS;   // (a pointer: its second-level table has 2^kb slots from entry e >> 4 on)
                for (uint32_t j2 = 0; j2 < (1u << kb); j2++) take(entry(tree + (((e >> 4) + j2) << 1)) >> 4, 128u >> kb);
              }
            }
            for (uint32_t m = 32u; m != 0u; m >>= 1) { all += __shfl_xor(all, m); lng += __shfl_xor(lng, m); }
            if (all == 0u || (uint64_t)lng * 100u >= (uint64_t)all * PROBE_LONG_PERCENT) every_short = false;
#ifdef BROTLI_AMD_REC_DEBUG
            if (blockIdx.x < 2u && lane_id() == 0) printf("probe: block %u code %u of %u: all %u long %u tree %u lds_limit %u mlen %d\n", blockIdx.x, ct, s.nbt1, all, lng, tree, a2.lds_limit, s.mlen);
#endif
            }
            if (every_short) s.engine_commands |= 8u;
          }
          return E_PROBE;
        }
        uint32_t ndirect = s.num_direct - 16;
        uint32_t num_dist_codes = 16 + ndirect + ((s.large_window ? 62u : 24u) << (s.postfix_bits + 1));
        uint32_t max_dist_symbol = s.large_window ? max_distance_symbol(ndirect, s.postfix_bits) : num_dist_codes;
        TRY(decode_context_map(s, s.nbt2 << 2, &s.num_dist_trees, &s.dist_ctx_map));
        TRY(decode_tree_group(s, 256, 256, s.num_lit_trees, &s.lit_trees));
        TRY(decode_tree_group(s, 704, 704, s.nbt1, &s.cmd_trees));
        TRY(decode_tree_group(s, num_dist_codes, max_dist_symbol, s.num_dist_trees, &s.dist_trees));
        {  // (scratch accounting of the reference's prealloc entry point: trees and map bytes alive in this metablock)
          const uint32_t trees = s.num_lit_trees + s.nbt1 + s.num_dist_trees, maps = s.nbt0 * 65u + s.nbt2 * 4u;
          s.peak_trees = trees > s.peak_trees ? trees : s.peak_trees; s.peak_maps = maps > s.peak_maps ? maps : s.peak_maps; s.any_compressed = 1;
        }
#ifdef BROTLI_AMD_PROFILE_HDR
        s.prof[4] += __builtin_amdgcn_s_memtime() - hdr_t0;
#endif
#ifdef BROTLI_AMD_PROFILE_HDR
        const uint64_t run_t0 = __builtin_amdgcn_s_memtime();
        int run_e = run_commands(s, mid, s.P, st); mid = nullptr;
        s.prof[5] = (s.prof[5] & 0xFFFFFu) + ((__builtin_amdgcn_s_memtime() - run_t0) >> 8);
        TRY(run_e);
#else
        {
          const BrotliAmdResume* const m_ = mid; mid = nullptr;
          const int run_e = run_commands(s, m_, s.P, st);
          // a pass that stops in front of this metablock goes on, next time, from the resume point -- which lies in front of a run of
          // metadata blocks that came before it: the next pass counts those again (num_metablocks was inflated)
          if (run_e == E_RETRY_ARENA) s.num_metablocks = counted_before;
          TRY(run_e);
        }
#endif
      }
    }
    // METABLOCK_DONE (decode.rs:3345-3381)
    if (s.mlen < 0) return E_BLOCK_LENGTH_2;
    {
      ColdScope c(s);
      BitReader& br = c.br;
      if (!s.is_last) { store_resume(s, br, &st->resume); continue; }
      if (!jump_to_byte_boundary(br)) return E_PADDING_2;
      NEED_INPUT(br);
      store_resume(s, br, &st->resume);
    }
    return E_SUCCESS;
  }
}

}  // namespace

// One decoding wave (+ up to seven helper waves) per stream; persistent blocks pull stream indices from `queue`.
// This file compiles into two objects (Makefile): the kernel of the launches without gangs, and -- with -DBROTLI_AMD_GANG_KERNEL -- the kernel of the
// gang launches, the only one that has the gang's form of the path engine in its call graph.  One kernel for both cost the metric 4.5 %
// (6.39 -> 6.68 ms) while the gang's decision sat in process_commands' registers (see there); since it does not, the second kernel costs
// launches without gangs nothing (BROTLI_AMD_GANG_KERNEL_ALWAYS=1 runs them through it) -- two objects all the same: insurance.
#ifdef BROTLI_AMD_GANG_KERNEL
#define BROTLI_AMD_KERNEL brotli_amd_decode_gang_kernel
#define BROTLI_AMD_LAUNCH brotli_amd_launch_decode_gang
#else
#define BROTLI_AMD_KERNEL brotli_amd_decode_kernel
#define BROTLI_AMD_LAUNCH brotli_amd_launch_decode
#endif
extern "C" __global__ __launch_bounds__(1024, 4) void BROTLI_AMD_KERNEL(const BrotliAmdStreamDesc* __restrict__ descs,
                                                                           BrotliAmdStreamStatus* __restrict__ status, uint32_t n_streams,
                                                                           uint32_t* __restrict__ queue, uint8_t* __restrict__ scratch,
                                                                           uint64_t scratch_per_block, uint32_t lds_arena_bytes,
                                                                           const uint8_t* __restrict__ dict) {
  const uint32_t lane = lane_id();
#ifdef BROTLI_AMD_PROFILE_SCAN
  const uint64_t scan_prof_t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef BROTLI_AMD_PROFILE_HDR
  const uint64_t hdr_prof_t0 = __builtin_amdgcn_s_memtime();
#endif
  // A gang launch (queue[2] blocks a stream, sixteen waves each; see GC_*): block b is member (b mod 8 gang) / 8 of the gang of stream
  // (b / (8 gang)) 8 + b mod 8 -- the members of a gang are eight block numbers apart, which is how the hardware deals blocks to the same
  // XCD (their L2 is one: what they hand each other does not cross the fabric; a matter of speed, not of correctness).  Member 0 owns the stream.
// A POOL launch (queue[2] bit 8, BROTLI_AMD_GANG_POOL_FLAG; as many blocks as CUs, at most as many streams): nobody is dealt to a gang.  Blocks take streams from the queue
  // as ever; a block that finds the queue empty -- at once, where the batch has fewer streams than CUs; when its own stream is done, otherwise --
  // joins the largest stream that is still being decoded and has fewer than seven helpers, for as long as that stream lasts, and then the next.
  // A stream's gang is whoever has joined it when one of its invocations of the path engine starts (GC_JOINED, GC_MEMBERS).
#ifdef BROTLI_AMD_GANG_KERNEL
  uint32_t gang_m = rfl(queue[2]), gang_role = 0, gang_stream = 0; uint64_t gang_addr = 0;
  const bool pool = (gang_m & BROTLI_AMD_GANG_POOL_FLAG) != 0u && blockDim.x == 64u * SC_WAVES && lds_arena_bytes <= GC_ARENA_CAP;
  const uint64_t pool_base = (uint64_t)rfl(queue[4]) | ((uint64_t)rfl(queue[5]) << 32);
  if ((gang_m & BROTLI_AMD_GANG_POOL_FLAG) != 0u) gang_m = 1u;
  bool pool_open = false;   // (this block's stream has a control block that says it is being decoded)
#else
  uint32_t gang_m = 1u, gang_role = 0, gang_stream = 0; uint64_t gang_addr = 0;
#endif
  if (gang_m > 1u && gang_m <= 16u && blockDim.x == 64u * SC_WAVES && lds_arena_bytes <= GC_ARENA_CAP) {
    gang_role = (blockIdx.x % (8u * gang_m)) >> 3;
    gang_stream = (blockIdx.x / (8u * gang_m)) * 8u + (blockIdx.x & 7u);
    if (gang_stream >= n_streams) return;   // (the streams are not a multiple of eight: a gang without a stream)
    gang_addr = ((uint64_t)rfl(queue[4]) | ((uint64_t)rfl(queue[5]) << 32)) + (uint64_t)gang_stream * GC_STRIDE;
  } else gang_m = 1u;
  const uint32_t scratch_slot = gang_m > 1u ? gang_stream : blockIdx.x;   // (this block's part of `scratch`: a gang's helper blocks have none, its owner takes its stream's)
  // waves 1.. are helpers (see helper_wave); the mailbox is cleared before the waves part ways
  if ((uint32_t)(uintptr_t)g_dynamic_lds == 0u) {
    const uint32_t nw = blockDim.x >> 6, nr = nw < SPEC_MAX_WAVES ? nw : SPEC_MAX_WAVES;  // waves in the block, waves that take part in rounds
    // behind the table arena: the rounds' slots -- or, in blocks of sixteen waves, the command engine's rings, and the
    // slots lie inside them (on REC: rounds and engine never run at the same time; process_commands clears the slots'
    // mailbox words after every invocation of the engine)
    const uint32_t behind = LDS_FIXED + lds_arena_bytes, slots = nw == SC_WAVES ? behind + SC_REC : behind;
    if (threadIdx.x < 20u)
      lds_st32(LDS_HCTL + 4u * threadIdx.x, threadIdx.x == HC_KIND && nw < 2u ? (uint32_t)HK_NO_ROUNDS : threadIdx.x == HC_BASE ? slots :
                                            threadIdx.x == HC_NW ? nr : threadIdx.x == HC_NW_ALL ? nw :
                                            threadIdx.x == HC_SCAN_BASE && nw == SC_WAVES ? behind :
                                            threadIdx.x == HC_GANG_ROLE ? gang_role : threadIdx.x == HC_GANG_M ? gang_m :
                                            threadIdx.x == HC_GANG_LO ? (uint32_t)gang_addr : threadIdx.x == HC_GANG_HI ? (uint32_t)(gang_addr >> 32) :
                                            0u);
    if (nw >= 2u && threadIdx.x < nr * 16u) lds_st32(slots + (threadIdx.x >> 4) * HL_SLOT + HL_CTL + 4u * (threadIdx.x & 15u), 0u);
  }  // (launched without helper waves: no rounds)
  __syncthreads();
  if (gang_role != 0u) {   // a helper block of a gang: the path engine's regions of its owner's stream, nothing else (see there)
    if ((uint32_t)(uintptr_t)g_dynamic_lds != 0u) return;   // (no LDS addressing: the owner finds the gang short of a block and goes on alone)
    if (queue[6] != 0u) return;   // (tests: helpers that never turn up -- BROTLI_AMD_GANG_NO_HELPERS --, as when the device has no CU for them)
#ifdef BROTLI_AMD_GANG_KERNEL
    if (threadIdx.x == 0u) (void)gang_add32(gang_ctl(), GC_JOINED, 1u);
    (void)pe16r::path_engine(rfl(threadIdx.x >> 6));
#endif
    return;
  }
  if (rfl(threadIdx.x >> 6) != 0u) {
    if ((uint32_t)(uintptr_t)g_dynamic_lds == 0u)
      helper_wave(rfl(threadIdx.x >> 6), as_global<gu8>(scratch + (uint64_t)scratch_slot * scratch_per_block + (scratch_per_block - BROTLI_AMD_SPEC_SCRATCH)));
    return;
  }
  // the decoding wave is a chain of dependent instructions; the waves beside it on its SIMD (helpers of this and other blocks)
  // poll and parse ahead: it goes first (measured: C2 14.0 -> 13.5 ms, 1024 x 1 MiB 8.58 -> 8.37 ms, engine blocks unchanged)
  __builtin_amdgcn_s_setprio(BROTLI_AMD_DECODER_PRIO);
  // all LDS addressing is absolute (see g_smem): the dynamic LDS block must start at LDS address 0.  If a toolchain
  // ever puts it elsewhere nothing below may touch LDS: every stream of this block is reported as failed instead.
  if (rfl((uint32_t)(uintptr_t)g_dynamic_lds) != 0u) {
    for (;;) {
      uint32_t idx = 0;
      if (lane == 0) idx = atomicAdd(queue, 1u);
      idx = rfl(idx);
      if (idx >= n_streams) return;
      if (lane == 0) {
        BrotliAmdStreamStatus* st = status + idx;
        st->result = 0; st->error_code = E_UNREACHABLE; st->decoded_size = 0; st->consumed = 0; st->produced = 0;
        st->num_metablocks = 0; st->spilled_metablocks = 0; st->num_commands = 0; st->engine_commands = 0;
      }
    }
  }
  // literal context LUT -> LDS once per block
  for (uint32_t i = lane; i < 2048; i += 64) lds_st8(LDS_CTX_LUT + i, kContextLookup[i]);
  // per-lane LUT images
  uint32_t lut = 0, bl = 0;
  if (lane < 24) lut = (uint32_t)kInsBase[lane] | ((uint32_t)kInsExtra[lane] << 16);
  else if (lane >= 32 && lane < 56) lut = (uint32_t)kCopyBase[lane - 32] | ((uint32_t)kCopyExtra[lane - 32] << 16);
  if (lane < 26) bl = (uint32_t)kBlockLenBase[lane] | ((uint32_t)kBlockLenExtra[lane] << 16);
  lds_sync();

  for (uint32_t pulls = 0;; pulls++) {
    uint32_t idx = 0;
#ifdef BROTLI_AMD_GANG_KERNEL
    if (pool && pool_open) {   // (the stream before: done, whichever way its turn of this loop ended -- its helpers may go, the pool has one stream less)
      if (lane == 0) { gang_st32(gang_ctl(), GC_EPOCH, GC_QUIT); (void)__hip_atomic_fetch_add(queue + 8, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      pool_open = false;
    }
#endif
    if (gang_m > 1u) {   // (a gang's owner: its stream, and no other)
      if (pulls != 0u) break;
      idx = gang_stream;
    } else {
      if (lane == 0) idx = atomicAdd(queue, 1u);
      idx = rfl(idx);
    }
    if (idx >= n_streams) break;
    // (the host may give an order in which to take the streams -- longest first, so that the last blocks to finish do not
    // start a long stream when the others are done: queue[1] != 0, stream of the k-th pull in queue[16 + k])
    if (queue[1] != 0u) idx = rfl(queue[16u + idx]);
#ifdef BROTLI_AMD_GANG_KERNEL
    if (pool) {   // this block owns the stream: its control block, nobody's help yet
      const uint64_t ca = pool_base + (uint64_t)idx * GC_STRIDE;
      hc_st(HC_GANG_LO, (uint32_t)ca); hc_st(HC_GANG_HI, (uint32_t)(ca >> 32)); hc_st(HC_GANG_ROLE, 0u); hc_st(HC_GANG_M, 0x108u);
      hc_st(HC_GANG_EPOCH, 0u); hc_st(HC_GANG_READY, 0u);
      lds_sync();
      pool_open = true;
    }
#endif
    const BrotliAmdStreamDesc d = descs[idx];
    BrotliAmdStreamStatus* st = status + idx;
    if (d.flags & BROTLI_AMD_FLAG_DEFER) {   // not this launch's stream (see the flag)
      if (lane == 0) {
        BrotliAmdResume z; z.bit_pos = 0; z.out_pos = 0; z.dist_rb[0] = z.dist_rb[1] = z.dist_rb[2] = z.dist_rb[3] = 0; z.dist_rb_idx = 0;
        z.window_bits = 0; z.large_window = 0; z.rb_size_log2 = 0; z.is_last_done = 0; z.reserved = 0; z.mid_valid = 0;
        st->result = BROTLI_AMD_RESULT_RETRY_ARENA; st->error_code = E_RETRY_ARENA; st->decoded_size = 0; st->consumed = 0; st->produced = 0;
        st->num_metablocks = 0; st->spilled_metablocks = 0; st->num_commands = 0; st->engine_commands = 0;
        st->peak_trees = 0; st->peak_map_bytes = 0; st->ring_bytes = 0; st->any_compressed = 0; st->resume = z;
      }
      continue;
    }

    Stream s;
    s.ar.glb = as_global<gu8>(scratch + (uint64_t)scratch_slot * scratch_per_block);
    s.ar.lds_limit = lds_arena_bytes;
    s.ar.top = 0;
    s.ar_end = (uint32_t)(scratch_per_block - BROTLI_AMD_SPEC_SCRATCH) & ~3u;
    s.ar.cold = s.ar_end;
    s.out = as_global<gu8>(d.out); s.out_cap = d.out_cap;
    s.dict = as_global<gcu8>(dict);
    s.in_bytes = as_global<gcu8>(d.in);
    s.flags = d.flags;
    s.lut_vgpr = lut; s.bl_vgpr = bl;
    s.num_metablocks = 0; s.num_spilled = 0; s.num_commands = 0; s.engine_commands = 0; s.general_engine = 0;
    s.peak_trees = 0; s.peak_maps = 0; s.any_compressed = 0;
    s.mlen = 0;
#ifdef BROTLI_AMD_PROFILE
    s.prof[0] = s.prof[1] = s.prof[2] = s.prof[3] = s.prof[4] = s.prof[5] = 0;
    const uint64_t prof_start = __builtin_amdgcn_s_memtime();
#endif
    s.is_last = 0; s.is_uncompressed = 0; s.is_metadata = 0;
    s.br.set_input((uint64_t)d.in, d.in_size);
    const bool resume = (d.flags & BROTLI_AMD_FLAG_RESUME) && d.resume.window_bits != 0;
    if (resume && (d.resume.out_pos > d.out_cap || d.resume.bit_pos > d.in_size * 8 ||
                   (d.resume.mid_valid != 0u && (d.resume.mid_out_pos > d.out_cap || d.resume.mid_bit_pos > d.in_size * 8)))) {  // a resume block that does not belong to these buffers
      if (lane == 0) {
        st->result = 0; st->error_code = E_UNREACHABLE; st->decoded_size = 0; st->consumed = 0; st->produced = 0;
        st->num_metablocks = 0; st->spilled_metablocks = 0; st->num_commands = 0; st->engine_commands = 0; st->resume = d.resume;
      }
      continue;
    }
    if (resume) {
      s.P = d.resume.out_pos;
      s.dist_rb0 = d.resume.dist_rb[0]; s.dist_rb1 = d.resume.dist_rb[1]; s.dist_rb2 = d.resume.dist_rb[2]; s.dist_rb3 = d.resume.dist_rb[3];
      s.dist_rb_idx = d.resume.dist_rb_idx;
      s.window_bits = d.resume.window_bits; s.large_window = d.resume.large_window;
      s.rb_size = d.resume.rb_size_log2 ? (1ull << d.resume.rb_size_log2) : 0;
      s.next_boundary = s.rb_size ? (s.P / s.rb_size + 1) * s.rb_size : 0;
      s.br.seek(d.resume.bit_pos);
      if (lane == 0) st->resume = d.resume;
    } else {
      if (lane == 0) {  // no metablock boundary yet (the host looks at window_bits to tell)
        BrotliAmdResume z; z.bit_pos = 0; z.out_pos = 0; z.dist_rb[0] = z.dist_rb[1] = z.dist_rb[2] = z.dist_rb[3] = 0; z.dist_rb_idx = 0;
        z.window_bits = 0; z.large_window = 0; z.rb_size_log2 = 0; z.is_last_done = 0; z.reserved = 0; z.mid_valid = 0;
        st->resume = z;
      }
      s.P = 0;
      s.dist_rb0 = 4; s.dist_rb1 = 11; s.dist_rb2 = 15; s.dist_rb3 = 16;  // state.rs:296, most recent first
      s.dist_rb_idx = 0;
      s.window_bits = 0; s.large_window = 0;
      s.rb_size = 0; s.next_boundary = 0;
      s.br.seek(0);
    }

    const bool mid = resume && d.resume.mid_valid != 0u;
#ifdef BROTLI_AMD_PROFILE_HDR
    const uint64_t hdr_prof_t1 = __builtin_amdgcn_s_memtime();
#endif
    int e = decode_stream(s, resume, d.in_size, st, mid ? &descs[idx].resume : nullptr);
#ifdef BROTLI_AMD_PROFILE_HDR
    if (blockIdx.x == 0 && lane == 0)
      printf("prefix codes (complex form): %llu, %llu symbols; ticks: code-length code %llu, symbol lengths %llu, build_tree %llu (all tables: histogram %llu, first pass %llu, second level + root %llu)\n",
             g_hdr_prof[4], g_hdr_prof[5], g_hdr_prof[0], g_hdr_prof[1], g_hdr_prof[2], g_hdr_prof[3], g_hdr_prof[6], g_hdr_prof[7]);
    if (blockIdx.x == 0 && lane == 0)
      printf("block 0: %llu ticks before the stream, %llu in it: headers %llu, command loops %llu (x 256)\n", (unsigned long long)(hdr_prof_t1 - hdr_prof_t0),
             (unsigned long long)(__builtin_amdgcn_s_memtime() - hdr_prof_t1), (unsigned long long)s.prof[4], (unsigned long long)s.prof[5]);
#endif

    // result mapping of the one-shot driver (decode.rs:2829-2916, 3382-3397; lib.rs:447-468)
    const bool over = s.br.over();
    if (e != E_NEEDS_MORE_INPUT && e != E_BLOCK_LENGTH_1 && e != E_NEEDS_MORE_OUTPUT && e != E_RETRY_ARENA && e != E_PROBE && over) e = E_NEEDS_MORE_INPUT;
    uint64_t decoded;
    if (e == E_SUCCESS || e == E_NEEDS_MORE_INPUT) decoded = s.P;          // everything produced is flushed
    else if (e == E_NEEDS_MORE_OUTPUT) decoded = s.out_cap;
    else {
      // fatal: the caller has what was flushed at the last ring-buffer boundary
      decoded = s.rb_size ? s.next_boundary - s.rb_size : 0;
      if (decoded > s.P) decoded = 0;
    }
    if (lane == 0) {
      st->result = e == E_SUCCESS ? 1 : e == E_NEEDS_MORE_INPUT ? 2 : e == E_NEEDS_MORE_OUTPUT ? 3 : e == E_RETRY_ARENA ? BROTLI_AMD_RESULT_RETRY_ARENA : e == E_PROBE ? BROTLI_AMD_RESULT_PROBE : 0;
      st->error_code = e;
      st->decoded_size = decoded;
      uint64_t c = (s.br.pos() + 7) >> 3;
      if (e == E_NEEDS_MORE_INPUT || c > d.in_size) c = d.in_size;
      st->consumed = c;
      st->produced = s.P;
      st->num_metablocks = s.num_metablocks;
      st->num_commands = s.num_commands;
      st->engine_commands = s.engine_commands;
      st->spilled_metablocks = s.num_spilled;
      st->peak_trees = s.peak_trees; st->peak_map_bytes = s.peak_maps; st->ring_bytes = s.rb_size; st->any_compressed = s.any_compressed;
#ifdef BROTLI_AMD_PROFILE
      // debugging aid: cycle split of the command loop in the (otherwise unused) resume block of the status
      st->resume.bit_pos = __builtin_amdgcn_s_memtime() - prof_start;
      st->resume.out_pos = s.prof[0];
      st->resume.dist_rb[0] = (int32_t)(s.prof[1] >> 8); st->resume.dist_rb[1] = (int32_t)(s.prof[2] >> 8); st->resume.dist_rb[2] = (int32_t)(s.prof[3] >> 8);
      st->resume.dist_rb[3] = (int32_t)s.prof[4]; st->resume.dist_rb_idx = (int32_t)s.prof[5];
#endif
    }
  }
#ifdef BROTLI_AMD_PROFILE_LEAN
  if (blockIdx.x == 0 && lane_id() == 0)
    printf("lean cmds %llu wait before flush %llu after flush stores %llu (ticks per cmd); ticks per cmd: head %llu literals %llu distance %llu copy %llu\n", g_lean_prof[4],
           g_lean_prof[5] / g_lean_prof[4], g_lean_prof[6] / g_lean_prof[4], g_lean_prof[0] / g_lean_prof[4],
           g_lean_prof[1] / g_lean_prof[4], g_lean_prof[2] / g_lean_prof[4], g_lean_prof[3] / g_lean_prof[4]);
#endif
#ifdef BROTLI_AMD_PROFILE_SPLIT
  if (blockIdx.x == 0 && lane_id() == 0 && g_split_prof[10] != 0)
    printf("split: %llu parser invocations (%llu left at a distance the lean loop does not take), %llu commands, %llu records executed; parser ticks %llu: ring full %llu, drains %llu, "
           "context from the copy's source: %llu waits of %llu ticks; context after a drain: %llu, loads %llu ticks; copier busy ticks %llu; commands out of records %llu, looks at the records' frontier %llu; waves %u, record ring at %u (arena %u)\n",
           g_split_prof[10], g_split_prof[7], g_split_prof[11], g_split_prof[12], g_split_prof[4], g_split_prof[0], g_split_prof[1], g_split_prof[8], g_split_prof[2],
           g_split_prof[9], g_split_prof[3], g_split_prof[5], g_split_prof[15], g_split_prof[14], hc_ld(HC_NW_ALL), hc_ld(HC_EXT_BASE), lds_arena_bytes);
  if (blockIdx.x == 0 && lane_id() == 0 && g_split_prof[10] != 0)
    printf("split parser ticks per command: head %llu, literals %llu, distance %llu, next root + record request + checks %llu, post %llu, context fetch + tail %llu, loop top %llu\n",
           g_split_prof[16] / g_split_prof[11], g_split_prof[17] / g_split_prof[11], g_split_prof[18] / g_split_prof[11], g_split_prof[19] / g_split_prof[11],
           g_split_prof[20] / g_split_prof[11], g_split_prof[21] / g_split_prof[11], 0ull);
#endif
#ifdef BROTLI_AMD_PROFILE_SPLIT
  if (blockIdx.x == 0 && lane_id() == 0 && g_split_prof[37] != 0)
    printf("record loop: %llu calls, %llu commands (%llu in hand-written runs, %llu with literals, %llu without by the compiled path, %llu words), %llu ticks in the loop; ticks: top %llu, runs %llu, "
           "head + context bytes %llu, literals %llu, distance %llu, counts + literal store %llu, copy / word + next record %llu\n",
           g_split_prof[37], g_split_prof[36], g_split_prof[32], g_split_prof[33], g_split_prof[34], g_split_prof[35], g_split_prof[38],
           g_split_prof[24], g_split_prof[25], g_split_prof[26], g_split_prof[27], g_split_prof[28], g_split_prof[29], g_split_prof[30]);
  if (blockIdx.x == 0 && lane_id() == 0 && g_split_prof[39] != 0)
    printf("record loop: %llu ticks between the s_memtime pairs around the runs' waits for the memory pipe (an empty pair costs what tools/ubench says)\n", g_split_prof[39]);
#endif
#ifdef BROTLI_AMD_PROFILE_SPEC
  if (blockIdx.x == 0 && lane_id() == 0)
    printf("spec rounds %llu lits %llu bits %llu ticks: chunk0 %llu wait %llu resolve %llu move %llu seek %llu\n", g_spec_prof[5], g_spec_prof[6], g_spec_prof[7],
           g_spec_prof[0], g_spec_prof[1], g_spec_prof[2], g_spec_prof[3], g_spec_prof[4]);
#endif
#ifdef BROTLI_AMD_PROFILE_SCAN
  if (blockIdx.x == 0 && lane_id() == 0 && g_scan_prof[17] != 0)
    printf("scan engine: %llu invocations, %llu commands, %llu batches, %llu entries; wave 0 ticks: execute %llu S/J1 %llu J2-32 %llu own copies %llu walk %llu resolve %llu wait for REC %llu D2/D4 %llu barrier behind S/J1 %llu barriers of J2-32 %llu\n",
           g_scan_prof[17], g_scan_prof[16], g_scan_prof[13], g_scan_prof[12], g_scan_prof[6], g_scan_prof[1], g_scan_prof[2], g_scan_prof[8], g_scan_prof[4],
           g_scan_prof[5], g_scan_prof[3], g_scan_prof[7], g_scan_prof[9], g_scan_prof[10]);
  if (blockIdx.x == 0 && lane_id() == 0 && g_scan_prof[17] != 0)
    printf("kernel ticks of block 0: %llu; the engine's ticks: %llu steps, %llu that only walk, %llu last ones\n", (unsigned long long)(__builtin_amdgcn_s_memtime() - scan_prof_t0),
           g_scan_prof[14], g_scan_prof[15], g_scan_prof[11]);
  if (blockIdx.x == 0 && lane_id() == 0 && g_scan_prof[17] != 0) printf("wave 1's ticks in REC: %llu; the last wave's in copies that depend on the group's own output: %llu for %llu copies\n", g_scan_prof[24], g_scan_prof[25], g_scan_prof[26]);
  if (blockIdx.x == 0 && lane_id() == 0 && g_scan_prof[17] != 0)
    printf("scan engine exits: input ends %llu, counts/limits %llu, distance %llu, by-hand precheck %llu, long run %llu; invocations that took < 64 commands %llu\n",
           g_scan_prof[18], g_scan_prof[19], g_scan_prof[20], g_scan_prof[21], g_scan_prof[22], g_scan_prof[23]);
#endif
#ifdef BROTLI_AMD_PROFILE_WAVES
  if (blockIdx.x == 0 && lane_id() == 0)
    for (int k = 0; k < 96; k++) if (g_wave_prof[k][16] != 0) {
      printf("barrier %2d (%llu times): ticks of waves 0..15 in front of it:", k, g_wave_prof[k][16]);
      for (int w = 0; w < 16; w++) printf(" %llu", g_wave_prof[k][w] / g_wave_prof[k][16]);
      printf("\n");
    }
#endif
#ifdef BROTLI_AMD_PROFILE_SCAN
  if (blockIdx.x == 0 && lane_id() == 0 && g_path_prof[33] != 0) {
    printf("\npath engine: %llu invocations, %llu regions, %llu commands (%llu listed, %llu executed); wave 0 ticks per region: input %llu J1 %llu path %llu records %llu closure %llu next8 %llu walk %llu details %llu resolve %llu execute %llu own copies %llu\n",
           g_path_prof[33], g_path_prof[20], g_path_prof[32], g_path_prof[26], g_path_prof[28], g_path_prof[0] / g_path_prof[20], g_path_prof[1] / g_path_prof[20], g_path_prof[2] / g_path_prof[20],
           g_path_prof[3] / g_path_prof[20], g_path_prof[4] / g_path_prof[20], g_path_prof[5] / g_path_prof[20], g_path_prof[6] / g_path_prof[20], g_path_prof[7] / g_path_prof[20],
           g_path_prof[8] / g_path_prof[20], g_path_prof[9] / g_path_prof[20], g_path_prof[10] / g_path_prof[20]);
    printf("\npath engine per region: %llu path positions, %llu closure states in %llu.%llu rounds, %llu.%llu sync rounds, %llu anchors; states by hand %llu in all (%llu records met that said so); records that hit the hop cap %llu, the closure cap %llu\n",
           g_path_prof[22] / g_path_prof[20], g_path_prof[24] / g_path_prof[20], g_path_prof[23] / g_path_prof[20], (g_path_prof[23] * 10 / g_path_prof[20]) % 10,
           g_path_prof[21] / g_path_prof[20], (g_path_prof[21] * 10 / g_path_prof[20]) % 10, g_path_prof[27] / g_path_prof[20], g_path_prof[25], g_path_prof[31], g_path_prof[29], g_path_prof[30]);
    printf("\npath engine resolve alone: %llu per region (the figure called resolve above is then the lane-per-command stores behind it)\n", g_path_prof[11] / g_path_prof[20]);
    printf("\npath engine per region: %llu items that get a wave (long copies from in front of the region, long literal runs), %llu copies that read the region's own output\n", g_path_prof[12] / g_path_prof[20], g_path_prof[13] / g_path_prof[20]);
    printf("\npath engine dependent copies: %llu per region, the last wave's ticks in them %llu per region\n", g_path_prof[35] / g_path_prof[20], g_path_prof[34] / g_path_prof[20]);
    printf("\npath engine literal-run regions: %llu literals in all\n", g_path_prof[19]);
    printf("\npath engine path phase: chains and entries %llu, ranks %llu (the rest: positions and literals)\n", g_path_prof[17] / g_path_prof[20], g_path_prof[18] / g_path_prof[20]);
    printf("\npath engine, two engines (engine 0's wave 0, ticks per region): waiting for the window %llu, for the stream %llu, for the region before's output %llu\n", g_path_prof[15] / g_path_prof[20], g_path_prof[16] / g_path_prof[20], g_path_prof[14] / g_path_prof[20]);
    printf("\nbetween two invocations of the engine in one metablock: %llu ticks in all, %llu times; inside the engine's calls %llu; from the command loop's start to the first invocation %llu; from the last one's end to the loop's end %llu; the command loops in all %llu\n", g_path_prof[36], g_path_prof[37], g_path_prof[38], g_path_prof[29], g_path_prof[39], g_path_prof[31]);
    { unsigned long long eng = 0; for (int k = 0; k <= 11; k++) eng += g_path_prof[k]; eng += g_path_prof[14] + g_path_prof[15] + g_path_prof[16]; eng += g_path_prof[17] + g_path_prof[18];
      printf("\nkernel ticks of block 0 in this launch: %llu; the path engine's regions so far (all launches): %llu\n", (unsigned long long)(__builtin_amdgcn_s_memtime() - scan_prof_t0), eng); }
  }
#endif
  if (gang_m > 1u && lane == 0) gang_st32(gang_ctl(), GC_EPOCH, GC_QUIT);   // the stream is done: the gang's helper blocks may go
#ifdef BROTLI_AMD_GANG_KERNEL
  if (pool && queue[6] == 0u) {
    // no stream of its own (any more): this block helps -- the largest stream that is being decoded, is large enough to have something to divide
    // and has fewer than seven helpers; when that one is done, the next; until the pool has no stream left
    for (;;) {
      if (__hip_atomic_load(queue + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
      unsigned long long best = 0ull;   // size << 16 | stream + 1
      for (uint32_t s_ = lane; s_ < n_streams; s_ += 64u) {
        gu8* const c_ = (gu8*)(uintptr_t)(pool_base + (uint64_t)s_ * GC_STRIDE);
        const uint32_t ep_ = gang_ld32(c_, GC_EPOCH), jn_ = gang_ld32(c_, GC_JOINED);
        const uint64_t sz_ = descs[s_].in_size;
        if (ep_ != GC_QUIT && jn_ < 7u && sz_ >= 65536ull && (descs[s_].flags & BROTLI_AMD_FLAG_DEFER) == 0u) {
          const unsigned long long key_ = ((sz_ < (1ull << 40) ? sz_ : (1ull << 40) - 1ull) << 16) | (unsigned long long)(s_ + 1u);
          best = key_ > best ? key_ : best;
        }
      }
      for (int off_ = 32; off_ > 0; off_ >>= 1) { const unsigned long long o_ = __shfl_xor(best, off_); best = o_ > best ? o_ : best; }
      best = (unsigned long long)rfl((uint32_t)best) | ((unsigned long long)rfl((uint32_t)(best >> 32)) << 32);
      if (best == 0ull) break;   // (nobody to join, and there never will be: streams only end and helpers only join -- no block keeps looking while the last streams are decoded)
      const uint32_t s_ = (uint32_t)(best & 0xFFFFull) - 1u;
      gu8* const c_ = (gu8*)(uintptr_t)(pool_base + (uint64_t)s_ * GC_STRIDE);
      const uint32_t last_ = gang_ld32(c_, GC_EPOCH);   // (read in front of the join: an invocation the owner starts behind the join is this block's too)
      if (last_ == GC_QUIT) continue;
      uint32_t j_ = 0;
      if (lane == 0) j_ = gang_add32(c_, GC_JOINED, 1u);
      j_ = rfl(j_);
      if (j_ >= 7u) continue;   // (somebody else was faster: the count stays one too high, which the owner's cap of seven does not mind)
      hc_st(HC_GANG_LO, (uint32_t)(uintptr_t)c_); hc_st(HC_GANG_HI, (uint32_t)((uint64_t)(uintptr_t)c_ >> 32)); hc_st(HC_GANG_ROLE, j_ + 1u); hc_st(HC_GANG_M, 0x108u);
      hc_st(HC_GANG_EPOCH, last_);
      hc_st(HC_KIND, (uint32_t)HK_PATHR);
      lds_release();
      hc_st(HC_SEQ, hc_ld(HC_SEQ) + 1u);   // the other waves of the block join (helper_wave)
      (void)pe16r::path_engine(0);          // ... and stay until the stream is done
    }
  }
#endif
  // no more streams: the helper waves may go
  hc_st(HC_KIND, 2);
  lds_release();
  hc_st(HC_SEQ, hc_ld(HC_SEQ) + 1u);
}

extern "C" uint32_t brotli_amd_lds_helper_bytes(uint32_t waves);
extern "C" hipError_t BROTLI_AMD_LAUNCH(const BrotliAmdStreamDesc* descs, BrotliAmdStreamStatus* status, uint32_t n_streams,
                                        uint32_t* queue, uint8_t* scratch, uint64_t scratch_per_block, uint32_t grid,
                                        uint32_t lds_arena_bytes, const uint8_t* dict, hipStream_t stream, int helper_waves) {
  if (n_streams == 0) return hipSuccess;
  static const bool no_helpers = getenv("BROTLI_AMD_NO_HELPERS") != nullptr;  // (experiments: one wave per block)
  // (sixteen waves: a block with the command engine; otherwise at most eight)
  const uint32_t waves = no_helpers || helper_waves < 2 ? 1u : (uint32_t)helper_waves >= SC_WAVES ? SC_WAVES : (uint32_t)helper_waves > 8u ? 8u : (uint32_t)helper_waves;
  size_t smem = (size_t)LDS_FIXED + lds_arena_bytes + brotli_amd_lds_helper_bytes(waves);
  {  // BROTLI_AMD_ENGINE=scan: the round-2 command engine only (A/B tests); the symbol is per device
    static const char* const eng = getenv("BROTLI_AMD_ENGINE");
    if (eng != nullptr) {
      // bit 0: the scan engine only; bit 1: no copier wave for context-modelled metablocks (the default: it does not pay, see DESIGN;
      // "split" turns it on); bit 2: no command records ("norec")
      // bit 3: the path engine as two engines of eight waves that take the stream's regions in turns ("path2"; experiment)
      // bit 4: tables beyond the LDS part are read where they lie instead of the trees in use being cached ("nocache": A/B)
      // bit 5: NO command records for metablocks without context ("norecall": A/B -- round 6 gave such metablocks the records and the hand-written run in blocks with helper waves)
      // (static storage: the asynchronous copy reads it after this function has returned)
      static const uint32_t mode = strcmp(eng, "scan") == 0 ? 3u : strcmp(eng, "split") == 0 ? 0u : strcmp(eng, "norec") == 0 ? 6u : strcmp(eng, "path2") == 0 ? 10u : strcmp(eng, "nocache") == 0 ? 18u : strcmp(eng, "norecall") == 0 ? 34u : 2u;
      hipError_t e2 = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_engine_mode), &mode, sizeof mode, 0, hipMemcpyHostToDevice, stream);
      if (e2 != hipSuccess) return e2;
    }
  }
  // The attribute belongs to the function, not to the launch: callers on several threads use different block shapes, so setting it
  // and launching is one critical section (the launch itself is asynchronous).
  static std::mutex launch_mutex;
  std::lock_guard<std::mutex> lock(launch_mutex);
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(BROTLI_AMD_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (err != hipSuccess) return err;
  // Blocks of one wave where the caller wants more than four blocks per CU in flight (a CU's registers hold four
  // four-wave blocks): such batches gain more from streams in flight than from helper waves in long literal runs.
  hipLaunchKernelGGL(BROTLI_AMD_KERNEL, dim3(grid), dim3(64u * waves), smem, stream, descs, status, n_streams, queue, scratch,
                     scratch_per_block, lds_arena_bytes, dict);
  return hipGetLastError();
}

#ifndef BROTLI_AMD_GANG_KERNEL   // (what follows exists once: in the object without the gang kernel)
// Test hook (BrotliAmdDebugBuildTree, include/brotli/batch.h): the table builder alone.  One wave builds the two-level table of
// `n_sym` code lengths (src/huffman/mod.rs:273-386) where a metablock's first table would lie and then decodes every fifteen-bit
// value through it as read_symbol does: decoded[v] = symbol << 4 | code length.  The test compares that -- not the table's layout,
// which is this kernel's own -- with the known-answer tables of src/huffman/tests.rs.
extern "C" __global__ __launch_bounds__(64) void brotli_amd_debug_build_tree_kernel(const uint8_t* __restrict__ lengths, uint32_t n_sym, uint16_t* __restrict__ decoded,
                                                                                 uint32_t* __restrict__ entries, uint32_t lds_arena_bytes) {
  const uint32_t lane = lane_id();
  if (rfl((uint32_t)(uintptr_t)g_dynamic_lds) != 0u) { if (lane == 0) *entries = 0u; return; }
  for (uint32_t i = lane; i < MAX_ALPHABET; i += 64) lds_st8(LDS_LENGTHS + i, i < n_sym ? lengths[i] : 0u);
  lds_sync();
  Arena a; a.glb = nullptr; a.lds_limit = lds_arena_bytes; a.top = 0; a.cold = 0;
  const uint32_t tree = a.alloc(max_table_entries(n_sym) * 2u);
  const uint32_t size = build_tree(a, tree, n_sym);
  lds_sync();
  for (uint32_t v = lane; v < 32768u; v += 64u) {
    uint32_t e = lds_ld16(LDS_FIXED + tree + ((v & 0xFFu) << 1)), len = e & 15u;
    if (len > ROOT_BITS) {
      e = lds_ld16(LDS_FIXED + tree + (((e >> 4) + ((v >> ROOT_BITS) & mask_bits(len - ROOT_BITS))) << 1));
      len = ROOT_BITS + (e & 15u);
    }
    decoded[v] = (uint16_t)(((e >> 4) << 4) | len);
  }
  if (lane == 0) *entries = size;
}
extern "C" hipError_t brotli_amd_launch_debug_build_tree(const uint8_t* d_lengths, uint32_t n_sym, uint16_t* d_decoded, uint32_t* d_entries, hipStream_t stream) {
  const uint32_t arena = 8192u;
  hipLaunchKernelGGL(brotli_amd_debug_build_tree_kernel, dim3(1), dim3(64), (size_t)LDS_FIXED + arena, stream, d_lengths, n_sym, d_decoded, d_entries, arena);
  return hipGetLastError();
}

extern "C" uint32_t brotli_amd_lds_fixed_bytes(void) { return LDS_FIXED; }
extern "C" uint32_t brotli_amd_lds_helper_bytes(uint32_t waves) {
  if (waves >= SC_WAVES) return SC_BYTES;  // (the rounds' slots lie inside the engine's rings)
  return waves > 1u ? waves * HL_SLOT : 0u;
}
#endif
