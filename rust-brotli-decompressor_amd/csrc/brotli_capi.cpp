// brotli_capi.cpp -- host side of libbrotli_decompressor.so: the reference's C ABI (include/brotli/decode.h)
// and the batch extension (include/brotli/batch.h) on top of the HIP decode kernel.
//
// Mirrors, by behaviour, reference src/ffi/mod.rs (entry points, argument validation, error latching),
// src/lib.rs:336-468 (one-shot helpers, BrotliDecoderReturnInfo) and the caller-visible contract of
// src/decode.rs:2779-3403 (BrotliDecompressStream: what is consumed, what is delivered, when each result is
// returned).  It owns no decoder: all decoding happens in brotli_kernels.hip.  The streaming entry point keeps
// the stream's compressed bytes and its output in device memory and re-launches the kernel from the last
// completed metablock boundary each time more input arrives (BrotliAmdResume), which is the device analogue
// of the reference's resumable state machine.  When no HIP device is usable every entry point fails with
// BROTLI_DECODER_ERROR_UNREACHABLE and a message -- there is no CPU path to fall back to.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <chrono>
#include <vector>

#include "brotli/batch.h"
#include "brotli/decode.h"
#include "brotli_device_abi.h"

extern "C" hipError_t brotli_amd_launch_decode(const BrotliAmdStreamDesc* descs, BrotliAmdStreamStatus* status, uint32_t n_streams,
                                               uint32_t* queue, uint8_t* scratch, uint64_t scratch_per_block, uint32_t grid,
                                               uint32_t lds_arena_bytes, const uint8_t* dict, hipStream_t stream, int waves_per_block);
// (the same kernel with the gang's form of the path engine in it: the launches that give every stream a gang of blocks -- csrc/brotli_kernels.hip, BROTLI_AMD_GANG_KERNEL)
extern "C" hipError_t brotli_amd_launch_decode_gang(const BrotliAmdStreamDesc* descs, BrotliAmdStreamStatus* status, uint32_t n_streams,
                                                    uint32_t* queue, uint8_t* scratch, uint64_t scratch_per_block, uint32_t grid,
                                                    uint32_t lds_arena_bytes, const uint8_t* dict, hipStream_t stream, int waves_per_block);
extern "C" uint32_t brotli_amd_lds_fixed_bytes(void);
extern "C" uint32_t brotli_amd_lds_helper_bytes(uint32_t waves);
extern "C" const uint8_t brotli_amd_dictionary[];  // dict_blob.c: data/dictionary.bin, 122784 bytes

namespace {

constexpr size_t kDictSize = 122784;
// One-wave blocks per CU at most (a CU's registers hold sixteen waves of the kernel) and the smallest table arena worth a
// first pass; what does not fit a pass's arena comes back in the next (retry_with_larger_arenas).
static const size_t kMaxBlocksPerCu = getenv("BROTLI_AMD_MAX_BLOCKS_PER_CU") ? (size_t)atoi(getenv("BROTLI_AMD_MAX_BLOCKS_PER_CU")) : 14;
static const uint32_t kMinSmallArena = getenv("BROTLI_AMD_MIN_SMALL_ARENA") ? (uint32_t)atoi(getenv("BROTLI_AMD_MIN_SMALL_ARENA")) : 3584u;  // (16 blocks per CU: 4016 bytes)
constexpr uint64_t kScratchPerBlock = (2u << 20) + BROTLI_AMD_SPEC_SCRATCH;  // worst-case table arena of one metablock + helper scratch
constexpr uint32_t kDefaultLdsPerBlock = 36 * 1024;

thread_local std::string g_last_error;
thread_local std::string g_last_note;   // what a call did differently without failing (engine blocks refused by the device: see launch())
// Blocks of sixteen waves with the command engine (csrc/brotli_scan_engine.h) for batches of at most one stream per CU;
// BROTLI_AMD_NO_SCAN=1 keeps the launch shapes without it (experiments, A/B measurements).
static const bool g_engine_wanted = getenv("BROTLI_AMD_NO_SCAN") == nullptr;  // (whether a device can hold such a block is decided per batch context, at its creation)
constexpr size_t kGang16MinBytes = (size_t)2 << 20;   // compressed bytes of a batch's largest stream from which a gang is sixteen blocks (plan_gangs)
constexpr uint32_t kGangPool = BROTLI_AMD_GANG_POOL_FLAG | 8u;   // plan_gangs' word for a pool launch (queue[2])
constexpr uint64_t kProbeMinMeanBytes = 8192;     // mean compressed size of a batch from which the device is asked what kind its streams are (submit())
constexpr uint32_t kEngineQueueMaxPerCu = 4;      // streams per CU up to which blocks of sixteen waves, one a CU, take a batch's streams one after the other -- where the
                                                  // DEVICE says they are a command engine's kind (probe_streams); beyond, streams in flight beat the engine (2048 x 1 MiB of the
                                                  // metric's make-up: 220 GB/s eight to a CU in one-wave blocks, 151 through engine blocks)
constexpr uint32_t kScanArena = 40960;  // table arena of such a block (with the engine's rings: about 108 KiB of LDS)

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

// ---- per-device constant data (the static dictionary) ----
std::mutex g_dict_mutex;
std::vector<uint8_t*> g_dict_by_device;

const uint8_t* device_dictionary(int dev) {
  std::lock_guard<std::mutex> lock(g_dict_mutex);
  if ((int)g_dict_by_device.size() <= dev) g_dict_by_device.resize(dev + 1, nullptr);
  if (!g_dict_by_device[dev]) {
    uint8_t* p = nullptr;
    if (!hip_ok(hipMalloc(&p, kDictSize + 64), "hipMalloc(dictionary)")) return nullptr;
    if (!hip_ok(hipMemcpy(p, brotli_amd_dictionary, kDictSize, hipMemcpyHostToDevice), "hipMemcpy(dictionary)")) { (void)hipFree(p); return nullptr; }
    g_dict_by_device[dev] = p;
  }
  return g_dict_by_device[dev];
}

// Every entry point leaves the caller's current HIP device as it found it.
struct DeviceGuard {
  int saved = -1;
  DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
  ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

bool current_device(int* dev) {
  int count = 0;
  if (!hip_ok(hipGetDeviceCount(&count), "hipGetDeviceCount")) return false;
  if (count <= 0) { g_last_error = "no HIP device present"; return false; }
  return hip_ok(hipGetDevice(dev), "hipGetDevice");
}

}  // namespace

// ================================================ batch ================================================
struct BrotliAmdBatch {
  int device = 0;
  uint32_t max_streams = 0, lds_arena = 0, grid_max = 0;
  uint32_t n = 0, grid = 0;
  BrotliAmdStreamDesc* d_descs = nullptr;
  BrotliAmdStreamStatus* d_status = nullptr;
  uint32_t* d_queue = nullptr;
  uint8_t* d_scratch = nullptr;
  uint64_t scratch_blocks = 0;
  BrotliAmdStreamDesc* h_descs = nullptr;    // pinned
  BrotliAmdStreamStatus* h_status = nullptr;  // pinned
  uint32_t* h_order = nullptr;                 // pinned: queue header + the order in which blocks take the streams
  bool ordered = false;
  const uint8_t* d_dict = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;  // around the first launch; around a launch of a later pass
  float retry_ms = 0.0f;  // kernel time of the later passes of the last job
  hipStream_t last_stream = nullptr;
  bool launched = false;
  // first-pass arena of this launch: the configured one, or a smaller one when the batch has more streams than the
  // device can hold blocks of the configured size (more waves in flight; what does not fit goes to the second pass)
  bool auto_arena = false;
  uint32_t per_cu_cap = 0;  // blocks per CU a first pass may ask for (lowered when most of a batch had to come back)
  uint32_t cur_per_cu = 0;  // blocks per CU the first pass of this launch was shaped for
  uint32_t cur_arena = 0, cus = 0, lds_fixed = 0;
  size_t lds_per_cu = 0;
  // second pass for streams whose tables did not fit the LDS arena of the first (BROTLI_AMD_FLAG_NO_SPILL)
  uint32_t max_arena = 0, retry_grid_max = 0, last_retry_count = 0, lds_helper = 0, lds_helper8 = 0, waves = 4, block_max = 0;
  BrotliAmdStreamDesc* d_retry_descs = nullptr;
  BrotliAmdStreamStatus* d_retry_status = nullptr;
  BrotliAmdStreamDesc* h_retry_descs = nullptr;    // pinned
  BrotliAmdStreamStatus* h_retry_status = nullptr;  // pinned
  // staging for BrotliAmdBatchDecodeHost
  uint8_t* d_stage_in = nullptr; size_t stage_in_cap = 0;
  uint8_t* d_stage_out = nullptr; size_t stage_out_cap = 0;
  // ... and its pinned host side: the caller's buffers are pageable as a rule, a copy engine wants pinned memory (one transfer per
  // direction in pieces, the host's own copies on several threads side by side with the transfers)
  uint8_t* h_pin_in = nullptr; size_t pin_in_cap = 0;
  uint8_t* h_pin_out = nullptr; size_t pin_out_cap = 0;
  hipStream_t copy_stream = nullptr;
  // streams that ran out of output get the reference's verdict (settle_output_limits): set by the batch entry points
  bool exact_limit = false;
  // blocks of sixteen waves with a command engine: the device's LDS holds one (decided at creation), nothing has refused one since
  bool engine_ok = false;
  uint8_t* d_settle = nullptr; size_t settle_cap = 0; uint32_t last_settle_count = 0;
  // several CUs on one stream: the blocks of a gang in this launch (0: none) and the gangs' control blocks (brotli_device_abi.h)
  uint32_t gang = 0, last_gang = 0;
  uint8_t* d_gang = nullptr; size_t gang_cap = 0;
  // the probe's answers for the batch it was asked about (probe_streams): the same descriptors again are not probed again
  std::vector<uint8_t> probe_kind; uint64_t probe_key = 0; float last_probe_ms = 0.0f;
};

namespace {

bool ensure_scratch(BrotliAmdBatch* b, uint32_t grid) {
  if (b->scratch_blocks >= grid) return true;
  if (b->d_scratch) (void)hipFree(b->d_scratch);
  b->d_scratch = nullptr; b->scratch_blocks = 0;
  if (!hip_ok(hipMalloc(&b->d_scratch, (size_t)grid * kScratchPerBlock), "hipMalloc(table scratch)")) return false;
  b->scratch_blocks = grid;
  return true;
}

int launch(BrotliAmdBatch* b, hipStream_t stream) {
  // queue header (pull counter, order flag) and, for batches of more streams than blocks, the order
  b->h_order[0] = 0; b->h_order[1] = b->ordered ? 1u : 0u;
  for (int i = 2; i < 16; i++) b->h_order[i] = 0;
  if (b->gang > 1u) {
    const size_t need = (size_t)b->n * BROTLI_AMD_GANG_CTL_BYTES;
    if (b->gang_cap < need) {
      if (b->d_gang) (void)hipFree(b->d_gang);
      b->d_gang = nullptr; b->gang_cap = 0;
      if (!hip_ok(hipMalloc(&b->d_gang, need), "hipMalloc(gang control)")) return -1;
      b->gang_cap = need;
    }
    if (!hip_ok(hipMemsetAsync(b->d_gang, 0, need, stream), "hipMemsetAsync(gang control)")) return -1;
    b->h_order[2] = b->gang; b->h_order[4] = (uint32_t)(uintptr_t)b->d_gang; b->h_order[5] = (uint32_t)((uint64_t)(uintptr_t)b->d_gang >> 32);
    b->h_order[6] = getenv("BROTLI_AMD_GANG_NO_HELPERS") != nullptr ? 1u : 0u;
    b->h_order[8] = b->n;   // (a pool: the streams that are not done yet)   // (tests: the helper blocks leave at once, the owners must find out and go on alone)
  }
  b->last_gang = b->gang;
  if (!hip_ok(hipMemcpyAsync(b->d_queue, b->h_order, sizeof(uint32_t) * (b->ordered ? 16 + (size_t)b->n : 16), hipMemcpyHostToDevice, stream), "hipMemcpyAsync(queue)")) return -1;
  if (!hip_ok(hipEventRecord(b->ev0, stream), "hipEventRecord")) return -1;
  static const bool gang_kernel_always = getenv("BROTLI_AMD_GANG_KERNEL_ALWAYS") != nullptr;   // (experiments: what the second kernel costs launches without gangs)
  hipError_t le = (b->gang > 1u || gang_kernel_always ? brotli_amd_launch_decode_gang : brotli_amd_launch_decode)(b->d_descs, b->d_status, b->n, b->d_queue, b->d_scratch, kScratchPerBlock, b->grid, b->cur_arena,
                                                                                             b->d_dict, stream, (int)b->waves);
  if (b->waves == 16u && (le == hipErrorInvalidValue || le == hipErrorLaunchOutOfResources || le == hipErrorSharedObjectInitFailed || le == hipErrorInvalidConfiguration)) {
    // the device refused a block of sixteen waves with the engine's LDS although its properties allow one: this context goes
    // on with blocks of eight waves, and says so (BrotliAmdLastNote); streams are no longer sent back for engine blocks
    (void)hipGetLastError();
    g_last_note = std::string("engine blocks refused (") + hipGetErrorString(le) + "): eight-wave blocks from now on";   // (a note, not an error: the retry below decides)
    b->engine_ok = false;
    b->waves = 8;
    if (b->gang > 1u) {   // (the gangs' helper blocks go with the engine blocks)
      b->gang = 0; b->last_gang = 0; b->grid = std::min(b->n, b->grid);
      b->h_order[2] = 0;
      if (!hip_ok(hipMemcpyAsync(b->d_queue, b->h_order, sizeof(uint32_t) * 16, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(queue)")) return -1;
    }
    for (uint32_t i = 0; i < b->n; i++) b->h_descs[i].flags &= ~(BROTLI_AMD_FLAG_ENGINE_ONLY | BROTLI_AMD_FLAG_DEFER);
    if (!hip_ok(hipMemcpyAsync(b->d_descs, b->h_descs, sizeof(BrotliAmdStreamDesc) * b->n, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(descs)")) return -1;
    le = brotli_amd_launch_decode(b->d_descs, b->d_status, b->n, b->d_queue, b->d_scratch, kScratchPerBlock, b->grid, b->cur_arena, b->d_dict, stream, 8);
  }
  if (!hip_ok(le, "brotli_amd_decode_kernel launch")) return -1;
  if (!hip_ok(hipEventRecord(b->ev1, stream), "hipEventRecord")) return -1;
  b->last_stream = stream;
  b->launched = true;
  return 0;
}

// Table arena of one-wave blocks packed per_cu to a CU (0: too small to be worth a pass).
uint32_t small_arena(const BrotliAmdBatch* b, uint32_t per_cu) {
  const uint32_t per_block = (uint32_t)(b->lds_per_cu / per_cu) & ~255u;
  return per_block > b->lds_fixed + kMinSmallArena ? (per_block - b->lds_fixed) & ~15u : 0u;
}

// What kind of stream is each of the batch's?  A launch of the shape at hand in which nothing is decoded: every stream's header is read up
// to the literal context map of its first compressed metablock (BROTLI_AMD_FLAG_PROBE) -- where that says 'an engine's kind', on through its literal codes
// and its first command code.  kind[i]: bit 0 there is such a metablock, bit 1 its literals do not depend on context, bit 2 it is large enough for a
// command engine, bit 3 (round 6) its commands are SHORT -- text: the engines' kind by the first three, and yet four such streams a CU on a wave each with
// the command records (lean_rec_commands) do 2.4 times what an engine block does with them one after the other: they are not sent to engine blocks.  (Round 4 guessed from the batch's size and its
// mean compressed size: 1024 x 1 MiB of engine-shaped streams went through one-wave blocks -- 129 GB/s where engine blocks do 148 --, and
// could not be told from 1024 context-modelled texts, which engine blocks take at half speed.  The probe costs a launch of some tens of
// microseconds and reads the facts.)
int probe_streams(BrotliAmdBatch* b, uint32_t n, hipStream_t stream, std::vector<uint8_t>& kind) {
  kind.assign(n, 0);
  if (!ensure_scratch(b, b->grid)) return -1;
  for (uint32_t i = 0; i < n; i++) b->h_descs[i].flags |= BROTLI_AMD_FLAG_PROBE;
  bool ok = hip_ok(hipMemcpyAsync(b->d_descs, b->h_descs, sizeof(BrotliAmdStreamDesc) * n, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(descs)");
  ok = ok && hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize(probe descs)");   // (pinned memory: the copy reads it when it runs, not when it is asked for)
  for (uint32_t i = 0; i < n; i++) b->h_descs[i].flags &= ~BROTLI_AMD_FLAG_PROBE;
  ok = ok && hip_ok(hipMemsetAsync(b->d_queue, 0, sizeof(uint32_t) * 16, stream), "hipMemsetAsync(queue)");
  ok = ok && hip_ok(brotli_amd_launch_decode(b->d_descs, b->d_status, n, b->d_queue, b->d_scratch, kScratchPerBlock, b->grid, b->cur_arena, b->d_dict, stream, (int)b->waves),
                    "brotli_amd_decode_kernel launch (probe)");
  ok = ok && hip_ok(hipMemcpyAsync(b->h_status, b->d_status, sizeof(BrotliAmdStreamStatus) * n, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(status)");
  ok = ok && hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize(probe)");
  if (!ok) return -1;
  for (uint32_t i = 0; i < n; i++) if (b->h_status[i].result == BROTLI_AMD_RESULT_PROBE) kind[i] = (uint8_t)(b->h_status[i].engine_commands & 15u);
  return 0;
}

// Several blocks on a stream (csrc/brotli_path_engine.h, PE_CFG_REMOTE): what a launch of sixteen-wave blocks, one a stream, gets on top.
// Returns 0 (nothing), 2 / 4 / 8 / 16 (GANGS: that many blocks a stream, dealt at the launch -- its owner and one, three or seven helper blocks that take
// the path engine's regions in turns with it; eight streams' gangs side by side, a gang's members eight block numbers apart: one XCD; streams beyond a
// multiple of eight leave their gangs' blocks without work) or 0x108 (a POOL: as many blocks as CUs; a block without a stream of its own -- at once where
// there are fewer streams than CUs, else when its stream is done -- joins the largest stream still being decoded), and the launch's blocks in *grid.
//   * Not for batches of small streams: a gang has something to divide from a dozen regions on -- 64 KiB of compressed data --, and costs a launch ten
//     microseconds (its blocks' start, the control blocks' zeroing, the helpers' last look at the word that lets them go).
//   * Gangs of eight up to an eighth of the CUs' streams, of four up to a quarter, of two up to half; of SIXTEEN up to a sixteenth where a stream is long
//     (kGang16MinBytes compressed: eight blocks on one long stream are busy building and consuming, not waiting -- one 64 MiB stream 26.6 -> 24.3 ms,
//     one of 1 GiB 387 -> 356 ms; streams of the metric's 4 MiB gain nothing: their invocations are a dozen regions).
//   * A pool where the sizes differ -- the largest more than twice the median, and a long pole worth it: 256 KiB compressed, a millisecond and more
//     alone -- and the gangs would be of four or two blocks or none: the long one gets seven helpers (one 64 MiB stream among 39 / 99 / 199 of 1 MiB:
//     43.5 / 76.7 / 127.6 -> 29 ms).  Not where the streams are of a size: they end within a few per cent of each other, and the control blocks'
//     zeroing and the owners' looks at them cost what the last invocations' help brings (a pool forced on 192 x 4 MiB: +1 %, on 250 x 4 MiB: -4 %).
// gang_env: BROTLI_AMD_GANG (-1 unset; 0, 1: nothing at all; 2, 4, 8: gangs of at most that many, no pool); pool_env: BROTLI_AMD_POOL (-1 unset; 0: no
// pool; 2: a pool whatever the sizes where there would be no gangs).  A pure function of its arguments: BrotliAmdDebugPlanGangs, tests/test_host_logic.
uint32_t plan_gangs(uint32_t n, uint32_t cus, const size_t* in_sizes, int gang_env, int pool_env, uint32_t* grid) {
  if (n == 0u || n > cus || gang_env == 0 || gang_env == 1) return 0u;
  size_t largest_in = 0;
  for (uint32_t i = 0; i < n; i++) largest_in = std::max<size_t>(largest_in, in_sizes[i]);
  if (largest_in < 65536u) return 0u;
  uint32_t gang = 0u;
  const uint32_t groups = (n + 7u) / 8u;
  uint32_t m = groups * 64u <= cus ? 8u : groups * 32u <= cus ? 4u : groups * 16u <= cus ? 2u : 0u;
  // (round 6) sixteen blocks a stream where the device has them and a stream is long enough to keep them busy -- 2 MiB compressed, a few hundred regions:
  // eight blocks on one long stream are BUSY (96 % of the launch building their windows' tables and taking their regions through), not waiting for one another
  if (m == 8u && groups * 128u <= cus && largest_in >= kGang16MinBytes) m = 16u;
  if (gang_env > 1 && m > (uint32_t)gang_env) m = gang_env >= 16 ? 16u : gang_env >= 8 ? 8u : gang_env >= 4 ? 4u : 2u;
  if (gang_env == 16 && groups * 128u <= cus) m = 16u;   // (experiments: sixteen whatever the sizes)
  if (m > 1u) { gang = m; *grid = groups * 8u * m; }
  if (m < 8u && pool_env != 0 && gang_env < 0) {
    std::vector<size_t> sz(in_sizes, in_sizes + n);
    std::nth_element(sz.begin(), sz.begin() + n / 2, sz.end());
    if ((largest_in > 2u * sz[n / 2] && largest_in >= (256u << 10)) || (pool_env == 2 && m == 0u)) { gang = kGangPool; *grid = cus; }
  }
  return gang;
}

int submit(BrotliAmdBatch* b, uint32_t n, hipStream_t stream) {  // h_descs[0..n) filled
  if (n == 0) { b->n = 0; b->launched = false; return 0; }
  if (!hip_ok(hipSetDevice(b->device), "hipSetDevice")) return -1;
  // arena of this launch (see cur_arena)
  b->cur_arena = b->lds_arena;
  uint32_t grid_max = b->grid_max;
  b->cur_per_cu = 0;
  if (b->auto_arena && n > b->grid_max && b->max_arena > b->lds_arena) {
    const uint32_t per_cu = (uint32_t)std::min<size_t>(b->per_cu_cap, ((size_t)n + b->cus - 1) / b->cus);  // blocks per CU wanted
    const uint32_t arena = per_cu > 4u ? small_arena(b, per_cu) : 0u;
    if (arena != 0 && arena < b->lds_arena) { b->cur_arena = arena; grid_max = b->cus * per_cu; b->cur_per_cu = per_cu; }
  }
  b->n = n;
  b->grid = std::min(n, grid_max);
  // Waves per block: one decoding wave plus helpers for long literal runs.  A CU's registers hold sixteen waves of this
  // kernel: eight-wave blocks where at most two blocks per CU are wanted, four-wave blocks up to four, one-wave blocks
  // beyond (streams in flight are worth more than helpers there).
  b->waves = b->grid > 4u * b->cus ? 1u : 4u;
  if (b->grid <= 2u * b->cus) {
    const uint32_t room = (uint32_t)std::min<size_t>(b->block_max, b->lds_per_cu / 2);
    if (b->auto_arena && room > b->lds_fixed + b->lds_helper8 + b->lds_arena) { b->cur_arena = (room - b->lds_fixed - b->lds_helper8) & ~15u; b->waves = 8; }
    else if (b->lds_fixed + b->lds_helper8 + b->cur_arena <= room) b->waves = 8;
  }
  // Up to three large streams per CU: sixteen-wave blocks, one per CU, take them one after the other (the command engine
  // decodes a stream 3.5x faster than one wave does; measured on 384 / 512 x 4 MiB of the metric's data: 61 / 81 GB/s
  // against 33 / 44 with two eight-wave blocks per CU, while 1024 streams are faster four to a CU).  Metablocks the
  // engine cannot take go back and continue in a launch of small blocks (BROTLI_AMD_FLAG_ENGINE_ONLY).
  static const bool no_wide = getenv("BROTLI_AMD_NO_ENGINE_QUEUE") != nullptr;  // (experiments)
  // (whether a block of sixteen waves fits is settled first: only then is the grid cut down to one block per CU)
  uint32_t arena16 = 0;
  bool can16 = false;
  if (b->engine_ok) {
    const uint32_t h16 = brotli_amd_lds_helper_bytes(16);
    const size_t room = b->lds_per_cu > (size_t)b->lds_fixed + h16 ? b->lds_per_cu - b->lds_fixed - h16 : 0;
    arena16 = b->auto_arena ? (uint32_t)std::min<size_t>(kScanArena, room & ~(size_t)15) : b->cur_arena;
    can16 = arena16 <= room && (!b->auto_arena || arena16 >= 16384u);
  }
  bool engine_queue = false;
  static const uint32_t queue_max = getenv("BROTLI_AMD_ENGINE_QUEUE_MAX") ? (uint32_t)atoi(getenv("BROTLI_AMD_ENGINE_QUEUE_MAX")) : kEngineQueueMaxPerCu;  // (streams per CU)
  std::vector<uint8_t> kind;
  for (uint32_t i = 0; i < n; i++) b->h_descs[i].flags &= ~(BROTLI_AMD_FLAG_ENGINE_ONLY | BROTLI_AMD_FLAG_DEFER);
  b->last_probe_ms = 0.0f;
  // (round 6) ... and beyond that many: one-wave blocks, fourteen a CU -- unless the streams are the RECORD LOOP's: context-modelled ones and text (the probe's
  // kinds 5 and 15), which four-wave blocks, four a CU, taking the streams off the queue one after the other, decode half as fast again as fourteen
  // one-wave blocks a CU do (4096 x alice29: 13.2 -> 19+ GB/s; 4096 x lcet10 at -q 5: 16.6 -> 24+): the same probe says which
  const bool few = n <= queue_max * b->cus;
  static const bool no_record_blocks = getenv("BROTLI_AMD_NO_RECORD_BLOCKS") != nullptr;  // (experiments)
  bool record_blocks = false;
  if (b->auto_arena && b->grid > b->cus && (few ? can16 && !no_wide : !no_record_blocks)) {
    // more streams than CUs, few enough for engine blocks to pay where the streams are the engines' kind: the device says which are.
    // Not for batches of small streams (a mean of less than 8 KiB compressed: an engine has nothing to spread out, and the probe -- a
    // launch and a wait on the caller's stream -- would cost such a batch more than its decode), and not twice for the same descriptors.
    uint64_t in_total = 0, in_engine = 0, key = 0xcbf29ce484222325ull ^ n;
    for (uint32_t i = 0; i < n; i++) {
      in_total += b->h_descs[i].in_size;
      for (uint64_t v : {(uint64_t)(uintptr_t)b->h_descs[i].in, (uint64_t)b->h_descs[i].in_size, (uint64_t)b->h_descs[i].flags}) key = (key ^ v) * 0x100000001b3ull;
    }
    if (in_total >= (uint64_t)n * kProbeMinMeanBytes) {
      if (b->probe_kind.size() == n && b->probe_key == key) kind = b->probe_kind;
      else {
        if (!ensure_scratch(b, b->grid)) return -1;   // (a batch object's first launch allocates its blocks' scratch: not the probe's time)
        const auto t0 = std::chrono::steady_clock::now();
        if (probe_streams(b, n, stream, kind) != 0) return -1;
        b->last_probe_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        b->probe_kind = kind; b->probe_key = key;
      }
      for (uint32_t i = 0; i < n; i++) if (kind[i] == 7u) in_engine += b->h_descs[i].in_size;   // (15: the engines' kind but for its short commands -- text)
      if (getenv("BROTLI_AMD_DEBUG_PROBE")) {
        uint32_t h[16] = {}; for (uint32_t i = 0; i < n; i++) h[kind[i] & 15u]++;
        fprintf(stderr, "probe: %u streams, kinds", n); for (int k = 0; k < 16; k++) if (h[k]) fprintf(stderr, " %d:%u", k, h[k]);
        fprintf(stderr, "; engine bytes %llu of %llu; results", (unsigned long long)in_engine, (unsigned long long)in_total);
        uint32_t r[8] = {}; for (uint32_t i = 0; i < n; i++) r[b->h_status[i].result < 8 ? b->h_status[i].result : 7]++;
        for (int k = 0; k < 8; k++) if (r[k]) fprintf(stderr, " %d:%u", k, r[k]);
        fprintf(stderr, "\n");
      }
      if (few) { if (in_engine * 2u >= in_total && in_engine != 0u) { engine_queue = true; b->grid = b->cus; b->cur_per_cu = 0; } }
      else {
        uint64_t in_rec = 0;
        for (uint32_t i = 0; i < n; i++) if (kind[i] == 5u || kind[i] == 15u) in_rec += b->h_descs[i].in_size;
        record_blocks = in_rec * 2u >= in_total && in_rec != 0u;
      }
    }
  }
  if (record_blocks) { b->cur_arena = b->lds_arena; b->cur_per_cu = 0; b->grid = std::min(n, b->grid_max); b->waves = 4u; }
  if (can16 && b->grid <= b->cus) { b->cur_arena = arena16; b->waves = 16; }
  // Fewer streams than half the CUs: GANGS of blocks, a CU each, on one stream -- its owner and one, three or seven helper blocks that take
  // the path engine's regions in turns with it (csrc/brotli_path_engine.h, PE_CFG_REMOTE).  Eight streams' gangs are launched side by side,
  // a gang's members eight block numbers apart (one XCD); streams beyond a multiple of eight leave their gangs' blocks without work.
  const int gang_env = getenv("BROTLI_AMD_GANG") ? atoi(getenv("BROTLI_AMD_GANG")) : -1;   // (experiments: 0 or 1 none, 2 / 4 / 8 at most that many)
  const int pool_env = getenv("BROTLI_AMD_POOL") ? atoi(getenv("BROTLI_AMD_POOL")) : -1;   // (experiments, tests: 0 no pool, 2 a pool whatever the sizes)
  b->gang = 0;
  if (b->waves == 16u && b->auto_arena && b->cur_arena <= 49152u && n <= b->cus) {
    std::vector<size_t> sz(n);
    for (uint32_t i = 0; i < n; i++) sz[i] = b->h_descs[i].in_size;
    uint32_t grid = b->grid;
    b->gang = plan_gangs(n, b->cus, sz.data(), gang_env, pool_env, &grid);
    b->grid = grid;
  }
  engine_queue = engine_queue && b->waves == 16u;
  if (engine_queue)   // (the engines' streams to the engine blocks; the others wait for the launch of small blocks behind it)
    for (uint32_t i = 0; i < n; i++) b->h_descs[i].flags |= kind[i] == 7u ? BROTLI_AMD_FLAG_ENGINE_ONLY : BROTLI_AMD_FLAG_DEFER;
  // where a larger arena exists, tables that do not fit this one are a reason to come back, not to spill
  if (b->cur_arena < b->max_arena)
    for (uint32_t i = 0; i < n; i++) if (!(b->h_descs[i].flags & BROTLI_AMD_BATCH_SPILL_IN_PLACE)) b->h_descs[i].flags |= BROTLI_AMD_FLAG_NO_SPILL;
  // (a gang's helper blocks have no scratch of their own -- a slot per stream --, but a launch whose kernel decides against gangs after all
  // (fewer waves than sixteen: BROTLI_AMD_LAUNCH experiments) indexes the scratch by block: there is a slot for every block as well)
  if (!ensure_scratch(b, std::max(n, b->grid))) return -1;
  // more streams than blocks: the blocks take them longest first (compressed size as the measure), so that no block starts
  // a long stream when the others are done
  static const bool no_order = getenv("BROTLI_AMD_NO_ORDER") != nullptr;  // (experiments)
  b->ordered = n > b->grid && !no_order;
  if (b->ordered) {
    uint32_t* order = b->h_order + 16;
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order, order + n, [b](uint32_t x, uint32_t y) { return b->h_descs[x].in_size > b->h_descs[y].in_size; });
  }
  if (!hip_ok(hipMemcpyAsync(b->d_descs, b->h_descs, sizeof(BrotliAmdStreamDesc) * n, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(descs)")) return -1;
  return launch(b, stream);
}

// Streams that came back with BROTLI_AMD_RESULT_RETRY_ARENA continue, from the metablock boundary they stopped at, in
// a launch whose blocks have a larger LDS arena (fewer blocks per CU): a first pass packed more than eight blocks to a
// CU is followed by one with eight, then by the configured arena, then by the largest block the device allows, where
// spilling to global memory is allowed; each pass takes only what the one before could not hold.
int run_retry_descs(BrotliAmdBatch* b, uint32_t m, uint32_t arena, uint32_t grid_max, int waves);
int retry_with_larger_arenas(BrotliAmdBatch* b) {
  b->last_retry_count = 0;
  b->retry_ms = 0.0f;
  uint32_t level = b->cur_per_cu;  // 0: the first pass had the configured arena already
  bool many_came_back = false;
  for (int pass = 0; pass < 4; pass++) {
    std::vector<uint32_t> idx;
    for (uint32_t i = 0; i < b->n; i++) if (b->h_status[i].result == BROTLI_AMD_RESULT_RETRY_ARENA) idx.push_back(i);
    const uint32_t m = (uint32_t)idx.size();
    // a good part of the batch did not fit the first pass: later batches of this object start with the shape that
    // did hold (nearly) all of it
    if (many_came_back && m <= b->n / 16) { b->per_cu_cap = std::max(4u, level); many_came_back = false; }
    if (idx.empty()) return 0;
    if (pass == 0) {
      b->last_retry_count = m;
      many_came_back = level > 4u && m > b->n / 16;
    }
    if (!b->h_retry_descs && run_retry_descs(b, 0, b->max_arena, b->retry_grid_max, 4) != 0) return -1;  // (allocates the pass's buffers)
    // shape of this pass
    uint32_t arena, grid_max; int waves; bool last;
    bool deferred = false;   // streams an engine launch sent back unread (BROTLI_AMD_FLAG_DEFER): the launch of small blocks they were promised
    if (pass == 0) for (uint32_t j = 0; j < m && !deferred; j++) deferred = (b->h_descs[idx[j]].flags & BROTLI_AMD_FLAG_DEFER) != 0u;
    if (deferred) {   // the shape submit() gives a batch of m streams without engine blocks: several blocks a CU, streams in flight
      const uint32_t per_cu = (uint32_t)std::min<size_t>(b->per_cu_cap, ((size_t)m + b->cus - 1) / b->cus);
      if (per_cu > 4u && small_arena(b, per_cu) != 0u && small_arena(b, per_cu) < b->lds_arena) { level = per_cu; arena = small_arena(b, per_cu); grid_max = b->cus * per_cu; waves = 1; }
      else { level = 4; arena = b->lds_arena; grid_max = b->grid_max; waves = 4; }
      last = false;
    } else
    if (level > 8u && small_arena(b, 8) > b->cur_arena) { level = 8; arena = small_arena(b, 8); grid_max = b->cus * 8u; waves = 1; last = false; }
    else if (level > 4u && b->lds_arena > b->cur_arena) { level = 4; arena = b->lds_arena; grid_max = b->grid_max; waves = 4; last = false; }
    else { level = 2; arena = b->max_arena; grid_max = b->retry_grid_max; waves = 4; last = true; }
    for (uint32_t j = 0; j < m; j++) {
      BrotliAmdStreamDesc d = b->h_descs[idx[j]];
      d.flags = ((last ? d.flags & ~BROTLI_AMD_FLAG_NO_SPILL : d.flags) & ~(BROTLI_AMD_FLAG_ENGINE_ONLY | BROTLI_AMD_FLAG_DEFER)) | BROTLI_AMD_FLAG_RESUME;
      d.resume = b->h_status[idx[j]].resume;
      b->h_retry_descs[j] = d;
    }
    if (run_retry_descs(b, m, arena, grid_max, waves) != 0) return -1;
    for (uint32_t j = 0; j < m; j++) {
      BrotliAmdStreamStatus& first = b->h_status[idx[j]];
      BrotliAmdStreamStatus next = b->h_retry_status[j];
      next.num_metablocks += first.num_metablocks;  // (the metablock a pass stopped in front of is counted by the pass that decodes it)
      next.num_commands += first.num_commands;
      next.engine_commands += first.engine_commands;
      next.peak_trees = std::max(next.peak_trees, first.peak_trees); next.peak_map_bytes = std::max(next.peak_map_bytes, first.peak_map_bytes);
      next.any_compressed |= first.any_compressed;
      first = next;
    }
    if (last) { if (many_came_back) b->per_cu_cap = 4; return 0; }
  }
  return 0;
}

// m descriptors in h_retry_descs -> h_retry_status, in a launch of the given shape on the job's stream (kernel time added to retry_ms)
int run_retry_descs(BrotliAmdBatch* b, uint32_t m, uint32_t arena, uint32_t grid_max, int waves) {
  if (!b->d_retry_descs) {
    bool ok = hip_ok(hipMalloc(&b->d_retry_descs, sizeof(BrotliAmdStreamDesc) * b->max_streams), "hipMalloc(retry descs)");
    ok = ok && hip_ok(hipMalloc(&b->d_retry_status, sizeof(BrotliAmdStreamStatus) * b->max_streams), "hipMalloc(retry status)");
    ok = ok && hip_ok(hipHostMalloc(&b->h_retry_descs, sizeof(BrotliAmdStreamDesc) * b->max_streams), "hipHostMalloc(retry descs)");
    ok = ok && hip_ok(hipHostMalloc(&b->h_retry_status, sizeof(BrotliAmdStreamStatus) * b->max_streams), "hipHostMalloc(retry status)");
    if (!ok) return -1;
  }
  if (m == 0) return 0;
  const uint32_t grid = std::min(m, grid_max);
  hipStream_t stream = b->last_stream;
  if (!ensure_scratch(b, std::max(grid, b->grid))) return -1;
  if (!hip_ok(hipMemcpyAsync(b->d_retry_descs, b->h_retry_descs, sizeof(BrotliAmdStreamDesc) * m, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(retry descs)")) return -1;
  if (!hip_ok(hipMemsetAsync(b->d_queue, 0, sizeof(uint32_t) * 16, stream), "hipMemsetAsync(queue)")) return -1;
  if (!hip_ok(hipEventRecord(b->ev2, stream), "hipEventRecord")) return -1;
  if (!hip_ok(brotli_amd_launch_decode(b->d_retry_descs, b->d_retry_status, m, b->d_queue, b->d_scratch, kScratchPerBlock, grid, arena,
                                       b->d_dict, stream, waves), "brotli_amd_decode_kernel launch (later pass)")) return -1;
  if (!hip_ok(hipEventRecord(b->ev3, stream), "hipEventRecord")) return -1;
  if (!hip_ok(hipMemcpyAsync(b->h_retry_status, b->d_retry_status, sizeof(BrotliAmdStreamStatus) * m, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(retry status)")) return -1;
  if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return -1;
  { float ms = 0.0f; if (hipEventElapsedTime(&ms, b->ev2, b->ev3) == hipSuccess) b->retry_ms += ms; }
  return 0;
}

// What the reference reports for a stream whose output buffer is too small depends on what the stream does up to its next
// ring-buffer flush point: it decodes into its ring and only notices the full buffer when it flushes (decode.rs:1693-1738;
// the driver ignores NEEDS_MORE_OUTPUT from the flush it forces when the input ends, decode.rs BrotliDecompressStream), so
// an error or the end of the input in front of that point wins over NEEDS_MORE_OUTPUT.  The kernel stops where the
// buffer ends; the streams it reports NEEDS_MORE_OUTPUT for are decoded once more, into scratch memory with room up to
// the flush point, and that outcome is mapped (the bytes in the caller's buffer stay: they are the same).
constexpr size_t kSettleChunkBytes = (size_t)2 << 30;
int settle_output_limits(BrotliAmdBatch* b) {
  b->last_settle_count = 0;
  std::vector<uint32_t> idx;
  for (uint32_t i = 0; i < b->n; i++)
    if (b->h_status[i].result == BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT && b->h_status[i].ring_bytes != 0) idx.push_back(i);
  size_t at = 0;
  while (at < idx.size()) {
    // a chunk of streams whose scratch outputs fit the budget together (a single stream beyond it keeps the kernel's verdict)
    std::vector<uint32_t> part; std::vector<size_t> off, cap2s; size_t total = 0;
    for (; at < idx.size(); at++) {
      const BrotliAmdStreamDesc& d0 = b->h_descs[idx[at]];
      // (one byte short of the flush point: the reference flushes as soon as its ring is full, so a stream that gets that far
      // is told NEEDS_MORE_OUTPUT there, whatever comes behind)
      const uint64_t rb = b->h_status[idx[at]].ring_bytes, cap2 = (d0.out_cap / rb + 1) * rb - 1;
      if (cap2 <= d0.out_cap) continue;  // (the buffer ends right in front of the flush point: nothing more to find out)
      const size_t need = (size_t)((cap2 + 255) & ~(uint64_t)255);
      if (need > kSettleChunkBytes) continue;
      if (total + need > kSettleChunkBytes && !part.empty()) break;
      part.push_back(idx[at]); off.push_back(total); cap2s.push_back((size_t)cap2); total += need;
    }
    if (part.empty()) continue;
    if (total > b->settle_cap) {
      if (b->d_settle) (void)hipFree(b->d_settle);
      b->d_settle = nullptr; b->settle_cap = 0;
      if (hipMalloc(&b->d_settle, total) != hipSuccess) { (void)hipGetLastError(); return 0; }  // (no memory for it: the kernel's verdict stands)
      b->settle_cap = total;
    }
    const uint32_t m = (uint32_t)part.size();
    if (!b->h_retry_descs && run_retry_descs(b, 0, b->max_arena, b->retry_grid_max, 4) != 0) return -1;
    for (uint32_t j = 0; j < m; j++) {
      BrotliAmdStreamDesc d = b->h_descs[part[j]];
      d.flags &= ~(BROTLI_AMD_FLAG_NO_SPILL | BROTLI_AMD_FLAG_ENGINE_ONLY | BROTLI_AMD_FLAG_DEFER | BROTLI_AMD_FLAG_RESUME);
      d.out = b->d_settle + off[j]; d.out_cap = cap2s[j];
      b->h_retry_descs[j] = d;
    }
    if (run_retry_descs(b, m, b->max_arena, b->retry_grid_max, 4) != 0) return -1;
    for (uint32_t j = 0; j < m; j++) {
      BrotliAmdStreamStatus& st = b->h_status[part[j]];
      const BrotliAmdStreamStatus& st2 = b->h_retry_status[j];
      const uint64_t cap = b->h_descs[part[j]].out_cap;
      if (st2.result == BROTLI_DECODER_RESULT_ERROR || (st2.result == BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT && st2.produced <= cap2s[j])) {
        const uint64_t produced = st.produced;
        st = st2;
        st.decoded_size = std::min<uint64_t>(st2.decoded_size, cap);
        st.produced = produced;  // (bytes in the caller's buffer)
      }
    }
    b->last_settle_count += m;
  }
  if (b->settle_cap > ((size_t)64 << 20)) {  // (a large scratch buffer is not kept for the next batch)
    (void)hipFree(b->d_settle);
    b->d_settle = nullptr; b->settle_cap = 0;
  }
  return 0;
}

}  // namespace

extern "C" BrotliAmdBatch* BrotliAmdBatchCreate(uint32_t max_streams, uint32_t lds_arena_bytes, uint32_t grid_blocks) {
  DeviceGuard guard;
  int dev = 0;
  if (!current_device(&dev)) return nullptr;
  if (max_streams == 0) max_streams = 1;
  BrotliAmdBatch* b = new (std::nothrow) BrotliAmdBatch();
  if (!b) return nullptr;
  b->device = dev;
  b->max_streams = max_streams;
  hipDeviceProp_t prop;
  if (!hip_ok(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) { delete b; return nullptr; }
  // LDS of a block = fixed carve + table arena (+ what the helper waves leave for each other, in blocks that have them)
  const uint32_t fixed = brotli_amd_lds_fixed_bytes(), helper = brotli_amd_lds_helper_bytes(4);
  uint32_t per_block = lds_arena_bytes ? lds_arena_bytes + fixed + helper : kDefaultLdsPerBlock;
  size_t lds_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
  if (per_block > prop.sharedMemPerBlock && prop.sharedMemPerBlock) per_block = (uint32_t)prop.sharedMemPerBlock;
  b->lds_arena = (per_block - fixed - helper) & ~15u;
  b->cur_arena = b->lds_arena;
  b->auto_arena = lds_arena_bytes == 0;
  b->per_cu_cap = (uint32_t)kMaxBlocksPerCu;
  b->cus = (uint32_t)prop.multiProcessorCount; b->lds_fixed = fixed; b->lds_helper = helper; b->lds_helper8 = brotli_amd_lds_helper_bytes(8); b->lds_per_cu = lds_cu;
  b->block_max = (uint32_t)std::min<size_t>(prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536, 65536);
  b->engine_ok = g_engine_wanted && lds_cu >= (size_t)fixed + brotli_amd_lds_helper_bytes(16) + 16384u;
  {  // the arena of the second pass: the largest block the device allows (at most 64 KiB: two such blocks per CU at least)
    uint32_t big = (uint32_t)std::min<size_t>(prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536, 65536);
    b->max_arena = big > fixed + helper ? (big - fixed - helper) & ~15u : 0;
    b->retry_grid_max = (uint32_t)prop.multiProcessorCount * (uint32_t)std::max<size_t>(1, lds_cu / big);
  }
  uint32_t blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(kMaxBlocksPerCu, lds_cu / per_block));
  b->grid_max = grid_blocks ? grid_blocks : (uint32_t)prop.multiProcessorCount * blocks_per_cu;
  b->d_dict = device_dictionary(dev);
  bool ok = b->d_dict != nullptr;
  ok = ok && hip_ok(hipMalloc(&b->d_descs, sizeof(BrotliAmdStreamDesc) * max_streams), "hipMalloc(descs)");
  ok = ok && hip_ok(hipMalloc(&b->d_status, sizeof(BrotliAmdStreamStatus) * max_streams), "hipMalloc(status)");
  ok = ok && hip_ok(hipMalloc(&b->d_queue, sizeof(uint32_t) * (16 + (size_t)max_streams)), "hipMalloc(queue)");
  ok = ok && hip_ok(hipHostMalloc(&b->h_descs, sizeof(BrotliAmdStreamDesc) * max_streams), "hipHostMalloc(descs)");
  ok = ok && hip_ok(hipHostMalloc(&b->h_status, sizeof(BrotliAmdStreamStatus) * max_streams), "hipHostMalloc(status)");
  ok = ok && hip_ok(hipHostMalloc(&b->h_order, sizeof(uint32_t) * (16 + (size_t)max_streams)), "hipHostMalloc(order)");
  ok = ok && hip_ok(hipEventCreate(&b->ev0), "hipEventCreate") && hip_ok(hipEventCreate(&b->ev1), "hipEventCreate");
  ok = ok && hip_ok(hipEventCreate(&b->ev2), "hipEventCreate") && hip_ok(hipEventCreate(&b->ev3), "hipEventCreate");
  ok = ok && hip_ok(hipMemset(b->d_status, 0, sizeof(BrotliAmdStreamStatus) * max_streams), "hipMemset(status)");
  if (ok) std::memset(b->h_status, 0, sizeof(BrotliAmdStreamStatus) * max_streams);
  if (!ok) { BrotliAmdBatchDestroy(b); return nullptr; }
  return b;
}

extern "C" void BrotliAmdBatchDestroy(BrotliAmdBatch* b) {
  if (!b) return;
  DeviceGuard guard;
  (void)hipSetDevice(b->device);
  if (b->launched && b->last_stream != nullptr) (void)hipStreamSynchronize(b->last_stream);
  else (void)hipDeviceSynchronize();
  if (b->d_descs) (void)hipFree(b->d_descs);
  if (b->d_status) (void)hipFree(b->d_status);
  if (b->d_queue) (void)hipFree(b->d_queue);
  if (b->d_gang) (void)hipFree(b->d_gang);
  if (b->d_scratch) (void)hipFree(b->d_scratch);
  if (b->d_retry_descs) (void)hipFree(b->d_retry_descs);
  if (b->d_retry_status) (void)hipFree(b->d_retry_status);
  if (b->h_retry_descs) (void)hipHostFree(b->h_retry_descs);
  if (b->h_retry_status) (void)hipHostFree(b->h_retry_status);
  if (b->d_stage_in) (void)hipFree(b->d_stage_in);
  if (b->d_stage_out) (void)hipFree(b->d_stage_out);
  if (b->h_pin_in) (void)hipHostFree(b->h_pin_in);
  if (b->h_pin_out) (void)hipHostFree(b->h_pin_out);
  if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
  if (b->d_settle) (void)hipFree(b->d_settle);
  if (b->h_descs) (void)hipHostFree(b->h_descs);
  if (b->h_status) (void)hipHostFree(b->h_status);
  if (b->h_order) (void)hipHostFree(b->h_order);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->ev2) (void)hipEventDestroy(b->ev2);
  if (b->ev3) (void)hipEventDestroy(b->ev3);
  delete b;
}

extern "C" int BrotliAmdBatchDecodeDevice(BrotliAmdBatch* b, uint32_t n, const void* const* d_in, const size_t* in_sizes, void* const* d_out,
                                          const size_t* out_caps, uint32_t flags, void* hip_stream) {
  if (!b || n > b->max_streams || (n && (!d_in || !in_sizes || !d_out || !out_caps))) { g_last_error = "invalid batch arguments"; return -1; }
  DeviceGuard guard;
  for (uint32_t i = 0; i < n; i++) {
    BrotliAmdStreamDesc& d = b->h_descs[i];
    std::memset(&d, 0, sizeof d);
    d.in = static_cast<const uint8_t*>(d_in[i]); d.in_size = in_sizes[i];
    d.out = static_cast<uint8_t*>(d_out[i]); d.out_cap = out_caps[i];
    d.flags = flags & (BROTLI_AMD_FLAG_LARGE_WINDOW | BROTLI_AMD_FLAG_NO_CANNY | BROTLI_AMD_BATCH_SPILL_IN_PLACE);
  }
  b->exact_limit = !(flags & BROTLI_AMD_BATCH_EAGER_OUTPUT_LIMIT);
  return submit(b, n, static_cast<hipStream_t>(hip_stream));
}

extern "C" int BrotliAmdBatchRelaunch(BrotliAmdBatch* b, void* hip_stream) {
  if (!b || b->n == 0) { g_last_error = "nothing to relaunch"; return -1; }
  DeviceGuard guard;
  if (!hip_ok(hipSetDevice(b->device), "hipSetDevice")) return -1;
  return launch(b, static_cast<hipStream_t>(hip_stream));
}

extern "C" int BrotliAmdBatchWait(BrotliAmdBatch* b, BrotliAmdResult* results) {
  if (!b) return -1;
  if (b->n == 0 || !b->launched) return 0;
  DeviceGuard guard;
  if (!hip_ok(hipSetDevice(b->device), "hipSetDevice")) return -1;
  if (!hip_ok(hipMemcpyAsync(b->h_status, b->d_status, sizeof(BrotliAmdStreamStatus) * b->n, hipMemcpyDeviceToHost, b->last_stream), "hipMemcpyAsync(status)")) return -1;
  if (!hip_ok(hipStreamSynchronize(b->last_stream), "hipStreamSynchronize")) return -1;
#ifdef BROTLI_AMD_GANG_STATS
  if (b->last_gang > 1u && b->d_gang && getenv("BROTLI_AMD_GANG_STATS")) {   // (profile build: the first stream's gang)
    unsigned long long st[40];
    if (hipMemcpy(st, b->d_gang + 704, sizeof st, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "gang of %u: invocations %llu, regions arrived at %llu, of them not usable %llu (no tables %llu, short of the window %llu, too far into it %llu), rebuilt by a new plan %llu, tables built %llu; "
              "ticks: owner waits for helpers to leave %llu, for the stream %llu (helpers %llu), for the region before's output %llu, for the regions' before that (executes that wait twice) %llu, at the end %llu; regions resolved %llu, declined %llu; owner inside the engine %llu (set-up %llu, its tables %llu, its regions' walk .. execute %llu), the releases behind a region's output %llu; invocations that took fewer than 64 commands %llu (none: %llu), commands in all %llu; from the stream's arrival, summed over the regions: walk done %llu, details %llu, resolve %llu, output complete %llu; (the walk's start %llu, the entry's states %llu, the anchors %llu); regions whose execute waited once %llu; invocations that resolved no region %llu; regions given up behind their details: the invocation had ended %llu, the region before said so %llu, the stream went on elsewhere %llu, no quota or commands left %llu; resolves that ended an invocation %llu (a limit cut the region: %llu); ticks waiting for the state behind the details %llu\n",
              b->last_gang, st[0], st[1], st[2], st[10], st[11], st[12], st[3], st[4], st[5], st[6], st[7], st[8], st[9], st[13], st[14], st[15], st[16], st[18], st[20], st[19], st[17], st[21], st[23], st[22], st[24], st[25], st[26], st[27], st[28], st[29], st[38], st[30], st[31], st[32], st[33], st[34], st[36], st[37], st[35], st[39]);
  }
#endif
#ifdef BROTLI_AMD_GANG_TRACE
  if (b->last_gang > 1u && b->d_gang) {   // (profile build: the first regions of one invocation of the first stream's gang, the shared 100 MHz clock at every hand-over)
    static unsigned long long tr[64][16];
    if (hipMemcpy(tr, b->d_gang + 1024 + (40u << 10), sizeof tr, hipMemcpyDeviceToHost) == hipSuccess)
      for (int k = 0; k < 64 && tr[k][6]; k++) {
        fprintf(stderr, "GT %d blk %llu m %llu ndep %llu :", k, tr[k][15] >> 32, tr[k][15] & 0xffffull, (tr[k][15] >> 16) & 0xffffull);
        for (int q = 0; q < 15; q++) fprintf(stderr, " %llu", tr[k][q]);
        fprintf(stderr, "\n");
      }
  }
#endif
  if (retry_with_larger_arenas(b) != 0) return -1;
  if (b->exact_limit && settle_output_limits(b) != 0) return -1;
  if (results) {
    for (uint32_t i = 0; i < b->n; i++) {
      const BrotliAmdStreamStatus& s = b->h_status[i];
      BrotliAmdResult& r = results[i];
      r.result = s.result; r.error_code = s.error_code; r.decoded_size = s.decoded_size; r.consumed = s.consumed;
      r.produced = s.produced; r.num_metablocks = s.num_metablocks; r.spilled_metablocks = s.spilled_metablocks; r.num_commands = s.num_commands;
      r.engine_commands = s.engine_commands; r.reserved = 0;
    }
  }
  return 0;
}

extern "C" uint32_t BrotliAmdBatchLastSecondPassCount(BrotliAmdBatch* b) { return b ? b->last_retry_count : 0; }
extern "C" uint32_t BrotliAmdBatchLastGang(BrotliAmdBatch* b) { return b ? (b->last_gang > 1u && b->last_gang <= 16u ? b->last_gang : 1u) : 0; }
extern "C" uint32_t BrotliAmdDebugPlanGangs(uint32_t n, uint32_t cus, const size_t* in_sizes, int gang_env, int pool_env, uint32_t* grid) {
  uint32_t g = n;
  const uint32_t r = (n != 0u && in_sizes != nullptr) ? plan_gangs(n, cus, in_sizes, gang_env, pool_env, &g) : 0u;
  if (grid) *grid = g;
  return r;
}
extern "C" float BrotliAmdBatchLastProbeMs(BrotliAmdBatch* b) { return b ? b->last_probe_ms : 0.0f; }
extern "C" uint32_t BrotliAmdBatchLastPool(BrotliAmdBatch* b) { return b && (b->last_gang & BROTLI_AMD_GANG_POOL_FLAG) != 0u ? 1u : 0u; }

extern "C" float BrotliAmdBatchLastKernelMs(BrotliAmdBatch* b) {
  if (!b || !b->launched) return 0.0f;
  float ms = 0.0f;
  if (!hip_ok(hipEventSynchronize(b->ev1), "hipEventSynchronize")) return -1.0f;
  if (!hip_ok(hipEventElapsedTime(&ms, b->ev0, b->ev1), "hipEventElapsedTime")) return -1.0f;
  return ms + b->retry_ms;
}

extern "C" int BrotliAmdBatchDecodeHost(BrotliAmdBatch* b, uint32_t n, const uint8_t* const* in, const size_t* in_sizes, uint8_t* const* out,
                                        const size_t* out_caps, uint32_t flags, BrotliAmdResult* results) {
  if (!b || n > b->max_streams || (n && (!in || !in_sizes || !out || !out_caps))) { g_last_error = "invalid batch arguments"; return -1; }
  if (n == 0) return 0;
  DeviceGuard guard;
  if (!hip_ok(hipSetDevice(b->device), "hipSetDevice")) return -1;
  // one input arena and one output arena, 64-byte aligned slots
  std::vector<size_t> in_off(n), out_off(n);
  size_t in_total = 0, out_total = 0;
  for (uint32_t i = 0; i < n; i++) {
    in_off[i] = in_total; in_total += (in_sizes[i] + 63) & ~(size_t)63; in_total += 64;
    out_off[i] = out_total; out_total += (out_caps[i] + 63) & ~(size_t)63; out_total += 64;
  }
  if (in_total > b->stage_in_cap) {
    if (b->d_stage_in) (void)hipFree(b->d_stage_in);
    b->d_stage_in = nullptr; b->stage_in_cap = 0;
    if (!hip_ok(hipMalloc(&b->d_stage_in, in_total), "hipMalloc(input arena)")) return -1;
    b->stage_in_cap = in_total;
  }
  if (out_total > b->stage_out_cap) {
    if (b->d_stage_out) (void)hipFree(b->d_stage_out);
    b->d_stage_out = nullptr; b->stage_out_cap = 0;
    if (!hip_ok(hipMalloc(&b->d_stage_out, out_total), "hipMalloc(output arena)")) return -1;
    b->stage_out_cap = out_total;
  }
  // pinned staging on the host (kept with the batch object) and a stream for the transfers; where the host has no pinned memory to give,
  // the transfers go stream by stream from and to the caller's own (pageable) buffers
  bool pinned = true;
  if (in_total > b->pin_in_cap) {
    if (b->h_pin_in) (void)hipHostFree(b->h_pin_in);
    b->h_pin_in = nullptr; b->pin_in_cap = 0;
    if (hipHostMalloc(&b->h_pin_in, in_total, hipHostMallocDefault) == hipSuccess) b->pin_in_cap = in_total;
    else { (void)hipGetLastError(); b->h_pin_in = nullptr; pinned = false; }
  }
  if (pinned && out_total > b->pin_out_cap) {
    if (b->h_pin_out) (void)hipHostFree(b->h_pin_out);
    b->h_pin_out = nullptr; b->pin_out_cap = 0;
    if (hipHostMalloc(&b->h_pin_out, out_total, hipHostMallocDefault) == hipSuccess) b->pin_out_cap = out_total;
    else { (void)hipGetLastError(); b->h_pin_out = nullptr; pinned = false; }
  }
  if (!b->copy_stream && !hip_ok(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking), "hipStreamCreate")) return -1;
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  // streams [lo, hi) copied by up to `hw` threads, split by bytes
  auto parallel_copy = [&](uint32_t lo, uint32_t hi, auto&& one) {
    size_t bytes = 0; for (uint32_t i = lo; i < hi; i++) bytes += one(i, false);
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(hw, bytes >> 20));
    if (nt <= 1) { for (uint32_t i = lo; i < hi; i++) (void)one(i, true); return; }
    std::vector<std::thread> ts; const size_t per = (bytes + nt - 1) / nt; uint32_t i0 = lo;
    for (unsigned t = 0; t < nt && i0 < hi; t++) {
      uint32_t i1 = i0; size_t acc = 0;
      while (i1 < hi && (acc < per || t + 1 == nt)) acc += one(i1++, false);
      try { ts.emplace_back([=, &one]() { for (uint32_t i = i0; i < i1; i++) (void)one(i, true); }); }
      catch (const std::system_error&) { for (uint32_t i = i0; i < i1; i++) (void)one(i, true); }   // (no thread to be had: this one does the part)
      i0 = i1;
    }
    for (auto& t : ts) t.join();
  };
  // upload: the inputs packed into pinned memory by several threads, piece by piece, each piece's transfer behind it
  if (!pinned) {
    for (uint32_t i = 0; i < n; i++)
      if (in_sizes[i] && !hip_ok(hipMemcpyAsync(b->d_stage_in + in_off[i], in[i], in_sizes[i], hipMemcpyHostToDevice, b->copy_stream), "hipMemcpyAsync(input)")) return -1;
  } else {
    uint32_t lo = 0;
    while (lo < n) {
      uint32_t hi = lo; size_t acc = 0;
      while (hi < n && acc < ((size_t)32 << 20)) acc += in_sizes[hi++];
      parallel_copy(lo, hi, [&](uint32_t i, bool go) -> size_t { if (go && in_sizes[i]) std::memcpy(b->h_pin_in + in_off[i], in[i], in_sizes[i]); return in_sizes[i]; });
      const size_t o0 = in_off[lo], o1 = hi < n ? in_off[hi] : in_total;
      if (!hip_ok(hipMemcpyAsync(b->d_stage_in + o0, b->h_pin_in + o0, o1 - o0, hipMemcpyHostToDevice, b->copy_stream), "hipMemcpyAsync(input)")) return -1;
      lo = hi;
    }
  }
  for (uint32_t i = 0; i < n; i++) {
    BrotliAmdStreamDesc& d = b->h_descs[i];
    std::memset(&d, 0, sizeof d);
    d.in = b->d_stage_in + in_off[i]; d.in_size = in_sizes[i];
    d.out = b->d_stage_out + out_off[i]; d.out_cap = out_caps[i];
    d.flags = flags & (BROTLI_AMD_FLAG_LARGE_WINDOW | BROTLI_AMD_FLAG_NO_CANNY | BROTLI_AMD_BATCH_SPILL_IN_PLACE);
  }
  b->exact_limit = !(flags & BROTLI_AMD_BATCH_EAGER_OUTPUT_LIMIT);
  if (!hip_ok(hipStreamSynchronize(b->copy_stream), "hipStreamSynchronize(upload)")) return -1;
  if (submit(b, n, nullptr) != 0) return -1;
  std::vector<BrotliAmdResult> local;
  if (!results) { local.resize(n); results = local.data(); }
  if (BrotliAmdBatchWait(b, results) != 0) return -1;
  // download: pieces of about 32 MiB into pinned memory, the copies into the caller's buffers (several threads) side by side with the
  // next piece's transfer
  if (!pinned) {
    for (uint32_t i = 0; i < n; i++) {
      const size_t got = (size_t)std::min<uint64_t>(results[i].decoded_size, out_caps[i]);
      if (got && !hip_ok(hipMemcpyAsync(out[i], b->d_stage_out + out_off[i], got, hipMemcpyDeviceToHost, b->copy_stream), "hipMemcpyAsync(output)")) return -1;
    }
    if (!hip_ok(hipStreamSynchronize(b->copy_stream), "hipStreamSynchronize(download)")) return -1;
  } else {
    std::vector<std::pair<uint32_t, uint32_t>> pieces; std::vector<hipEvent_t> evs;
    uint32_t lo = 0;
    bool ok = true;
    const auto got_of = [&](uint32_t i) { return (size_t)std::min<uint64_t>(results[i].decoded_size, out_caps[i]); };
    while (lo < n && ok) {
      uint32_t hi = lo; size_t acc = 0;
      while (hi < n && acc < ((size_t)32 << 20)) { acc += got_of(hi); hi++; }
      // a transfer per run of streams that filled their slots (the rule); a stream that stopped short of its slot ends a run, so that a
      // failed stream with a large buffer costs its decoded bytes, not its capacity
      for (uint32_t r0 = lo; r0 < hi && ok; ) {
        uint32_t r1 = r0;
        while (r1 + 1 < hi && got_of(r1) + 4096 >= out_caps[r1]) r1++;
        const size_t o0 = out_off[r0], o1 = out_off[r1] + got_of(r1);
        if (o1 > o0) ok = hip_ok(hipMemcpyAsync(b->h_pin_out + o0, b->d_stage_out + o0, o1 - o0, hipMemcpyDeviceToHost, b->copy_stream), "hipMemcpyAsync(output)");
        r0 = r1 + 1;
      }
      hipEvent_t ev = nullptr;
      ok = ok && hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate") && hip_ok(hipEventRecord(ev, b->copy_stream), "hipEventRecord");
      evs.push_back(ev); pieces.emplace_back(lo, hi);
      lo = hi;
    }
    for (size_t k = 0; k < pieces.size() && ok; k++) {
      ok = hip_ok(hipEventSynchronize(evs[k]), "hipEventSynchronize");
      if (ok) parallel_copy(pieces[k].first, pieces[k].second, [&](uint32_t i, bool go) -> size_t {
        const size_t got = (size_t)std::min<uint64_t>(results[i].decoded_size, out_caps[i]);
        if (go && got) std::memcpy(out[i], b->h_pin_out + out_off[i], got);
        return got; });
    }
    (void)hipStreamSynchronize(b->copy_stream);
    for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
    if (!ok) return -1;
  }
  return 0;
}

extern "C" const char* BrotliAmdLastError(void) { return g_last_error.c_str(); }
extern "C" const char* BrotliAmdLastNote(void) { return g_last_note.c_str(); }

// Test hook: the device's table builder alone (see brotli_amd_debug_build_tree_kernel).
extern "C" hipError_t brotli_amd_launch_debug_build_tree(const uint8_t* d_lengths, uint32_t n_sym, uint16_t* d_decoded, uint32_t* d_entries, hipStream_t stream);
extern "C" int BrotliAmdDebugBuildTree(const uint8_t* code_lengths, uint32_t alphabet_size, uint16_t* decoded, uint32_t* table_entries) {
  if (code_lengths == nullptr || decoded == nullptr || table_entries == nullptr || alphabet_size == 0u || alphabet_size > 1128u) return -1;
  uint8_t* d_len = nullptr; uint16_t* d_dec = nullptr; uint32_t* d_n = nullptr;
  bool ok = hip_ok(hipMalloc(&d_len, alphabet_size), "hipMalloc") && hip_ok(hipMalloc(&d_dec, 32768 * sizeof(uint16_t)), "hipMalloc") && hip_ok(hipMalloc(&d_n, sizeof(uint32_t)), "hipMalloc");
  ok = ok && hip_ok(hipMemcpy(d_len, code_lengths, alphabet_size, hipMemcpyHostToDevice), "hipMemcpy");
  ok = ok && hip_ok(brotli_amd_launch_debug_build_tree(d_len, alphabet_size, d_dec, d_n, nullptr), "brotli_amd_debug_build_tree_kernel launch");
  ok = ok && hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
  ok = ok && hip_ok(hipMemcpy(decoded, d_dec, 32768 * sizeof(uint16_t), hipMemcpyDeviceToHost), "hipMemcpy") && hip_ok(hipMemcpy(table_entries, d_n, sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy");
  if (d_len) (void)hipFree(d_len);
  if (d_dec) (void)hipFree(d_dec);
  if (d_n) (void)hipFree(d_n);
  return ok && *table_entries != 0u ? 0 : -1;
}

// ============================================ error strings ============================================
// reference src/state.rs:533-578 (including the historical "FL_SPACE" spelling of CL_SPACE)
extern "C" const char* BrotliDecoderErrorString(BrotliDecoderErrorCode c) {
  switch ((int)c) {
    case 0: return "NO_ERROR";
    case 1: return "SUCCESS";
    case 2: return "NEEDS_MORE_INPUT";
    case 3: return "NEEDS_MORE_OUTPUT";
    case -1: return "ERROR_FORMAT_EXUBERANT_NIBBLE";
    case -2: return "ERROR_FORMAT_RESERVED";
    case -3: return "ERROR_FORMAT_EXUBERANT_META_NIBBLE";
    case -4: return "ERROR_FORMAT_SIMPLE_HUFFMAN_ALPHABET";
    case -5: return "ERROR_FORMAT_SIMPLE_HUFFMAN_SAME";
    case -6: return "ERROR_FORMAT_FL_SPACE";
    case -7: return "ERROR_FORMAT_HUFFMAN_SPACE";
    case -8: return "ERROR_FORMAT_CONTEXT_MAP_REPEAT";
    case -9: return "ERROR_FORMAT_BLOCK_LENGTH_1";
    case -10: return "ERROR_FORMAT_BLOCK_LENGTH_2";
    case -11: return "ERROR_FORMAT_TRANSFORM";
    case -12: return "ERROR_FORMAT_DICTIONARY";
    case -13: return "ERROR_FORMAT_WINDOW_BITS";
    case -14: return "ERROR_FORMAT_PADDING_1";
    case -15: return "ERROR_FORMAT_PADDING_2";
    case -16: return "ERROR_FORMAT_DISTANCE";
    case -19: return "ERROR_DICTIONARY_NOT_SET";
    case -20: return "ERROR_INVALID_ARGUMENTS";
    case -21: return "ERROR_ALLOC_CONTEXT_MODES";
    case -22: return "ERROR_ALLOC_TREE_GROUPS";
    case -25: return "ERROR_ALLOC_CONTEXT_MAP";
    case -26: return "ERROR_ALLOC_RING_BUFFER_1";
    case -27: return "ERROR_ALLOC_RING_BUFFER_2";
    case -30: return "ERROR_ALLOC_BLOCK_TYPE_TREES";
    case -31: return "ERROR_UNREACHABLE";
    default: return "ERROR_UNREACHABLE";
  }
}

extern "C" uint32_t BrotliDecoderVersion(void) { return 0x1000f00; }  // ffi/mod.rs:588-590

// ============================================== one-shot ==============================================
namespace {

// window bits announced by the first bytes of a stream (RFC 7932 section 9.1; same answer as the reference's
// lg_window_size, src/decode.rs:1221-1253).  0 when not decidable.
uint32_t peek_window_bits(const uint8_t* in, size_t n) {
  if (n == 0) return 0;
  uint8_t b = in[0];
  if ((b & 1) == 0) return 16;
  if ((b & 0xE) != 0) return 17 + ((b >> 1) & 7);
  uint32_t n3 = (b >> 4) & 7;
  if (n3 == 1) {  // large window: 6 bits of WBITS follow the reserved bit
    if (n < 2 || (b & 0x80)) return 0;
    uint32_t w = in[1] & 0x3F;
    return (w >= 10 && w <= 30) ? w : 0;
  }
  return n3 ? 8 + n3 : 17;
}

// Device resources of the one-shot entry points: one set per calling thread (the reference's one-shot calls share
// nothing -- src/lib.rs:447-468 builds a fresh state per call -- so concurrent callers must not serialise on a lock);
// freed when the thread ends.
struct OneShot {
  BrotliAmdBatch* batch = nullptr;
  uint8_t* d_in = nullptr; size_t in_cap = 0;
  uint8_t* d_out = nullptr; size_t out_cap = 0;
  int device = -1;
  void release() {
    if (batch) BrotliAmdBatchDestroy(batch);
    if (d_in || d_out) { DeviceGuard guard; if (device >= 0) (void)hipSetDevice(device); if (d_in) (void)hipFree(d_in); if (d_out) (void)hipFree(d_out); }
    batch = nullptr; d_in = d_out = nullptr; in_cap = out_cap = 0; device = -1;
  }
  ~OneShot() { release(); }
};
thread_local OneShot t_oneshot;

void fill_error(BrotliDecoderReturnInfo* r, BrotliDecoderErrorCode code, const char* msg) {
  std::memset(r, 0, sizeof *r);
  r->result = BROTLI_DECODER_RESULT_ERROR;
  r->code = code;
  std::snprintf(r->error, sizeof r->error, "%s", msg ? msg : BrotliDecoderErrorString(code));
}

bool grow(uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return true;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  size_t want = std::max<size_t>(need, 4096);
  if (!hip_ok(hipMalloc(p, want + 256), "hipMalloc(one-shot staging)")) return false;
  *cap = want;
  return true;
}

// Decode with output capacity `cap` on the device; returns false on a runtime (HIP) failure.
bool run_once(OneShot& o, size_t n_in, size_t cap, uint32_t flags, BrotliAmdStreamStatus* st) {
  BrotliAmdStreamDesc& d = o.batch->h_descs[0];
  std::memset(&d, 0, sizeof d);
  d.in = o.d_in; d.in_size = n_in; d.out = o.d_out; d.out_cap = cap; d.flags = flags;
  if (submit(o.batch, 1, nullptr) != 0) return false;
  if (BrotliAmdBatchWait(o.batch, nullptr) != 0) return false;
  *st = o.batch->h_status[0];
  return true;
}

// reference src/lib.rs:447-468 (brotli_decode) + BrotliDecoderReturnInfo::new (lib.rs:343-370)
BrotliDecoderReturnInfo oneshot_decode(const uint8_t* in, size_t n_in, uint8_t* out, size_t cap, BrotliAmdStreamStatus* status_out = nullptr) {
  BrotliDecoderReturnInfo r;
  OneShot& o = t_oneshot;
  DeviceGuard guard;
  int dev = 0;
  if (!current_device(&dev)) { fill_error(&r, BROTLI_DECODER_ERROR_UNREACHABLE, ("HIP device unavailable: " + g_last_error).c_str()); return r; }
  if (o.batch && o.device != dev) o.release();  // the caller moved to another device
  if (!o.batch) { o.batch = BrotliAmdBatchCreate(1, 0, 0); o.device = dev; }
  if (!o.batch) { fill_error(&r, BROTLI_DECODER_ERROR_UNREACHABLE, ("HIP device unavailable: " + g_last_error).c_str()); return r; }
  const uint32_t flags = BROTLI_AMD_FLAG_LARGE_WINDOW;  // lib.rs:457 -> BrotliState::new -> large_window = true
  BrotliAmdStreamStatus st;
  bool ok = grow(&o.d_in, &o.in_cap, n_in) && grow(&o.d_out, &o.out_cap, cap);
  ok = ok && (n_in == 0 || hip_ok(hipMemcpy(o.d_in, in, n_in, hipMemcpyHostToDevice), "hipMemcpy(input)"));
  ok = ok && run_once(o, n_in, cap, flags, &st);
  if (ok && st.result == BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT) {
    // The reference only notices a full output buffer at its next ring-buffer flush (decode.rs:1693-1738),
    // so what it reports depends on what the stream does up to the next multiple of the window size.  Decode
    // again with room up to that point and map the outcome.
    uint32_t wbits = peek_window_bits(in, n_in);
    if (wbits) {
      size_t rb = st.ring_bytes ? (size_t)st.ring_bytes : (size_t)1 << wbits;  // (the emulated ring: smaller than the window for a short last metablock)
      size_t cap2 = (cap / rb + 1) * rb - 1;  // (one byte short of the flush point: the reference flushes as soon as its ring is full)
      BrotliAmdStreamStatus st2;
      if (grow(&o.d_out, &o.out_cap, cap2) && run_once(o, n_in, cap2, flags, &st2)) {
        if (st2.result == BROTLI_DECODER_RESULT_ERROR) {
          st = st2;
          if (st.decoded_size > cap) st.decoded_size = cap;
        } else if (st2.result == BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT && st2.produced <= cap2) {
          st = st2; st.decoded_size = std::min<uint64_t>(st2.decoded_size, cap);
        } else {
          st.result = BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT; st.error_code = BROTLI_DECODER_NEEDS_MORE_OUTPUT; st.decoded_size = cap;
        }
      } else ok = false;
    }
  }
  if (!ok) { fill_error(&r, BROTLI_DECODER_ERROR_UNREACHABLE, ("HIP runtime failure: " + g_last_error).c_str()); return r; }
  size_t got = (size_t)std::min<uint64_t>(st.decoded_size, cap);
  if (got && !hip_ok(hipMemcpy(out, o.d_out, got, hipMemcpyDeviceToHost), "hipMemcpy(output)")) {
    fill_error(&r, BROTLI_DECODER_ERROR_UNREACHABLE, ("HIP runtime failure: " + g_last_error).c_str());
    return r;
  }
  if (status_out) *status_out = st;
  std::memset(&r, 0, sizeof r);
  r.decoded_size = got;
  r.result = (BrotliDecoderResult)st.result;
  r.code = (BrotliDecoderErrorCode)st.error_code;
  std::snprintf(r.error, sizeof r.error, "%s", BrotliDecoderErrorString(r.code));
  return r;
}

// reference src/ffi/mod.rs:45-61: pointer/length pairs that a slice could not be made of
template <typename T>
bool valid_slice(const T* p, size_t len) {
  if (len == 0) return true;
  if (!p) return false;
  if (((uintptr_t)p) % alignof(T) != 0) return false;
  if (len > (size_t)(PTRDIFF_MAX) / sizeof(T)) return false;
  return (uintptr_t)p + len * sizeof(T) >= (uintptr_t)p;
}

BrotliDecoderReturnInfo invalid_arguments() {
  BrotliDecoderReturnInfo r;
  fill_error(&r, BROTLI_DECODER_ERROR_INVALID_ARGUMENTS, nullptr);
  return r;
}

}  // namespace

extern "C" BrotliDecoderReturnInfo BrotliDecoderDecompressWithReturnInfo(size_t encoded_size, const uint8_t* encoded_buffer, size_t decoded_size,
                                                                         uint8_t* decoded_buffer) {
  if (!valid_slice(encoded_buffer, encoded_size) || !valid_slice(decoded_buffer, decoded_size)) return invalid_arguments();
  return oneshot_decode(encoded_buffer, encoded_size, decoded_buffer, decoded_size);
}

extern "C" BrotliDecoderResult BrotliDecoderDecompress(size_t encoded_size, const uint8_t* encoded_buffer, size_t* decoded_size,
                                                       uint8_t* decoded_buffer) {
  if (!valid_slice(decoded_size, 1)) return BROTLI_DECODER_RESULT_ERROR;  // ffi/mod.rs:269-271
  BrotliDecoderReturnInfo r = BrotliDecoderDecompressWithReturnInfo(encoded_size, encoded_buffer, *decoded_size, decoded_buffer);
  *decoded_size = r.decoded_size;
  return r.result == BROTLI_DECODER_RESULT_SUCCESS ? BROTLI_DECODER_RESULT_SUCCESS : BROTLI_DECODER_RESULT_ERROR;
}

extern "C" BrotliDecoderReturnInfo BrotliDecoderDecompressPrealloc(size_t encoded_size, const uint8_t* encoded_buffer, size_t decoded_size,
                                                                   uint8_t* decoded_buffer, size_t scratch_u8_size, uint8_t* scratch_u8_buffer,
                                                                   size_t scratch_u32_size, uint32_t* scratch_u32_buffer, size_t scratch_hc_size,
                                                                   HuffmanCode* scratch_hc_buffer) {
  if (!valid_slice(encoded_buffer, encoded_size) || !valid_slice(decoded_buffer, decoded_size) ||
      !valid_slice(scratch_u8_buffer, scratch_u8_size) || !valid_slice(scratch_u32_buffer, scratch_u32_size) ||
      !valid_slice(scratch_hc_buffer, scratch_hc_size))
    return invalid_arguments();
  // The reference decodes out of the three scratch slices (src/lib.rs:374-401: stack allocators over them) and a
  // request they cannot serve panics, which the C ABI reports as ERROR_UNREACHABLE with decoded_size 0
  // (src/ffi/mod.rs:686-713).  Nothing is decoded out of them here (the tables live in the GPU's LDS), but the same
  // requests are accounted: the context-map prefix code at creation (state.rs:395), block-type and block-length trees
  // at the first compressed metablock (decode.rs:2958-2969), per metablock its prefix codes (1080 cells and one u32
  // each, huffman/mod.rs:61-72), its context modes and maps (decode.rs:1295, 3155), and the ring buffer
  // (decode.rs:1843-1855).  The model is the peak of what is alive at once: a slice large enough for the peak but too
  // fragmented for the reference's first-fit allocator succeeds here and fails there (documented in decode.h).
  constexpr uint64_t kTable = 1080;  // BROTLI_HUFFMAN_MAX_TABLE_SIZE, huffman/mod.rs:36
  if (scratch_hc_size < kTable) { BrotliDecoderReturnInfo r; fill_error(&r, BROTLI_DECODER_ERROR_UNREACHABLE, "scratch exhausted (HuffmanCode)"); return r; }
  BrotliAmdStreamStatus st;
  std::memset(&st, 0, sizeof st);
  BrotliDecoderReturnInfo r = oneshot_decode(encoded_buffer, encoded_size, decoded_buffer, decoded_size, &st);
  if (r.code == BROTLI_DECODER_ERROR_UNREACHABLE || r.code == BROTLI_DECODER_ERROR_INVALID_ARGUMENTS) return r;
  const uint64_t need_hc = kTable + (st.any_compressed ? 6 * kTable : 0) + (uint64_t)st.peak_trees * kTable;
  const uint64_t need_u32 = st.peak_trees;
  const uint64_t need_u8 = (st.ring_bytes ? st.ring_bytes + 42 + 24 : 0) + st.peak_map_bytes;
  if (need_hc > scratch_hc_size || need_u32 > scratch_u32_size || need_u8 > scratch_u8_size) {
    fill_error(&r, BROTLI_DECODER_ERROR_UNREACHABLE, need_hc > scratch_hc_size ? "scratch exhausted (HuffmanCode)" : need_u32 > scratch_u32_size ? "scratch exhausted (u32)" : "scratch exhausted (u8)");
    return r;
  }
  return r;
}

// ============================================== streaming ==============================================
struct BrotliDecoderStateStruct {
  brotli_alloc_func alloc_func; brotli_free_func free_func; void* opaque;
  bool large_window, canny, used, finished, have_resume;
  int error_code;          // BrotliDecoderErrorCode, latched when fatal (decode.rs:2796-2798)
  int pending_error;       // fatal code found by the device, reported once everything before it is delivered
  char error_text[256]; bool has_error_text;
  BrotliAmdBatch* batch;
  int device;
  // Device copies of the part of the stream that can still matter: compressed bytes from a little in front of the last
  // completed metablock boundary (in_base = stream offset of d_in[0]; d_in_len = stream bytes received in all), output
  // from one window in front of that boundary or from the first byte not yet copied off the device, whichever is lower
  // (out_base = output offset of d_out[0]).  The kernel is handed pointers biased by the bases, so that it goes on
  // addressing the stream and the output from their beginnings.
  uint8_t* d_in; size_t d_in_len, d_in_cap; uint64_t in_base;
  uint8_t* d_out; size_t d_out_cap; uint64_t out_base;
  BrotliAmdResume resume;
  uint64_t fetched;        // output bytes already copied off the device
  uint64_t total_out;      // output bytes handed to the caller (partial_pos_out)
  uint8_t* outq; size_t outq_len, outq_off, outq_cap;  // fetched but not yet handed over
  uint64_t device_commands; // commands the device has decoded for this stream in all its launches together (BrotliAmdDecoderDeviceCommands)
};

namespace {

void* st_alloc(BrotliDecoderState* s, size_t n) { return s->alloc_func ? s->alloc_func(s->opaque, n) : std::malloc(n); }
void st_free(BrotliDecoderState* s, void* p) { if (!p) return; if (s->free_func) s->free_func(s->opaque, p); else std::free(p); }

bool fatal(int code) { return code < 0; }

void set_runtime_error(BrotliDecoderState* s, const char* what) {
  s->error_code = BROTLI_DECODER_ERROR_UNREACHABLE;
  std::snprintf(s->error_text, sizeof s->error_text, "%s: %s", what, g_last_error.c_str());
  s->has_error_text = true;
}

// (Re)allocates a device buffer of at least `need` bytes that starts with bytes [from, from + keep) of the old one.
bool dev_rebase(uint8_t** p, size_t* cap, size_t from, size_t keep, size_t need) {
  if (from == 0 && need <= *cap && *p) return true;
  size_t want = std::max<size_t>(std::max<size_t>(need, from == 0 ? *cap * 2 : *cap), 1 << 16);
  uint8_t* np = nullptr;
  if (!hip_ok(hipMalloc(&np, want + 256), "hipMalloc(stream buffer)")) return false;
  if (*p && keep && !hip_ok(hipMemcpy(np, *p + from, keep, hipMemcpyDeviceToDevice), "hipMemcpy(rebase)")) { (void)hipFree(np); return false; }
  if (*p) (void)hipFree(*p);
  *p = np; *cap = want;
  return true;
}

size_t hand_over(BrotliDecoderState* s, uint8_t* dst, size_t room) {
  size_t n = std::min(room, s->outq_len - s->outq_off);
  if (n) { std::memcpy(dst, s->outq + s->outq_off, n); s->outq_off += n; s->total_out += n; }
  if (s->outq_off == s->outq_len) s->outq_off = s->outq_len = 0;
  return n;
}

// What lies in front of everything a later pass can touch is dropped: input below the last completed metablock
// boundary (with a margin: the reader fetches whole 256-byte pieces), output below both the bytes still to be copied
// off the device and one window (the farthest a back-reference reaches) in front of that boundary.  Memory of an
// instance stays O(window + one metablock), not O(stream).
bool trim_buffers(BrotliDecoderState* s, bool eager = false) {
  if (!s->have_resume || s->resume.window_bits == 0) return true;
  const uint64_t in_keep = (s->resume.bit_pos >> 3) > 1024 ? ((s->resume.bit_pos >> 3) - 1024) & ~(uint64_t)255 : 0;
  if (in_keep > s->in_base && in_keep - s->in_base >= std::max<uint64_t>(1 << 16, s->d_in_cap / 2)) {
    const size_t from = (size_t)(in_keep - s->in_base), keep = (size_t)(s->d_in_len - in_keep);
    if (!dev_rebase(&s->d_in, &s->d_in_cap, from, keep, keep)) return false;
    s->in_base = in_keep;
  }
  const uint64_t window = 1ull << s->resume.window_bits;
  uint64_t out_keep = s->resume.out_pos > window ? s->resume.out_pos - window : 0;
  if (s->fetched < out_keep) out_keep = s->fetched;
  out_keep &= ~(uint64_t)255;
  if (out_keep > s->out_base && out_keep - s->out_base >= (eager ? std::max<uint64_t>(1 << 16, s->d_out_cap / 4) : std::max<uint64_t>(1 << 20, s->d_out_cap / 2))) {
    const size_t from = (size_t)(out_keep - s->out_base);
    const size_t keep = s->d_out_cap - from;  // (whatever the last pass wrote lies below the buffer's end)
    if (!dev_rebase(&s->d_out, &s->d_out_cap, from, keep, s->d_out_cap)) return false;
    s->out_base = out_keep;
  }
  return true;
}

// Copies what the reference would have flushed by now off the device, behind what the caller has not taken yet.
// 0 = ok, 1 = HIP failure, 2 = allocation failure
int fetch_output(BrotliDecoderState* s, uint64_t deliverable) {
  if (deliverable <= s->fetched) return 0;
  size_t n = (size_t)(deliverable - s->fetched);
  if (s->outq_len + n > s->outq_cap) {
    size_t ncap = std::max(s->outq_cap * 2, s->outq_len + n);
    uint8_t* nq = static_cast<uint8_t*>(st_alloc(s, ncap));
    if (!nq) return 2;
    if (s->outq_len) std::memcpy(nq, s->outq, s->outq_len);
    st_free(s, s->outq);
    s->outq = nq; s->outq_cap = ncap;
  }
  if (!hip_ok(hipMemcpy(s->outq + s->outq_len, s->d_out + (s->fetched - s->out_base), n, hipMemcpyDeviceToHost), "hipMemcpy(output)")) return 1;
  s->outq_len += n;
  s->fetched = deliverable;
  return 0;
}

// One decode pass over everything received so far, from the last completed metablock boundary.
// 0 = ok, 1 = HIP failure, 2 = allocation failure
int decode_pass(BrotliDecoderState* s, BrotliAmdStreamStatus* st) {
  for (;;) {
    const size_t pending_in = (size_t)(s->d_in_len - s->in_base);
    if (!dev_rebase(&s->d_out, &s->d_out_cap, 0, s->d_out_cap, std::max<size_t>(s->d_out_cap, std::max<size_t>(1 << 16, 6 * pending_in)))) return 1;
    BrotliAmdStreamDesc& d = s->batch->h_descs[0];
    std::memset(&d, 0, sizeof d);
    d.in = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(s->d_in) - (uintptr_t)s->in_base); d.in_size = s->d_in_len;
    d.out = reinterpret_cast<uint8_t*>(reinterpret_cast<uintptr_t>(s->d_out) - (uintptr_t)s->out_base); d.out_cap = s->out_base + s->d_out_cap;
    d.flags = (s->large_window ? BROTLI_AMD_FLAG_LARGE_WINDOW : 0u) | (s->canny ? 0u : BROTLI_AMD_FLAG_NO_CANNY);
    if (s->have_resume) { d.flags |= BROTLI_AMD_FLAG_RESUME; d.resume = s->resume; }
    if (submit(s->batch, 1, nullptr) != 0) return 1;
    if (BrotliAmdBatchWait(s->batch, nullptr) != 0) return 1;
    *st = s->batch->h_status[0];
    s->device_commands += st->num_commands;
    if (st->resume.window_bits != 0) { s->resume = st->resume; s->have_resume = true; }
    // bytes the reference would have flushed by now: all of them on success / needs-more-input, the part
    // below the last ring-buffer boundary on a fatal error (decode.rs:2835-2846, 2899-2913)
    if (int e = fetch_output(s, st->decoded_size)) return e;
    if (st->result != BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT) return 0;
    // device output buffer exhausted: everything below the resume point is final.  What is dead is dropped; where that
    // does not leave half the buffer free, the buffer doubles.
    const uint64_t before = s->out_base;
    if (!trim_buffers(s, true)) return 1;
    const uint64_t live = (s->have_resume ? s->resume.out_pos : 0) > s->out_base ? (s->have_resume ? s->resume.out_pos : 0) - s->out_base : 0;
    if (s->out_base == before || live > s->d_out_cap / 2)
      if (!dev_rebase(&s->d_out, &s->d_out_cap, 0, s->d_out_cap, s->d_out_cap * 2)) return 1;
  }
}

}  // namespace

extern "C" BrotliDecoderState* BrotliDecoderCreateInstance(brotli_alloc_func alloc_func, brotli_free_func free_func, void* opaque) {
  if ((alloc_func == nullptr) != (free_func == nullptr)) return nullptr;  // ffi/mod.rs:132-135
  void* mem = alloc_func ? alloc_func(opaque, sizeof(BrotliDecoderStateStruct)) : std::malloc(sizeof(BrotliDecoderStateStruct));
  if (!mem) return nullptr;
  BrotliDecoderState* s = static_cast<BrotliDecoderState*>(mem);
  std::memset(s, 0, sizeof *s);
  s->alloc_func = alloc_func; s->free_func = free_func; s->opaque = opaque;
  s->large_window = false;  // ffi/mod.rs:127
  s->canny = true;          // state.rs: canny_ringbuffer_allocation = true
  s->error_code = BROTLI_DECODER_SUCCESS;
  s->device = -1;
  return s;
}

extern "C" void BrotliDecoderDestroyInstance(BrotliDecoderState* s) {
  if (!s) return;
  if (s->batch) BrotliAmdBatchDestroy(s->batch);
  if (s->d_in || s->d_out) {
    DeviceGuard guard;
    if (s->device >= 0) (void)hipSetDevice(s->device);
    if (s->d_in) (void)hipFree(s->d_in);
    if (s->d_out) (void)hipFree(s->d_out);
  }
  st_free(s, s->outq);
  brotli_free_func f = s->free_func; void* opaque = s->opaque;
  if (f) f(opaque, s); else std::free(s);
}

extern "C" BROTLI_BOOL BrotliDecoderSetParameter(BrotliDecoderState* s, BrotliDecoderParameter param, uint32_t value) {
  if (!s) return BROTLI_FALSE;
  if (s->used) return BROTLI_FALSE;  // only in the UNINITED state (ffi/mod.rs:163-166)
  switch (param) {
    case BROTLI_DECODER_PARAM_DISABLE_RING_BUFFER_REALLOCATION: s->canny = (value == 0); return BROTLI_TRUE;
    case BROTLI_DECODER_PARAM_LARGE_WINDOW: s->large_window = (value != 0); return BROTLI_TRUE;
  }
  return BROTLI_TRUE;
}

extern "C" BrotliDecoderResult BrotliDecoderDecompressStream(BrotliDecoderState* s, size_t* available_in, const uint8_t** next_in,
                                                             size_t* available_out, uint8_t** next_out, size_t* total_out) {
  if (!s || !available_in || !next_in || !available_out || !next_out) {  // ffi/mod.rs:397-407
    if (s) s->error_code = BROTLI_DECODER_ERROR_INVALID_ARGUMENTS;
    return BROTLI_DECODER_RESULT_ERROR;
  }
  if (!valid_slice(*next_in, *available_in) || !valid_slice(*next_out, *available_out)) {
    s->error_code = BROTLI_DECODER_ERROR_INVALID_ARGUMENTS;
    return BROTLI_DECODER_RESULT_ERROR;
  }
  if (fatal(s->error_code)) return BROTLI_DECODER_RESULT_ERROR;  // decode.rs:2796-2798
  if ((uint64_t)*available_in >= (1ull << 32)) {                 // decode.rs:2799-2801
    s->error_code = BROTLI_DECODER_ERROR_INVALID_ARGUMENTS;
    return BROTLI_DECODER_RESULT_ERROR;
  }
  // Output the decoder OWES comes first, and while it does not fit no input is consumed (a caller that sees
  // NEEDS_MORE_OUTPUT finds its input where it left it: bit_reader/mod.rs:295-306).  Owed is what the reference has to write
  // before it decodes on: the end of the stream (decode.rs:3382-3397), the bytes in front of a fatal error, and a full ring
  // buffer (decode.rs:1693-1738) -- this decoder keeps no ring, so "a window's worth not yet taken" stands for that.  What a
  // call that ended in NEEDS_MORE_INPUT had no room for is NOT owed: the reference wrote what fitted, took the call's input
  // and kept the rest for later calls (decode.rs:2835-2846), and so does this.
  if (s->outq_len != s->outq_off) {
    const uint64_t ring = (s->have_resume && s->resume.window_bits) ? (1ull << s->resume.window_bits) : ~0ull;
    if (s->finished || s->pending_error || (uint64_t)(s->outq_len - s->outq_off) >= ring) {
      size_t n0 = hand_over(s, *next_out, *available_out);
      *next_out += n0; *available_out -= n0;
      if (s->outq_len != s->outq_off) {
        if (total_out) *total_out = (size_t)s->total_out;
        s->error_code = BROTLI_DECODER_NEEDS_MORE_OUTPUT;
        return BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT;
      }
    }
  }
  const size_t given = *available_in;
  if (!s->finished && !s->pending_error && given) {
    DeviceGuard guard;
    // lazily bind to the current device
    if (!s->batch) {
      int dev = 0;
      if (!current_device(&dev)) { set_runtime_error(s, "HIP device unavailable"); return BROTLI_DECODER_RESULT_ERROR; }
      s->batch = BrotliAmdBatchCreate(1, 0, 0);
      if (!s->batch) { set_runtime_error(s, "HIP device unavailable"); return BROTLI_DECODER_RESULT_ERROR; }
      s->device = dev;
    }
    if (!hip_ok(hipSetDevice(s->device), "hipSetDevice")) { set_runtime_error(s, "HIP runtime failure"); return BROTLI_DECODER_RESULT_ERROR; }
    const size_t fill = (size_t)(s->d_in_len - s->in_base);
    if (!dev_rebase(&s->d_in, &s->d_in_cap, 0, fill, fill + given) ||
        !hip_ok(hipMemcpy(s->d_in + fill, *next_in, given, hipMemcpyHostToDevice), "hipMemcpy(input)")) {
      set_runtime_error(s, "HIP runtime failure");
      return BROTLI_DECODER_RESULT_ERROR;
    }
    s->d_in_len += given;
    *next_in += given; *available_in = 0;
    s->used = true;
    BrotliAmdStreamStatus st;
    if (int e = decode_pass(s, &st)) {
      if (e == 2) { s->error_code = BROTLI_DECODER_ERROR_ALLOC_RING_BUFFER_2; return BROTLI_DECODER_RESULT_ERROR; }
      set_runtime_error(s, "HIP runtime failure");
      return BROTLI_DECODER_RESULT_ERROR;
    }
    if (st.result == BROTLI_DECODER_RESULT_SUCCESS) {
      s->finished = true;
      // give back what lies beyond the end of the stream (decode.rs:3374-3378); it is part of this call's input
      size_t unused = (size_t)(s->d_in_len - st.consumed);
      if (unused > given) unused = given;
      *next_in -= unused; *available_in += unused;
    } else if (st.result == BROTLI_DECODER_RESULT_ERROR) {
      s->pending_error = st.error_code;
    } else if (!trim_buffers(s)) {
      set_runtime_error(s, "HIP runtime failure");
      return BROTLI_DECODER_RESULT_ERROR;
    }
  }
  size_t n = hand_over(s, *next_out, *available_out);
  *next_out += n; *available_out -= n;
  if (total_out) *total_out = (size_t)s->total_out;
  if (s->outq_len != s->outq_off && (s->finished || s->pending_error)) { s->error_code = BROTLI_DECODER_NEEDS_MORE_OUTPUT; return BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT; }
  if (s->pending_error) { s->error_code = s->pending_error; return BROTLI_DECODER_RESULT_ERROR; }
  if (s->finished) { s->error_code = BROTLI_DECODER_SUCCESS; return BROTLI_DECODER_RESULT_SUCCESS; }
  s->error_code = BROTLI_DECODER_NEEDS_MORE_INPUT;
  return BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT;
}

extern "C" BrotliDecoderResult BrotliDecoderDecompressStreaming(BrotliDecoderState* s, size_t* available_in, const uint8_t* next_in,
                                                                size_t* available_out, uint8_t* next_out) {
  return BrotliDecoderDecompressStream(s, available_in, &next_in, available_out, &next_out, nullptr);
}

extern "C" BROTLI_BOOL BrotliDecoderHasMoreOutput(const BrotliDecoderState* s) {
  if (!s || fatal(s->error_code)) return BROTLI_FALSE;  // decode.rs:2259-2263
  return s->outq_len != s->outq_off ? BROTLI_TRUE : BROTLI_FALSE;
}

extern "C" const uint8_t* BrotliDecoderTakeOutput(BrotliDecoderState* s, size_t* size) {
  if (!s || !size) return nullptr;
  size_t want = *size ? *size : ((size_t)1 << 24);  // decode.rs:2273
  if (fatal(s->error_code) || s->outq_len == s->outq_off) { *size = 0; return nullptr; }
  size_t n = std::min(want, s->outq_len - s->outq_off);
  const uint8_t* p = s->outq + s->outq_off;
  s->outq_off += n; s->total_out += n;
  *size = n;
  return p;  // valid until the next call on this instance
}

extern "C" uint64_t BrotliAmdDecoderDeviceCommands(const BrotliDecoderState* s) { return s ? s->device_commands : 0; }
extern "C" BROTLI_BOOL BrotliDecoderIsUsed(const BrotliDecoderState* s) { return (s && s->used) ? BROTLI_TRUE : BROTLI_FALSE; }
extern "C" BROTLI_BOOL BrotliDecoderIsFinished(const BrotliDecoderState* s) {
  return (s && s->finished && !s->pending_error && s->outq_len == s->outq_off) ? BROTLI_TRUE : BROTLI_FALSE;
}
extern "C" BrotliDecoderErrorCode BrotliDecoderGetErrorCode(const BrotliDecoderState* s) {
  return s ? (BrotliDecoderErrorCode)s->error_code : BROTLI_DECODER_ERROR_INVALID_ARGUMENTS;
}
extern "C" const char* BrotliDecoderGetErrorString(const BrotliDecoderState* s) {
  if (s && s->has_error_text) return s->error_text;  // ffi/mod.rs:571-580
  return BrotliDecoderErrorString(BrotliDecoderGetErrorCode(s));
}

extern "C" uint8_t* BrotliDecoderMallocU8(BrotliDecoderState* s, size_t size) { return s ? static_cast<uint8_t*>(st_alloc(s, size)) : nullptr; }
extern "C" void BrotliDecoderFreeU8(BrotliDecoderState* s, uint8_t* data, size_t) { if (s) st_free(s, data); }
extern "C" size_t* BrotliDecoderMallocUsize(BrotliDecoderState* s, size_t size) {
  if (!s || size > SIZE_MAX / sizeof(size_t)) return nullptr;  // ffi/mod.rs:507-510
  return static_cast<size_t*>(st_alloc(s, size * sizeof(size_t)));
}
extern "C" void BrotliDecoderFreeUsize(BrotliDecoderState* s, size_t* data, size_t) { if (s) st_free(s, data); }

// Debug aid for tests/ (not part of the public headers): device bytes a streaming instance holds at the moment.
extern "C" __attribute__((visibility("default"))) size_t brotli_amd_debug_stream_device_bytes(const BrotliDecoderState* s) {
  return s ? s->d_in_cap + s->d_out_cap : 0;
}

// Debug/profiling aid for tools/ (not part of the public headers): raw status block of stream i after Wait.
extern "C" __attribute__((visibility("default"))) const BrotliAmdStreamStatus* brotli_amd_debug_status(BrotliAmdBatch* b, uint32_t i) {
  return (b && i < b->n) ? &b->h_status[i] : nullptr;
}
