// brotli_scan_engine.h -- the two-pass command engine (included by brotli_kernels.hip inside its namespace).
//
// What it replaces: the serial walk of src/decode.rs:2359-2726 (ProcessCommandsInternal) over a metablock whose
// literals do not depend on context (every literal block type has a constant context map: DetectTrivialLiteralBlockTypes,
// decode.rs:1525-1553), for blocks of sixteen waves that own one stream.
//
// The position of command k + 1 is only known once command k has been parsed, and one wave parses a command in a few
// thousand clocks.  The engine breaks that chain the brute-force way the hardware is good at: it parses a command at
// EVERY bit position of the stream, 64 positions per wave instruction, sixteen waves at a time, and only then follows
// the one chain of positions that is real.
//
//   pass 1, all waves, per step of SC_N stream bits (bulk-synchronous, phases separated by barriers):
//     S / J1    the literal a prefix-code word starting at this bit would be, and its length        (decode.rs:378-398)
//     J2 .. J32 binary lifting: where the 2nd, 4th, ... 32nd literal after this bit would start
//     REC       the command that would start at this bit: command symbol + extra bits (decode.rs:2134-2189), its
//               insert_len literals skipped through J*, its distance symbol + extra bits (decode.rs:2066-2131),
//               packed with the bit distance to the next command
//     D2 / D4   bits from this bit to the second / fourth command after the one that would start here
//   pass 2:
//     walk      wave 0 follows the stream's real chain through D4 / D2 / REC: one LDS round trip per four commands, the
//               next hop's words asked for before the current hop is written down; the lanes behind such an anchor find
//               their own command (64 entries a batch, up to SC_GROUP batches a tick)
//     resolve   wave 0, lane = command: output offsets (prefix sums), block counts, the distance ring
//               (decode.rs:2017-2049) and every limit the reference checks; the first command that needs anything
//               unusual (dictionary word, invalid distance, block switch, end of the metablock / output / ring segment)
//               ends the engine's part in front of it and the checked command loop takes over for that command
//     execute   all waves: literals out of S through J*, and the LZ77 copies (decode.rs:2641-2680) whose source lies in
//               front of the group; copies that read the group's own output (overlapping ones among them, as pattern
//               fills) are done afterwards, in order, by the last wave
//   The passes overlap: while the other fifteen waves build REC for step s, wave 0 walks and resolves the region of
//   step s - 1 (REC, D2 and D4 are double-buffered), and what it posts is executed by everyone at the start of step s + 1.
//
// Everything lives in LDS rings indexed by stream bit position (mod SC_R); what a phase reads is always behind the
// frontier of the phase that produces it (the lags below).  Results never depend on the speculation: a REC entry is
// either the exact parse of the command that starts there or zero ("walk it by hand").
#pragma once

constexpr uint32_t SC_WAVES = 16;                 // waves of a block that runs the engine
#ifndef BROTLI_AMD_SCAN_R
#define BROTLI_AMD_SCAN_R 8192
#endif
constexpr uint32_t SC_R = BROTLI_AMD_SCAN_R;      // ring size in stream bits
constexpr uint32_t SC_M = SC_R - 1;
constexpr uint32_t SC_N = SC_R / 4 - 128;         // bits a step advances by: 30 windows of 64 (two per wave for the fifteen waves that build REC)
constexpr uint32_t SC_N2 = SC_R / 2;              // REC / D2 / D4 rings: two steps (the one being built, the one being walked), a power of two
constexpr uint32_t SC_IN_DW = 2 * SC_R / 32;      // input ring in dwords (+ 2 mirrored at the end): it runs one step ahead of the tables
constexpr uint32_t SC_GROUP = 2;                  // batches wave 0 may post per tick
// frontier lags (bits): J2/J4[b] read J1 up to b + 45, J8/J16[b] read J4 up to b + 180, J32[b] reads J16 up to b + 240;
// REC[b] reads J* up to b + 63 + 63 * 15 and the input 96 bits on
constexpr uint32_t SC_LAG_REC = 1088, SC_LAG_32 = 256, SC_LAG_16 = 192, SC_LAG_4 = 64, SC_LAG_IN = 64;
constexpr uint32_t SC_AHEAD = SC_LAG_REC + SC_LAG_32 + SC_LAG_16 + SC_LAG_4 + SC_LAG_IN;  // input frontier - REC frontier
// what is executed at the start of step s + 1 was walked in step s out of the region of step s - 1, while S / J1 of step
// s + 1 are being written: three steps and the lags must fit the ring
static_assert(3 * SC_N + SC_AHEAD <= SC_R && 2 * SC_N <= SC_N2 && SC_N % 64 == 0, "ring too small");
static_assert(4 * SC_N + SC_AHEAD + 64 <= SC_IN_DW * 32, "input ring too small");
constexpr uint32_t SC_LONG_EXIT = 8192;           // literal runs from here on go back to the rounds of the helper waves
constexpr uint32_t SC_MIN_QUOTA = 2048;           // output bytes that must be possible for the engine to start
constexpr uint32_t SC_MIN_INPUT = 2 * SC_N + SC_AHEAD + 64;  // stream bits that must be left for the engine to start

// LDS layout, offsets from the engine's base
constexpr uint32_t SC_CTL = 0;                    // 256: control words
constexpr uint32_t SC_IN = 256;                   // input ring: SC_IN_DW + 2 dwords
constexpr uint32_t SC_S = SC_IN + (SC_IN_DW + 2) * 4 + 8;   // literal at every bit
constexpr uint32_t SC_J1 = SC_S + SC_R;           // code length at every bit
constexpr uint32_t SC_J2 = SC_J1 + SC_R;
constexpr uint32_t SC_J4 = SC_J2 + SC_R;
constexpr uint32_t SC_J8 = SC_J4 + SC_R;
constexpr uint32_t SC_J16 = SC_J8 + SC_R;
constexpr uint32_t SC_J32 = SC_J16 + SC_R;        // u16
constexpr uint32_t SC_REC = SC_J32 + 2 * SC_R;    // 8 bytes per bit of two steps
constexpr uint32_t SC_D24 = SC_REC + 8 * SC_N2;   // u32 per bit: D2 = bits to the command after next (low half), D4 = bits to the fourth command from here (high half); 0: not known
constexpr uint32_t SC_XL = SC_D24 + 4 * SC_N2;    // 2 x SC_GROUP x 64 x 16: the group being executed and the one being posted
constexpr uint32_t SC_BYTES = SC_XL + 2 * SC_GROUP * 1024;
static_assert(SC_S % 16 == 0 && SC_REC % 16 == 0 && SC_XL % 16 == 0, "alignment");
static_assert(SC_WAVES * HL_SLOT <= 8 * SC_N2, "the literal rounds' slots of a sixteen-wave block lie on REC");

// control words: the invocation's parameters (written by the decoding wave before the others join), then what wave 0
// posts per tick: a group of up to SC_GROUP batches (entries in SC_XL, per batch its entry count and output position) and
// two flags -- the walk stopped short of its limit because the group was full; the engine's part ends with this group
enum { SCC_BASE_DW = 4, SCC_IN_LIMIT = 5, SCC_LIT_TREE = 6, SCC_CMD_TREE = 7, SCC_DT0 = 8, SCC_POSTFIX = 12, SCC_NUM_DIRECT = 13,
       SCC_OUT_LO = 14, SCC_OUT_HI = 15, SCC_ENTRY = 16, SCC_FLAGS = 18, SCC_DICT_LO = 20, SCC_DICT_HI = 21 /* the static dictionary (path engine) */,
       // the two group buffers (posted in even / odd ticks): words from SCC_GRP + 16 * parity
       SCC_GRP = 32, SCG_NG = 0, SCG_ANYDEP = 1, SCG_K = 2 /* + batch */, SCG_P = 6 /* + 2 * batch */ };
enum { SCF_BEHIND = 1, SCF_LEAVE = 2 };
enum { SCK_NONE = 0, SCK_EXPLICIT = 1, SCK_SHORT = 2, SCK_IMPLICIT = 3 };

__device__ __forceinline__ uint32_t sc_ctl_ld(uint32_t sb, uint32_t k) { return rfl(*reinterpret_cast<lds_vu32*>(&g_smem[sb + SC_CTL + 4u * k])); }
__device__ __forceinline__ void sc_ctl_st(uint32_t sb, uint32_t k, uint32_t v) { if (lane_id() == 0) *reinterpret_cast<lds_vu32*>(&g_smem[sb + SC_CTL + 4u * k]) = v; }

#ifdef BROTLI_AMD_PROFILE_SCAN
__device__ unsigned long long g_scan_prof[28];  // ([24]: wave 1's ticks in REC, [25] / [26]: the last wave's dependent copies)
#define SCAN_PROF(k) do { if (me == 0) { uint64_t _t = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0) sp_acc[k] += _t - sp_t; sp_t = _t; } } while (0)
#define SCAN_COUNT(k, v) do { if (me == 0 && blockIdx.x == 0) sp_acc[k] += (v); } while (0)
#else
#define SCAN_PROF(k) do { } while (0)
#define SCAN_COUNT(k, v) do { } while (0)
#endif

// Between the stages of two chains of dependent LDS reads written side by side: the machine scheduler otherwise puts each
// chain back together (one chain's reads, waits and all, then the other's), and the round trips no longer overlap.
#ifdef BROTLI_AMD_SCAN_NO_STAGES
#define SC_STAGE() do { } while (0)
#else
#define SC_STAGE() __builtin_amdgcn_sched_barrier(0)
#endif

// inclusive prefix sum over the wave (LLVM's buildScan: row shifts inside rows of 16, then row broadcasts)
__device__ __forceinline__ uint32_t sc_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ uint32_t sc_msb64(uint64_t m) { return 63u - (uint32_t)__clzll((long long)m); }  // m != 0

// the 64 stream bits from bit p (relative to the engine's origin), out of the input ring
__device__ __forceinline__ void sc_bits64(uint32_t sb, uint32_t p, uint32_t& lo, uint32_t& hi) {
  const uint32_t q = sb + SC_IN + (((p >> 5) & (SC_IN_DW - 1u)) << 2), sh = p & 31u;
  const uint32_t a0 = lds_ld32(q), a1 = lds_ld32(q + 4u), a2 = lds_ld32(q + 8u);
  lo = __builtin_amdgcn_alignbit(a1, a0, sh);
  hi = __builtin_amdgcn_alignbit(a2, a1, sh);
}
__device__ __forceinline__ uint32_t sc_bits32(uint32_t sb, uint32_t p) {
  const uint32_t q = sb + SC_IN + (((p >> 5) & (SC_IN_DW - 1u)) << 2);
  return __builtin_amdgcn_alignbit(lds_ld32(q + 4u), lds_ld32(q), p & 31u);
}
// two-level table lookup (entry layout of read_symbol): symbol and code length of the word that starts x
__device__ __forceinline__ void sc_lookup(uint32_t tree_addr, uint32_t x, uint32_t& sym, uint32_t& len) {
  uint32_t e = lds_ld16(tree_addr + ((x & 0xFFu) << 1));
  uint32_t L = e & 15u;
  if (L > ROOT_BITS) {
    const uint32_t idx = (e >> 4) + __builtin_amdgcn_ubfe(x, ROOT_BITS, L - ROOT_BITS);
    e = lds_ld16(tree_addr + (idx << 1));
    L = ROOT_BITS + (e & 15u);
  }
  sym = e >> 4; len = L;
}
// position after n (< 64) literals from p
__device__ __forceinline__ uint32_t sc_skip(uint32_t sb, uint32_t p, uint32_t n) {
  if (n & 1u) p += lds_ld8(sb + SC_J1 + (p & SC_M));
  if (n & 2u) p += lds_ld8(sb + SC_J2 + (p & SC_M));
  if (n & 4u) p += lds_ld8(sb + SC_J4 + (p & SC_M));
  if (n & 8u) p += lds_ld8(sb + SC_J8 + (p & SC_M));
  if (n & 16u) p += lds_ld8(sb + SC_J16 + (p & SC_M));
  if (n & 32u) p += lds_ld16(sb + SC_J32 + ((p & SC_M) << 1));
  return p;
}

// Head of the command whose first bits are lo/hi: ReadCommandInternal, decode.rs:2134-2189 (kCmdLut regenerated as in
// process_commands).  Works per lane; the walker calls it with the same position in every lane.
struct ScHead { uint32_t bits, insert, copy, implicit, dctx; };
__device__ __forceinline__ ScHead sc_head(uint32_t lo, uint32_t hi, uint32_t cmd_tree_addr, uint32_t lut_vgpr) {
  uint32_t cmd, L;
  sc_lookup(cmd_tree_addr, lo, cmd, L);
  const uint32_t cell = cmd >> 6;
  const uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u);
  const uint32_t copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);
  const uint32_t ie = bperm(ins_code << 2, lut_vgpr), ce = bperm((32u + copy_code) << 2, lut_vgpr);
  uint64_t w = (((uint64_t)hi << 32) | lo) >> L;
  const uint32_t ib = ie >> 16, cb = ce >> 16;
  ScHead h;
  h.insert = (ie & 0xFFFFu) + ((uint32_t)w & ((1u << ib) - 1u));
  w >>= ib;
  h.copy = (ce & 0xFFFFu) + ((uint32_t)w & ((1u << cb) - 1u));
  h.bits = L + ib + cb;
  h.implicit = cmd < 128u ? 1u : 0u;
  h.dctx = copy_code > 2u ? 3u : copy_code;
  return h;
}
// Distance symbol + extra bits at lo/hi: ReadDistanceInternal, decode.rs:2066-2131 (no large window: at most 24 extra bits)
struct ScDist { uint32_t kind, val, bits; };
// (what follows the lookup of the distance symbol `code`, a word of L bits)
__device__ __forceinline__ ScDist sc_dist_finish(uint32_t code, uint32_t L, uint32_t lo, uint32_t hi, uint32_t postfix_bits, uint32_t num_direct) {
  ScDist d;
  if (code < 16u) { d.kind = SCK_SHORT; d.val = code; d.bits = L; return d; }
  int32_t distval = (int32_t)code - (int32_t)num_direct;
  uint32_t dc = code, nbits = 0;
  if (distval >= 0) {
    const uint32_t postfix = (uint32_t)distval & ((1u << postfix_bits) - 1u);
    distval >>= postfix_bits;
    nbits = ((uint32_t)distval >> 1) + 1u;
    const uint32_t extra = (uint32_t)((((uint64_t)hi << 32) | lo) >> L) & ((1u << (nbits & 31u)) - 1u);
    const uint32_t offset = (((uint32_t)distval & 1u) + 2u) << (nbits & 31u);
    dc = ((offset - 4u + extra) << postfix_bits) + postfix + num_direct;
  }
  d.kind = SCK_EXPLICIT; d.val = dc - 16u + 1u; d.bits = L + nbits;
  // (a large-window stream's distance codes go up to 62 extra bits -- decode.rs:152-187 --: one with more than 24 is a
  // distance no window of 2^30 holds, or an invalid one; its length is right, its value says "the checked loop's")
  if (nbits > 24u) d.val = 1u << 30;
  return d;
}
__device__ __forceinline__ ScDist sc_dist(uint32_t lo, uint32_t hi, uint32_t dtree_addr, uint32_t postfix_bits, uint32_t num_direct) {
  uint32_t code, L;
  sc_lookup(dtree_addr, lo, code, L);
  return sc_dist_finish(code, L, lo, hi, postfix_bits, num_direct);
}

// How the engine hands the stream back (state in LDS_LEAN, like lean_commands)
enum { SCX_BEGIN = 0, SCX_LITERALS_REST = 1, SCX_POST_DISTANCE = 2 };
enum { L_SC_POS_LO = L_SPEC_LO, L_SC_POS_HI = L_SPEC_HI };  // (the literal-scratch address is not needed while the engine runs: restored by the caller)

// One invocation: every wave of the block calls it (wave 0 from process_commands, the others from helper_wave).
// Returns (wave 0) the number of commands it took; exit form and state in LDS_LEAN.
__device__ __noinline__ uint32_t scan_engine(const uint32_t me_) {
  const uint32_t lane = lane_id();
  const uint32_t me = rfl(me_);
  const uint32_t sb = hc_ld(HC_SCAN_BASE);
  __syncthreads();  // the parameters are in place
#ifdef BROTLI_AMD_PROFILE_SCAN
  uint64_t sp_acc[24] = {}; uint64_t sp_t = __builtin_amdgcn_s_memtime();
#endif
  const uint32_t lit_tree = sc_ctl_ld(sb, SCC_LIT_TREE), cmd_tree = sc_ctl_ld(sb, SCC_CMD_TREE);
  const uint32_t dt0 = sc_ctl_ld(sb, SCC_DT0), dt1 = sc_ctl_ld(sb, SCC_DT0 + 1), dt2 = sc_ctl_ld(sb, SCC_DT0 + 2), dt3 = sc_ctl_ld(sb, SCC_DT0 + 3);
  const uint32_t postfix_bits = sc_ctl_ld(sb, SCC_POSTFIX), num_direct = sc_ctl_ld(sb, SCC_NUM_DIRECT);
  const uint32_t base_dw = sc_ctl_ld(sb, SCC_BASE_DW), in_limit = sc_ctl_ld(sb, SCC_IN_LIMIT);
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)sc_ctl_ld(sb, SCC_OUT_LO) | ((uint64_t)sc_ctl_ld(sb, SCC_OUT_HI) << 32));
  gcu32* const in_dw = BitReader::base() + base_dw;
  uint32_t lut_vgpr = 0;  // insert / copy code LUT image, as in the kernel
  if (lane < 24) lut_vgpr = (uint32_t)kInsBase[lane] | ((uint32_t)kInsExtra[lane] << 16);
  else if (lane >= 32 && lane < 56) lut_vgpr = (uint32_t)kCopyBase[lane - 32] | ((uint32_t)kCopyExtra[lane - 32] << 16);

  // ---- wave 0: the stream's state (uniform) ----
  uint32_t b = sc_ctl_ld(sb, SCC_ENTRY);  // next command (bits from the engine's origin)
  uint64_t P = 0; uint32_t quota = 0, bl0 = 0, bl1 = 0, bl2 = 0, ncmd = 0; int32_t mlen = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, max_backward = 0;
  if (me == 0) {
    P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
    quota = LEAN_LD(L_QUOTA); mlen = (int32_t)LEAN_LD(L_MLEN);
    bl0 = LEAN_LD(L_BL0); bl1 = LEAN_LD(L_BL1); bl2 = LEAN_LD(L_BL2);
    d0 = (int32_t)LEAN_LD(L_D0); d1 = (int32_t)LEAN_LD(L_D1); d2 = (int32_t)LEAN_LD(L_D2); d3 = (int32_t)LEAN_LD(L_D3);
    max_backward = (int32_t)LEAN_LD(L_MAX_BACKWARD);
    sc_ctl_st(sb, SCC_FLAGS, 0u);
    sc_ctl_st(sb, SCC_GRP + SCG_NG, 0u); sc_ctl_st(sb, SCC_GRP + SCG_ANYDEP, 0u); sc_ctl_st(sb, SCC_GRP + 16u + SCG_NG, 0u); sc_ctl_st(sb, SCC_GRP + 16u + SCG_ANYDEP, 0u);
  }
  // a literal run being walked by hand (commands REC does not hold): literals still to skip, then the distance and the copy
  bool in_run = false; uint32_t run_p = 0, run_rem = 0, run_copy = 0, run_implicit = 0, run_dctx = 0;
  uint32_t exit_form = SCX_BEGIN, exit_copy = 0, exit_dctx = 0; int32_t exit_dcode = 0;
  uint32_t exit_why = 0;  // (profiling) 0 input ends, 1 block counts / output limits, 2 distance, 3 a command to walk by hand that does not fit, 4 long literal run
  (void)exit_why;

  // frontiers (bits from the origin): below them the ring holds valid entries
  uint32_t f_1 = 0, f_4 = 0, f_16 = 0, f_32 = 0, f_rec = 0, f_d = 0;
  const uint32_t limit_dw = (in_limit + 31u) >> 5;  // dwords of the input that may be read
  // input of the first step (later steps find theirs in the ring: it is fetched one step ahead)
  if (SC_N + SC_AHEAD <= in_limit)
    for (uint32_t i = threadIdx.x; i < ((SC_N + SC_AHEAD) >> 5); i += 64u * SC_WAVES) {
      const uint32_t v = in_dw[i];
      lds_st32(sb + SC_IN + (i << 2), v);
      if (i < 2u) lds_st32(sb + SC_IN + ((SC_IN_DW + i) << 2), v);
    }
  __syncthreads();

  // A tick is either a STEP (the tables advance by SC_N bits; meanwhile wave 0 walks the region completed in the step
  // before), a SYNC tick (no tables: wave 0 catches up with everything that is complete) or the FINAL one (the last
  // group is executed).  Every tick starts by executing the group posted in the tick before.
  enum { M_STEP = 0, M_SYNC = 1, M_FINAL = 2 };
  uint32_t mode = (SC_N + SC_AHEAD <= in_limit) ? (uint32_t)M_STEP : (uint32_t)M_FINAL;
  uint32_t par = 0;  // the group buffer this tick posts into (the other one holds what the tick before posted)
  for (;; par ^= 1u) {
    const uint32_t gx = SCC_GRP + 16u * (par ^ 1u), xl_exec = sb + SC_XL + (par ^ 1u) * (SC_GROUP * 1024u);  // the group to execute
    const uint32_t gp = SCC_GRP + 16u * par, xl_post = sb + SC_XL + par * (SC_GROUP * 1024u);                // the group to post
    SCAN_COUNT(14, mode == M_STEP ? 1u : 0u); SCAN_COUNT(15, mode == M_SYNC ? 1u : 0u); SCAN_COUNT(11, mode == M_FINAL ? 1u : 0u);
    // ================= part 1 (all waves): the posted group =================
    {
      const uint32_t ng = sc_ctl_ld(sb, gx + SCG_NG);
      for (uint32_t g = 0; g < ng; g++) {
        const uint32_t k_exec = sc_ctl_ld(sb, gx + SCG_K + g);
        gu8* const o = out + ((uint64_t)sc_ctl_ld(sb, gx + SCG_P + 2u * g) | ((uint64_t)sc_ctl_ld(sb, gx + SCG_P + 2u * g + 1u) << 32));
        const uint32_t xb = xl_exec + g * 1024u;
        // Wave w takes entries w, w + 16, ...: the loads of their copies first (copies whose source lies in front of the
        // group), then their literals (lane i the i-th literal of the entry: out of S through J*, the entries' chains side
        // by side), then the copies' stores.  Two entries at a time where the batch has at most 32, else four.
#define SC_EXEC(NS) do { \
        uint32_t hold[NS], x0[NS], cn[NS], off[NS], q[NS], nl[NS]; \
        u32x4 xe[NS]; \
        _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {  /* the entries: one 16-byte read each, all under way before the first is looked at */ \
          const uint32_t k = me + t * SC_WAVES; \
          xe[t] = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[xb + ((k < k_exec ? k : 0u) << 4)]); \
        } \
        _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { \
          const uint32_t k = me + t * SC_WAVES; \
          x0[t] = rfl(xe[t].x); cn[t] = rfl(xe[t].y); off[t] = rfl(xe[t].w); hold[t] = 0; \
          const uint32_t dist = rfl(xe[t].z); \
          if (k >= k_exec) { x0[t] = 0u; cn[t] = 0u; } \
          if (cn[t] == 0u || (x0[t] >> 31) != 0u) { cn[t] = 0u; continue; } \
          gu8* const dst = o + off[t] + ((x0[t] >> 16) & 63u); gu8* const src = dst - dist; \
          if (cn[t] <= 64u) { if (lane < cn[t]) hold[t] = src[lane]; } \
          else { \
            const uint32_t n16 = cn[t] >> 4; \
            for (uint32_t c = lane; c < n16; c += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16); \
            const uint32_t tail = n16 << 4; \
            if (tail + lane < cn[t]) dst[tail + lane] = src[tail + lane]; \
            cn[t] = 0u; \
          } \
        } \
        uint32_t nmax = 0; \
        _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { q[t] = x0[t] & SC_M; nl[t] = (x0[t] >> 16) & 63u; nmax = nl[t] > nmax ? nl[t] : nmax; } \
        if (nmax != 0u) { \
          SC_XHOP(NS, 1u, lds_ld8(sb + SC_J1 + (q[t] & SC_M))) \
          if (nmax > 2u) { SC_XHOP(NS, 2u, lds_ld8(sb + SC_J2 + (q[t] & SC_M))) } \
          if (nmax > 4u) { SC_XHOP(NS, 4u, lds_ld8(sb + SC_J4 + (q[t] & SC_M))) } \
          if (nmax > 8u) { SC_XHOP(NS, 8u, lds_ld8(sb + SC_J8 + (q[t] & SC_M))) } \
          if (nmax > 16u) { SC_XHOP(NS, 16u, lds_ld8(sb + SC_J16 + (q[t] & SC_M))) } \
          if (nmax > 32u) { SC_XHOP(NS, 32u, lds_ld16(sb + SC_J32 + ((q[t] & SC_M) << 1))) } \
          uint32_t sy[NS]; \
          _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) sy[t] = lds_ld8(sb + SC_S + (q[t] & SC_M)); \
          SC_STAGE(); \
          _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) if (lane < nl[t]) o[off[t] + lane] = (uint8_t)sy[t]; \
        } \
        _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) \
          if (cn[t] != 0u && lane < cn[t]) o[off[t] + nl[t] + lane] = (uint8_t)hold[t]; \
      } while (0)
        // (a hop: the reads of all entries first, then the additions -- one LDS round trip per level, not one per entry and
        // level; lanes that do not hop read somewhere inside the ring and drop what they get)
#define SC_XHOP(NS, BIT, EXPR) { uint32_t hop_[NS]; _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) hop_[t] = EXPR; \
        SC_STAGE(); \
        _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) q[t] += ((lane & BIT) != 0u) ? hop_[t] : 0u; }
        if (k_exec <= 2u * SC_WAVES) { if (me < k_exec) SC_EXEC(2u); }
        else SC_EXEC(4u);
#undef SC_XHOP
#undef SC_EXEC
      }
    }
    SCAN_PROF(6);
    if (f_d < f_rec) {
      // D2 / D4 of the region whose REC the tick before completed: bits from every bit to the second and the fourth command
      // after the one that would start there (0 where one of them is not in REC or lies beyond the region)
      for (uint32_t w0 = (f_d >> 6) + me; w0 < (f_rec >> 6); w0 += 2u * SC_WAVES) {
        uint32_t p[2], d1[2], d2[2], d3[2], d4[2];
        p[0] = (w0 << 6) + lane; p[1] = ((w0 + SC_WAVES < (f_rec >> 6) ? w0 + SC_WAVES : w0) << 6) + lane;
#define SC_DLT(pos) (lds_ld32(sb + SC_REC + (((pos) & (SC_N2 - 1u)) << 3)) & 0x7FFu)
        // (the reads of a stage unconditional -- a lane without a command there reads its own entry again and drops it --, so
        // that the two windows' reads of a stage go out together)
        uint32_t t_[2];
        _Pragma("unroll") for (int u = 0; u < 2; u++) d1[u] = SC_DLT(p[u]);
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) { const bool on = d1[u] != 0u && p[u] + d1[u] < f_rec; t_[u] = SC_DLT(on ? p[u] + d1[u] : p[u]); t_[u] = on ? t_[u] : 0u; }
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) d2[u] = t_[u] ? d1[u] + t_[u] : 0u;
        _Pragma("unroll") for (int u = 0; u < 2; u++) { const bool on = d2[u] != 0u && p[u] + d2[u] < f_rec; t_[u] = SC_DLT(on ? p[u] + d2[u] : p[u]); t_[u] = on ? t_[u] : 0u; }
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) d3[u] = t_[u] ? d2[u] + t_[u] : 0u;
        _Pragma("unroll") for (int u = 0; u < 2; u++) { const bool on = d3[u] != 0u && p[u] + d3[u] < f_rec; t_[u] = SC_DLT(on ? p[u] + d3[u] : p[u]); t_[u] = on ? t_[u] : 0u; }
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) d4[u] = t_[u] ? d3[u] + t_[u] : 0u;
#undef SC_DLT
        _Pragma("unroll") for (int u = 0; u < 2; u++) {
          lds_st32(sb + SC_D24 + ((p[u] & (SC_N2 - 1u)) << 2), d2[u] | (d4[u] << 16));
        }
      }
      f_d = f_rec;
    }
    SCAN_PROF(7);
    uint32_t walk_limit = f_rec;  // REC, D2 and D4 are complete below this (once the barrier in front of the walk is passed)
    uint32_t pre_v = 0, pre_i = 0; bool pre_ok = false;
    if (mode == M_STEP) {
      // ================= pass 1: one step =================
      const uint32_t e_rec = f_rec + SC_N;
      const uint32_t t_32 = e_rec + SC_LAG_REC, t_16 = t_32 + SC_LAG_32, t_4 = t_16 + SC_LAG_16, t_1 = t_4 + SC_LAG_4, t_in = t_1 + SC_LAG_IN;
      // the next step's input: requested now, put into the ring at the end of this step's tables
      pre_i = (t_in >> 5) + threadIdx.x;
      pre_ok = threadIdx.x < (SC_N >> 5) && pre_i < limit_dw;
      if (pre_ok) pre_v = in_dw[pre_i];
      // S / J1: the literal that would start at every bit, and its length -- two windows per wave and pass
      for (uint32_t w0 = (f_1 >> 6) + me; w0 < (t_1 >> 6); w0 += 2u * SC_WAVES) {
        uint32_t p[2], x[2], e[2], L[2];
        p[0] = (w0 << 6) + lane; p[1] = ((w0 + SC_WAVES < (t_1 >> 6) ? w0 + SC_WAVES : w0) << 6) + lane;
        _Pragma("unroll") for (int u = 0; u < 2; u++) x[u] = sc_bits32(sb, p[u]);
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) e[u] = lds_ld16(lit_tree + ((x[u] & 0xFFu) << 1));
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) L[u] = e[u] & 15u;
        if (__ballot(L[0] > ROOT_BITS || L[1] > ROOT_BITS) != 0ull) {
          uint32_t e2[2];
          _Pragma("unroll") for (int u = 0; u < 2; u++) {
            const bool sec = L[u] > ROOT_BITS;
            const uint32_t idx = sec ? (e[u] >> 4) + __builtin_amdgcn_ubfe(x[u], ROOT_BITS, L[u] - ROOT_BITS) : (x[u] & 0xFFu);
            e2[u] = lds_ld16(lit_tree + (idx << 1));
          }
          SC_STAGE();
          _Pragma("unroll") for (int u = 0; u < 2; u++) if (L[u] > ROOT_BITS) { e[u] = e2[u]; L[u] = ROOT_BITS + (e2[u] & 15u); }
        }
        _Pragma("unroll") for (int u = 0; u < 2; u++) { lds_st8(sb + SC_S + (p[u] & SC_M), e[u] >> 4); lds_st8(sb + SC_J1 + (p[u] & SC_M), L[u]); }
      }
      f_1 = t_1;
      SCAN_PROF(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the group's stores: in memory before anyone reads them as copy sources)
      __syncthreads();
      SCAN_PROF(9);
      // binary lifting, two levels per pass: TO2[p] = position after two FROM-hops, TO4[p] after four
#define SC_LEVELS(FROM, TO2, TO4, f_to, t_to) \
      for (uint32_t w0 = ((f_to) >> 6) + me; w0 < ((t_to) >> 6); w0 += 2u * SC_WAVES) { \
        uint32_t p[2], a1[2], a2[2], a3[2], a4[2]; \
        p[0] = (w0 << 6) + lane; p[1] = ((w0 + SC_WAVES < ((t_to) >> 6) ? w0 + SC_WAVES : w0) << 6) + lane; \
        _Pragma("unroll") for (int u = 0; u < 2; u++) a1[u] = lds_ld8(sb + FROM + (p[u] & SC_M)); \
        SC_STAGE(); \
        _Pragma("unroll") for (int u = 0; u < 2; u++) a2[u] = a1[u] + lds_ld8(sb + FROM + ((p[u] + a1[u]) & SC_M)); \
        SC_STAGE(); \
        _Pragma("unroll") for (int u = 0; u < 2; u++) a3[u] = a2[u] + lds_ld8(sb + FROM + ((p[u] + a2[u]) & SC_M)); \
        SC_STAGE(); \
        _Pragma("unroll") for (int u = 0; u < 2; u++) a4[u] = a3[u] + lds_ld8(sb + FROM + ((p[u] + a3[u]) & SC_M)); \
        SC_STAGE(); \
        _Pragma("unroll") for (int u = 0; u < 2; u++) { lds_st8(sb + TO2 + (p[u] & SC_M), a2[u]); lds_st8(sb + TO4 + (p[u] & SC_M), a4[u]); } \
      } \
      f_to = (t_to); \
      SCAN_PROF(2); \
      __syncthreads(); \
      SCAN_PROF(10);
      SC_LEVELS(SC_J1, SC_J2, SC_J4, f_4, t_4)
      SC_LEVELS(SC_J4, SC_J8, SC_J16, f_16, t_16)
#undef SC_LEVELS
      for (uint32_t w0 = (f_32 >> 6) + me; w0 < (t_32 >> 6); w0 += 2u * SC_WAVES) {
        uint32_t p[2], a[2], c[2];
        p[0] = (w0 << 6) + lane; p[1] = ((w0 + SC_WAVES < (t_32 >> 6) ? w0 + SC_WAVES : w0) << 6) + lane;
        _Pragma("unroll") for (int u = 0; u < 2; u++) a[u] = lds_ld8(sb + SC_J16 + (p[u] & SC_M));
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) c[u] = lds_ld8(sb + SC_J16 + ((p[u] + a[u]) & SC_M));
        SC_STAGE();
        _Pragma("unroll") for (int u = 0; u < 2; u++) lds_st16(sb + SC_J32 + ((p[u] & SC_M) << 1), a[u] + c[u]);
      }
      f_32 = t_32;
      SCAN_PROF(2);
      __syncthreads();
      SCAN_PROF(10);
      if (me != 0) {
#ifdef BROTLI_AMD_PROFILE_SCAN
        const uint64_t rec_t0 = __builtin_amdgcn_s_memtime();
#endif
        // REC: the command that would start at every bit of [f_rec, e_rec) -- fifteen waves, two windows per wave and pass,
        // stage by stage (wave 0 is walking the step before meanwhile)
        for (uint32_t w0 = (f_rec >> 6) + (me - 1u); w0 < (e_rec >> 6); w0 += 2u * (SC_WAVES - 1u)) {
          uint32_t p[2], lo[2], hi[2], q[2], kind[2], val[2];
          ScHead h[2];
          bool ok[2];
          p[0] = (w0 << 6) + lane; p[1] = ((w0 + (SC_WAVES - 1u) < (e_rec >> 6) ? w0 + (SC_WAVES - 1u) : w0) << 6) + lane;
          _Pragma("unroll") for (int u = 0; u < 2; u++) sc_bits64(sb, p[u], lo[u], hi[u]);
          SC_STAGE();
          {  // heads (sc_head, the two lookups side by side)
            uint32_t e[2], L[2];
            _Pragma("unroll") for (int u = 0; u < 2; u++) e[u] = lds_ld16(cmd_tree + ((lo[u] & 0xFFu) << 1));
            SC_STAGE();
            _Pragma("unroll") for (int u = 0; u < 2; u++) L[u] = e[u] & 15u;
            if (__ballot(L[0] > ROOT_BITS || L[1] > ROOT_BITS) != 0ull) {
              uint32_t e2[2];
              _Pragma("unroll") for (int u = 0; u < 2; u++) {
                const bool sec = L[u] > ROOT_BITS;
                const uint32_t idx = sec ? (e[u] >> 4) + __builtin_amdgcn_ubfe(lo[u], ROOT_BITS, L[u] - ROOT_BITS) : (lo[u] & 0xFFu);
                e2[u] = lds_ld16(cmd_tree + (idx << 1));
              }
              SC_STAGE();
              _Pragma("unroll") for (int u = 0; u < 2; u++) if (L[u] > ROOT_BITS) { e[u] = e2[u]; L[u] = ROOT_BITS + (e2[u] & 15u); }
            }
            uint32_t ie[2], ce[2], cc[2];
            _Pragma("unroll") for (int u = 0; u < 2; u++) {
              const uint32_t cmd = e[u] >> 4, cell = cmd >> 6;
              const uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u);
              cc[u] = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);
              ie[u] = bperm(ins_code << 2, lut_vgpr); ce[u] = bperm((32u + cc[u]) << 2, lut_vgpr);
              h[u].implicit = cmd < 128u ? 1u : 0u;
            }
            SC_STAGE();
            _Pragma("unroll") for (int u = 0; u < 2; u++) {
              uint64_t w = (((uint64_t)hi[u] << 32) | lo[u]) >> L[u];
              const uint32_t ib = ie[u] >> 16, cb = ce[u] >> 16;
              h[u].insert = (ie[u] & 0xFFFFu) + ((uint32_t)w & ((1u << ib) - 1u));
              w >>= ib;
              h[u].copy = (ce[u] & 0xFFFFu) + ((uint32_t)w & ((1u << cb) - 1u));
              h[u].bits = L[u] + ib + cb;
              h[u].dctx = cc[u] > 2u ? 3u : cc[u];
              ok[u] = h[u].insert < 64u && h[u].copy < 8192u;
              q[u] = p[u] + h[u].bits;
            }
          }
          // the literals are skipped (sc_skip, level by level for both)
          // (a hop: both windows' reads, then both additions; a lane that does not hop drops what it read)
#define SC_HOP(BIT, EXPR) { uint32_t hop_[2]; _Pragma("unroll") for (int u = 0; u < 2; u++) hop_[u] = EXPR; SC_STAGE(); \
          _Pragma("unroll") for (int u = 0; u < 2; u++) q[u] += (h[u].insert & BIT) ? hop_[u] : 0u; }
          SC_HOP(1u, lds_ld8(sb + SC_J1 + (q[u] & SC_M)))
          SC_HOP(2u, lds_ld8(sb + SC_J2 + (q[u] & SC_M)))
          SC_HOP(4u, lds_ld8(sb + SC_J4 + (q[u] & SC_M)))
          SC_HOP(8u, lds_ld8(sb + SC_J8 + (q[u] & SC_M)))
          SC_HOP(16u, lds_ld8(sb + SC_J16 + (q[u] & SC_M)))
          SC_HOP(32u, lds_ld16(sb + SC_J32 + ((q[u] & SC_M) << 1)))
#undef SC_HOP
          {  // distances
            uint32_t dlo[2], dhi[2];
            _Pragma("unroll") for (int u = 0; u < 2; u++) sc_bits64(sb, q[u], dlo[u], dhi[u]);
            SC_STAGE();
            ScDist d[2];
            uint32_t dta[2], de[2], dL[2];
            _Pragma("unroll") for (int u = 0; u < 2; u++) {  // (sc_dist, the two lookups side by side)
              dta[u] = h[u].dctx == 0u ? dt0 : h[u].dctx == 1u ? dt1 : h[u].dctx == 2u ? dt2 : dt3;
              de[u] = lds_ld16(dta[u] + ((dlo[u] & 0xFFu) << 1));
            }
            SC_STAGE();
            _Pragma("unroll") for (int u = 0; u < 2; u++) dL[u] = de[u] & 15u;
            if (__ballot(dL[0] > ROOT_BITS || dL[1] > ROOT_BITS) != 0ull) {
              uint32_t e2[2];
              _Pragma("unroll") for (int u = 0; u < 2; u++) {
                const bool sec = dL[u] > ROOT_BITS;
                const uint32_t idx = sec ? (de[u] >> 4) + __builtin_amdgcn_ubfe(dlo[u], ROOT_BITS, dL[u] - ROOT_BITS) : (dlo[u] & 0xFFu);
                e2[u] = lds_ld16(dta[u] + (idx << 1));
              }
              SC_STAGE();
              _Pragma("unroll") for (int u = 0; u < 2; u++) if (dL[u] > ROOT_BITS) { de[u] = e2[u]; dL[u] = ROOT_BITS + (e2[u] & 15u); }
            }
            _Pragma("unroll") for (int u = 0; u < 2; u++) d[u] = sc_dist_finish(de[u] >> 4, dL[u], dlo[u], dhi[u], postfix_bits, num_direct);
            _Pragma("unroll") for (int u = 0; u < 2; u++) {
              kind[u] = h[u].implicit ? (uint32_t)SCK_IMPLICIT : d[u].kind; val[u] = h[u].implicit ? 0u : d[u].val;
              if (!h[u].implicit) { q[u] += d[u].bits; ok[u] = ok[u] && val[u] < (1u << 26); }
            }
          }
          _Pragma("unroll") for (int u = 0; u < 2; u++) {
            const uint32_t delta = q[u] - p[u];
            const bool good = ok[u] && delta != 0u && delta < 2048u;
            const uint32_t rlo = delta | (h[u].bits << 11) | (h[u].insert << 17) | ((h[u].copy & 0x1FFu) << 23);
            const uint32_t rhi = (h[u].copy >> 9) | (kind[u] << 4) | (val[u] << 6);
            const uint32_t ra = sb + SC_REC + ((p[u] & (SC_N2 - 1u)) << 3);
            lds_st32(ra, good ? rlo : 0u);
            lds_st32(ra + 4u, good ? rhi : 0u);
          }
        }
#ifdef BROTLI_AMD_PROFILE_SCAN
        if (me == 1u && blockIdx.x == 0 && lane == 0) g_scan_prof[24] += __builtin_amdgcn_s_memtime() - rec_t0;
#endif
      }
      // (f_rec moves when the step's D2 / D4 are done, below)
    }

    // ================= part 2: wave 0 walks, resolves and posts; the last wave does the executed group's own copies =================
    // (in STEP ticks the barrier behind S / J1 has put everyone's stores of part 1 in memory; in SYNC and FINAL ticks this one does)
    if (mode != M_STEP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    if (me == SC_WAVES - 1u && sc_ctl_ld(sb, gx + SCG_ANYDEP) != 0u) {
#ifdef BROTLI_AMD_PROFILE_SCAN
      const uint64_t dep_t0 = __builtin_amdgcn_s_memtime(); uint32_t dep_n = 0;
#endif
      // copies of the executed group that read the group's own output: one after the other (a wave's stores are visible
      // to its later loads)
      const uint32_t ng = sc_ctl_ld(sb, gx + SCG_NG);
      for (uint32_t g = 0; g < ng; g++) {
        const uint32_t k_exec = sc_ctl_ld(sb, gx + SCG_K + g);
        gu8* const o = out + ((uint64_t)sc_ctl_ld(sb, gx + SCG_P + 2u * g) | ((uint64_t)sc_ctl_ld(sb, gx + SCG_P + 2u * g + 1u) << 32));
        const uint32_t xa = xl_exec + g * 1024u + (lane << 4);
        const uint32_t x0 = lds_ld32(xa), xn = lds_ld32(xa + 4u), xd = lds_ld32(xa + 8u), xo = lds_ld32(xa + 12u);
        uint64_t dm = __ballot(lane < k_exec && (x0 >> 31) != 0u);
        while (dm) {
          const uint32_t k = (uint32_t)__builtin_ctzll(dm);
          dm &= dm - 1ull;
#ifdef BROTLI_AMD_PROFILE_SCAN
          dep_n++;
#endif
          const uint32_t n = rdlane(xn, k), dist = rdlane(xd, k), dpos = rdlane(xo, k) + ((rdlane(x0, k) >> 16) & 63u);
          gu8* const dst = o + dpos; gu8* const src = dst - dist;
          if (dist < n) {
            // the copy overlaps itself (decode.rs:2657-2663, 2690-2720: byte by byte, so a pattern of `dist` bytes repeats)
            if (dist >= 64u) { for (uint32_t c = lane; c < n + lane; c += 64u) if (c < n) dst[c] = src[c]; }  // a step reads what earlier steps wrote
            else {
              uint32_t mm = lane % dist; const uint32_t step = 64u % dist;
              for (uint32_t c = 0; c < n; c += 64u) { if (c + lane < n) dst[c + lane] = src[mm]; mm += step; if (mm >= dist) mm -= dist; }
            }
          } else if (n <= 64u) { uint32_t t = 0; if (lane < n) t = src[lane]; if (lane < n) dst[lane] = (uint8_t)t; }
          else {
            const uint32_t n16 = n >> 4;
            for (uint32_t c = lane; c < n16; c += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16);
            const uint32_t tail = n16 << 4;
            if (tail + lane < n) dst[tail + lane] = src[tail + lane];
          }
        }
      }
#ifdef BROTLI_AMD_PROFILE_SCAN
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (blockIdx.x == 0 && lane == 0) { g_scan_prof[25] += __builtin_amdgcn_s_memtime() - dep_t0; g_scan_prof[26] += dep_n; }
#endif
    }
    if (me == 0) {
      SCAN_PROF(8);
      uint32_t ng = 0, flags = 0, any_dep = 0;
      const uint64_t p_group = P;  // copies that read at or behind this are the ones wave 0 does itself, next tick
      bool stop = mode == M_FINAL, step_done = mode == M_FINAL;
      while (!stop && !step_done && ng < SC_GROUP) {
        // ---- walk: up to 64 entries.  The walk writes an entry's first lane only ("anchor"): the command's position (rC; a
        // run of four commands whose total length D4 knows takes four lanes, the three behind the anchor are found
        // afterwards, by their own lanes), or, for the pieces of a command walked by hand, the piece itself (rA, rB, rC
        // with bit 31 set).
        uint32_t rA = 0, rB = 0, rC = 0, K = 0;
        uint64_t anchors = 0;
        uint32_t man_lane = 64u, man_end = 0;  // lane of the copy of a command walked by hand, and the bit after its distance
#define SC_APPEND(a_, b_, c_) do { const uint32_t a__ = (a_), b__ = (b_), c__ = (c_); \
          asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0" \
                       : "+v"(rA), "+v"(rB), "+v"(rC) : "s"(a__), "s"(b__), "s"(c__), "s"(K) : "m0"); anchors |= 1ull << K; K++; } while (0)
#define SC_ANCHOR(c_, n_) do { const uint32_t c__ = (c_); \
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(rC) : "s"(c__), "s"(K) : "m0"); anchors |= 1ull << K; K += (n_); } while (0)
        for (;;) {
          if (in_run) {
            // literals of a command walked by hand: pieces of up to 32, one lane each; then its distance and copy
            while (run_rem != 0u && K < 64u) {
              if (run_p >= f_32) { step_done = true; break; }
              const uint32_t n = run_rem < 32u ? run_rem : 32u;
              const uint32_t nxt = n == 32u ? run_p + rfl(lds_ld16(sb + SC_J32 + ((run_p & SC_M) << 1))) : rfl(sc_skip(sb, run_p, n));
              SC_APPEND(0u, 0u, 0x80000000u | (n << 24) | (run_p & SC_M));
              run_p = nxt; run_rem -= n;
            }
            if (step_done || K >= 64u) break;
            if (run_p >= f_32) { step_done = true; break; }
            uint32_t kind = SCK_IMPLICIT, val = 0, dbits = 0;
            if (!run_implicit) {
              uint32_t dlo, dhi;
              sc_bits64(sb, run_p, dlo, dhi);
              const uint32_t dtree = run_dctx == 0u ? dt0 : run_dctx == 1u ? dt1 : run_dctx == 2u ? dt2 : dt3;
              const ScDist d = sc_dist(dlo, dhi, dtree, postfix_bits, num_direct);
              kind = rfl(d.kind); val = rfl(d.val); dbits = rfl(d.bits);
            }
            man_lane = K; man_end = run_p + dbits;
            SC_APPEND(run_copy, (kind << 30) | (val & 0x3FFFFFFFu), 0xC0000000u);
            b = man_end; in_run = false;
          }
          // one LDS round trip per hop: four commands where D4 knows the way, two where D2 does (the end of a region,
          // mostly), else one (D2 / D4 and the command's own length are asked for together)
          // (the next hop's two reads go out before this hop is written down: the walk is a chain of LDS round trips, and
          // what lies between two of them is all that can be taken off it; a read at or beyond walk_limit is not used)
          uint32_t d24_ = lds_ld32(sb + SC_D24 + ((b & (SC_N2 - 1u)) << 2)), rec_ = lds_ld32(sb + SC_REC + ((b & (SC_N2 - 1u)) << 3));
          while (K < 64u && b < walk_limit) {
            const uint32_t d24 = rfl(d24_), delta = rfl(rec_) & 0x7FFu;
            // (selects, not branches: the scalar unit does them in a handful of instructions)
            const uint32_t d4_ = K <= 60u ? d24 >> 16 : 0u, d2_ = K <= 62u ? d24 & 0xFFFFu : 0u;
            const uint32_t step_ = d4_ != 0u ? d4_ : d2_ != 0u ? d2_ : delta;
            const uint32_t n_ = d4_ != 0u ? 4u : d2_ != 0u ? 2u : 1u;
            if (step_ == 0u) break;
            const uint32_t b_here = b;
            b += step_;
            d24_ = lds_ld32(sb + SC_D24 + ((b & (SC_N2 - 1u)) << 2)); rec_ = lds_ld32(sb + SC_REC + ((b & (SC_N2 - 1u)) << 3));
            SC_STAGE();
            SC_ANCHOR(b_here, n_);
          }
          if (K >= 64u) break;
          if (b >= walk_limit) { step_done = true; break; }
          // not in REC: a command to walk by hand.  The batch so far goes first, so that the counts below are exact.
          if (K != 0u) break;
          uint32_t lo, hi;
          sc_bits64(sb, b, lo, hi);
          const ScHead h = sc_head(lo, hi, cmd_tree, lut_vgpr);
          const uint32_t hins = rfl(h.insert), hcopy = rfl(h.copy), hbits = rfl(h.bits), himp = rfl(h.implicit), hctx = rfl(h.dctx);
          if (hbits == 0u || hins >= SC_LONG_EXIT || bl1 == 0u || hins > bl0 || (uint64_t)hins + hcopy >= (uint64_t)quota || (!himp && bl2 == 0u)) { stop = true; exit_why = hins >= SC_LONG_EXIT ? 4u : 3u; break; }
          in_run = true; run_p = b + hbits; run_rem = hins; run_copy = hcopy; run_implicit = himp; run_dctx = hctx;
        }
#undef SC_APPEND
#undef SC_ANCHOR
        SCAN_PROF(4);
        // ---- the lanes behind an anchor find their command: one or two hops from the anchor's position, then its REC ----
        {
          const uint64_t am = anchors & (~0ull >> (63u - lane));       // anchors at or below this lane
          const uint32_t al = am ? sc_msb64(am) : 0u, j = lane - al;
          const uint32_t ac = bperm(al << 2, rC);
          if (lane < K && (ac >> 31) == 0u) {
            uint32_t pos = ac;
            if (j & 2u) pos += lds_ld32(sb + SC_D24 + ((pos & (SC_N2 - 1u)) << 2)) & 0xFFFFu;
            if (j & 1u) pos += lds_ld32(sb + SC_REC + ((pos & (SC_N2 - 1u)) << 3)) & 0x7FFu;
            const uint32_t ra = sb + SC_REC + ((pos & (SC_N2 - 1u)) << 3);
            rA = lds_ld32(ra); rB = lds_ld32(ra + 4u); rC = pos;
          }
        }
        // ---- resolve: lane k = entry k of the batch (selects, not branches: every lane does the same) ----
        const bool active = lane < K;
        const bool manual = (rC >> 31) != 0u;
        const uint32_t ins = active ? (manual ? (rC >> 24) & 63u : (rA >> 17) & 63u) : 0u;
        const uint32_t copy = active ? (manual ? rA : (rA >> 23) | ((rB & 15u) << 9)) : 0u;
        const uint32_t kind = active ? (manual ? rB >> 30 : (rB >> 4) & 3u) : (uint32_t)SCK_NONE;
        const uint32_t val = manual ? rB & 0x3FFFFFFFu : rB >> 6;
        const uint32_t litidx = (manual ? rC : rC + ((rA >> 11) & 63u)) & SC_M;
        const uint32_t iscmd = active ? (manual ? (rC >> 30) & 1u : 1u) : 0u;
        const uint32_t isdist = (kind == SCK_EXPLICIT || kind == SCK_SHORT) ? 1u : 0u;
        const uint32_t s1 = sc_scan(ins | (iscmd << 16) | (isdist << 24));
        const uint32_t s2 = sc_scan(ins + copy);
        const uint32_t lit_incl = s1 & 0xFFFFu, cmd_incl = (s1 >> 16) & 0xFFu, dst_incl = s1 >> 24;
        const uint32_t out_excl = s2 - (ins + copy);
        bool ok = lit_incl <= bl0 && cmd_incl <= bl1 && dst_incl <= bl2 && s2 < quota;
        // the distance ring (TakeDistanceFromRingBuffer, decode.rs:2017-2049): short codes read the last four distances
        // that were pushed; a lane whose source is itself a short code waits for it
        const bool need = kind == SCK_SHORT || kind == SCK_IMPLICIT;
        const uint32_t code = kind == SCK_SHORT ? val : 0u;
        const bool pushes = kind == SCK_EXPLICIT || (kind == SCK_SHORT && val != 0u);
        const uint64_t pm = __ballot(pushes);
        int32_t dist = kind == SCK_EXPLICIT ? (int32_t)val : 0;
        // the pushing lanes in batch order: lane r of `perm` is the lane of the r-th push (every lane is sent somewhere:
        // the pushes to their rank, the others behind them, so the scatter is a permutation)
        const uint32_t npush = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));  // pushes in front of this lane
        const uint32_t n_all = (uint32_t)__popcll(pm);
        const uint32_t perm = (uint32_t)__builtin_amdgcn_ds_permute((int)((pushes ? npush : n_all + lane - npush) << 2), (int)lane);
        if (__ballot(need) != 0ull) {
          const uint32_t back = code == 0u ? 0u : 3u - ((0xaaafff1bu >> (code << 1)) & 3u);
          // the (back + 1)-th last push in front of this lane is the source; the ring the batch started with behind them
          const bool from_carry = npush <= back;
          const uint32_t ci = back - npush;  // (meaningful when from_carry)
          const int32_t carry = ci == 0u ? d0 : ci == 1u ? d1 : ci == 2u ? d2 : d3;
          // (read by every lane: inside a branch the permute would only see the lanes that took it; a lane that takes its value
          // from the old ring reads some lane and does not look at what it got)
          const uint32_t src = bperm(((npush - 1u - back) & 63u) << 2, perm);
          uint32_t resolved = need ? 0u : 1u;
          const int32_t mag = (int32_t)((0xfa5fa500u >> (code << 1)) & 3u);
          while (__ballot(resolved == 0u) != 0ull) {
            const int32_t sv = (int32_t)bperm(src << 2, (uint32_t)dist);
            const uint32_t sr = bperm(src << 2, resolved);
            const bool can = resolved == 0u && (from_carry || sr != 0u);
            int32_t v = from_carry ? carry : sv;
            const int32_t vp = v + mag, vm = v - mag;
            v = code == 0u ? v : (code & 1u) ? vp : (vm <= 0 ? 0x7fffffff : vm);
            dist = can ? v : dist; resolved = can ? 1u : resolved;
          }
        }
        {
          // max distance at the copy (decode.rs:2583-2589); beyond it the distance names a dictionary word
          const uint64_t pk = P + out_excl + ins;
          const int32_t maxd = pk < (uint64_t)(uint32_t)max_backward ? (int32_t)pk : max_backward;
          ok = ok && (kind == SCK_NONE || (dist > 0 && dist <= maxd));
        }
        const uint64_t stopmask = __ballot(active && !ok);
        const uint32_t kp = stopmask ? (uint32_t)__builtin_ctzll(stopmask) : K;
        uint32_t extra_cmd = 0, extra_dst = 0;
        if (kp < K) {
          stop = true;
          exit_why = rdlane((lit_incl <= bl0 && cmd_incl <= bl1 && dst_incl <= bl2 && s2 < quota) ? 2u : 1u, kp);
          if (kp == man_lane) {
            // the copy of a command walked by hand: its literals are out, its distance is read -- the checked loop
            // goes on behind the distance (decode.rs:2583, postReadDistance)
            exit_form = SCX_POST_DISTANCE; exit_copy = rdlane(copy, kp); exit_dcode = (int32_t)rdlane((uint32_t)dist, kp);
            const uint32_t kk = rdlane(kind, kp), vv = rdlane(val, kp);
            exit_dctx = (kk == SCK_IMPLICIT || (kk == SCK_SHORT && vv == 0u)) ? 1u : 0u;
            extra_cmd = 1u; extra_dst = rdlane(isdist, kp);  // its command and distance symbols are read
            b = man_end;
          } else {
            exit_form = SCX_BEGIN;
            b = rdlane(rC, kp);  // (never a lane of a literal run: those were checked before the run began)
          }
          in_run = false;
        }
        // totals of what will be executed
        uint32_t lit_tot = 0, cmd_tot = 0, dst_tot = 0, out_tot = 0;
        if (kp != 0u) { const uint32_t t1 = rdlane(s1, kp - 1u); lit_tot = t1 & 0xFFFFu; cmd_tot = (t1 >> 16) & 0xFFu; dst_tot = t1 >> 24; out_tot = rdlane(s2, kp - 1u); }
        cmd_tot += extra_cmd; dst_tot += extra_dst;
        {  // the ring after the executed lanes: their last pushes in front of the old entries
          const uint32_t got = (uint32_t)__popcll(pm & ((kp >= 64u) ? ~0ull : ((1ull << kp) - 1ull)));
          if (got != 0u) {
            const uint32_t dperm = bperm(perm << 2, (uint32_t)dist);  // lane r: the distance of the r-th push
            const int32_t o0 = d0, o1 = d1, o2 = d2;
            d0 = (int32_t)rdlane(dperm, got - 1u);
            d1 = got >= 2u ? (int32_t)rdlane(dperm, got - 2u) : o0;
            d2 = got >= 3u ? (int32_t)rdlane(dperm, got - 3u) : got == 2u ? o0 : o1;
            d3 = got >= 4u ? (int32_t)rdlane(dperm, got - 4u) : got == 3u ? o0 : got == 2u ? o1 : o2;
          }
        }
        if (kp != 0u) {
          // the batch for the executing waves; a copy whose source reaches into the group's own output is wave 0's (bit 31)
          const uint64_t src_end = (P - p_group) + out_excl + ins + copy;
          const uint32_t dep = (copy != 0u && src_end > (uint64_t)(uint32_t)dist) ? 1u : 0u;
          if (__ballot(lane < kp && dep != 0u) != 0ull) any_dep = 1u;
          if (lane < kp) {
            const uint32_t xa = xl_post + ng * 1024u + (lane << 4);
            lds_st32(xa, litidx | (ins << 16) | (dep << 31)); lds_st32(xa + 4u, copy); lds_st32(xa + 8u, (uint32_t)dist); lds_st32(xa + 12u, out_excl);
          }
          sc_ctl_st(sb, gp + SCG_K + ng, kp); sc_ctl_st(sb, gp + SCG_P + 2u * ng, (uint32_t)P); sc_ctl_st(sb, gp + SCG_P + 2u * ng + 1u, (uint32_t)(P >> 32));
          ng++;
        }
        // state after the batch
        P += out_tot; bl0 -= lit_tot; bl1 -= cmd_tot; bl2 -= dst_tot; quota -= out_tot; mlen -= (int32_t)out_tot; ncmd += cmd_tot;
        SCAN_PROF(5);
        SCAN_COUNT(12, kp); SCAN_COUNT(13, 1);
      }
      if (stop) flags |= SCF_LEAVE;
      else if (!step_done) flags |= SCF_BEHIND;  // the group is full: the walk goes on in a tick of its own
      if (mode == M_FINAL) flags = SCF_LEAVE;
      sc_ctl_st(sb, gp + SCG_NG, ng); sc_ctl_st(sb, SCC_FLAGS, flags); sc_ctl_st(sb, gp + SCG_ANYDEP, any_dep);
    }
    if (mode == M_STEP && pre_ok) {
      // the next step's input goes into its ring slots (nothing reads them before the next step)
      const uint32_t slot = pre_i & (SC_IN_DW - 1u);
      lds_st32(sb + SC_IN + (slot << 2), pre_v);
      if (slot < 2u) lds_st32(sb + SC_IN + ((SC_IN_DW + slot) << 2), pre_v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ---- REC of the step is complete; the group is posted ----
    SCAN_PROF(3);
    if (mode == M_STEP) f_rec += SC_N;
    // ---- what comes next (every wave decides the same from the posted flags) ----
    if (mode == M_FINAL) break;
    const uint32_t flags = sc_ctl_ld(sb, SCC_FLAGS);
    const bool can_step = f_rec + SC_N + SC_AHEAD <= in_limit;
    if (flags & SCF_LEAVE) mode = M_FINAL;
    else if (flags & SCF_BEHIND) mode = M_SYNC;
    else if (walk_limit < f_rec) mode = can_step ? (uint32_t)M_STEP : (uint32_t)M_SYNC;  // a complete region that has not been walked yet
    else mode = can_step ? (uint32_t)M_STEP : (uint32_t)M_FINAL;
  }
  __syncthreads();  // every store of the engine is in memory before the decoding wave goes on alone
  if (me != 0) return 0;
#ifdef BROTLI_AMD_PROFILE_SCAN
  if (blockIdx.x == 0 && lane == 0) { for (int k = 0; k < 16; k++) g_scan_prof[k] += sp_acc[k]; g_scan_prof[16] += ncmd; g_scan_prof[17] += 1; g_scan_prof[18 + (exit_why < 5u ? exit_why : 0u)] += 1; if (ncmd < 64u) g_scan_prof[23] += 1; }
#endif
  // ---- hand the stream back (LDS_LEAN, as lean_commands does) ----
  uint32_t pos = b, lits_left = 0, insert_len = 0, copy_len = exit_copy; int32_t dcode = exit_dcode; uint32_t dctx = exit_dctx;
  if (in_run) {  // inside a command walked by hand: the checked loop finishes its literals, distance and copy
    exit_form = SCX_LITERALS_REST; pos = run_p; lits_left = run_rem; insert_len = run_rem; copy_len = run_copy;
    dcode = run_implicit ? 0 : -1; dctx = run_dctx;
    mlen -= (int32_t)run_rem;  // (the reference takes a command's whole insert length off when it reads the command)
    bl1 -= 1u; ncmd += 1u;
  }
  if (lane == 0) {
    LEAN_ST(L_SC_POS_LO, pos); LEAN_ST(L_SC_POS_HI, exit_form);
    LEAN_ST(L_P_LO, (uint32_t)P); LEAN_ST(L_P_HI, (uint32_t)(P >> 32)); LEAN_ST(L_QUOTA, quota); LEAN_ST(L_MLEN, mlen);
    LEAN_ST(L_BL0, bl0); LEAN_ST(L_BL1, bl1); LEAN_ST(L_BL2, bl2);
    LEAN_ST(L_D0, d0); LEAN_ST(L_D1, d1); LEAN_ST(L_D2, d2); LEAN_ST(L_D3, d3); LEAN_ST(L_NCMD_LO, ncmd);
    LEAN_ST(L_INSERT, insert_len); LEAN_ST(L_COPY, copy_len); LEAN_ST(L_DCODE, dcode); LEAN_ST(L_DCTX, dctx); LEAN_ST(L_LITS_LEFT, lits_left);
  }
  lds_sync();
  return ncmd;
}
