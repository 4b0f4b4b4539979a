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
//   pass 2:
//     walk      wave 0 follows REC from the stream's real position: one LDS round trip per command, 64 commands a batch
//     resolve   wave 0, lane = command: output offsets (prefix sums), block counts, the distance ring
//               (decode.rs:2017-2049) and every limit the reference checks; the first command that needs anything
//               unusual (dictionary word, overlapping copy, block switch, end of the metablock / output / ring segment)
//               ends the engine's part in front of it and the checked command loop takes over for that command
//     execute   all waves: literals out of S through J*, then the LZ77 copies (decode.rs:2641-2680); copies whose source
//               lies in the batch's own output are done last, in order, by wave 0
//
// Everything lives in LDS rings indexed by stream bit position (mod SC_R); what a phase reads is always behind the
// frontier of the phase that produces it (the lags below).  Results never depend on the speculation: a REC entry is
// either the exact parse of the command that starts there or zero ("walk it by hand").
#pragma once

constexpr uint32_t SC_WAVES = 16;                 // waves of a block that runs the engine
constexpr uint32_t SC_R = 4096;                   // ring size in stream bits
constexpr uint32_t SC_M = SC_R - 1;
constexpr uint32_t SC_N = 2048;                   // bits a step advances by
constexpr uint32_t SC_IN_DW = SC_R / 32;          // input ring in dwords (+ 2 mirrored at the end)
// frontier lags (bits): J(2n)[b] reads Jn at up to b + 15 n; REC[b] reads J* up to b + 63 + 63 * 15 and the input 96 bits on
constexpr uint32_t SC_LAG_REC = 1088, SC_LAG_32 = 256, SC_LAG_16 = 128, SC_LAG_8 = 64, SC_LAG_4 = 64, SC_LAG_2 = 64, SC_LAG_IN = 64;
constexpr uint32_t SC_AHEAD = SC_LAG_REC + SC_LAG_32 + SC_LAG_16 + SC_LAG_8 + SC_LAG_4 + SC_LAG_2 + SC_LAG_IN;  // input frontier - REC frontier
static_assert(SC_N + SC_AHEAD <= SC_R, "ring too small for one step plus the lags");
constexpr uint32_t SC_LONG_EXIT = 8192;           // literal runs from here on go back to the rounds of the helper waves
constexpr uint32_t SC_MIN_QUOTA = 2048;           // output bytes that must be possible for the engine to start

// LDS layout, offsets from the engine's base
constexpr uint32_t SC_CTL = 0;                    // 128: control words
constexpr uint32_t SC_IN = 128;                   // input ring: SC_IN_DW + 2 dwords
constexpr uint32_t SC_S = SC_IN + (SC_IN_DW + 2) * 4 + 8;   // literal at every bit
constexpr uint32_t SC_J1 = SC_S + SC_R;           // code length at every bit
constexpr uint32_t SC_J2 = SC_J1 + SC_R;
constexpr uint32_t SC_J4 = SC_J2 + SC_R;
constexpr uint32_t SC_J8 = SC_J4 + SC_R;
constexpr uint32_t SC_J16 = SC_J8 + SC_R;
constexpr uint32_t SC_J32 = SC_J16 + SC_R;        // u16
constexpr uint32_t SC_REC = SC_J32 + 2 * SC_R;    // 8 bytes per bit of the current step
constexpr uint32_t SC_XL = SC_REC + 8 * SC_N;     // 64 x 16: the batch being executed
constexpr uint32_t SC_BYTES = SC_XL + 64 * 16;
static_assert(SC_S % 16 == 0 && SC_REC % 16 == 0 && SC_XL % 16 == 0, "alignment");

enum { SCC_CMD = 0, SCC_K = 1, SCC_P_LO = 2, SCC_P_HI = 3, SCC_BASE_DW = 4, SCC_IN_LIMIT = 5, SCC_LIT_TREE = 6, SCC_CMD_TREE = 7,
       SCC_DT0 = 8, SCC_POSTFIX = 12, SCC_NUM_DIRECT = 13, SCC_OUT_LO = 14, SCC_OUT_HI = 15, SCC_ENTRY = 16 };
enum { SCC_CMD_EXEC = 1, SCC_CMD_STEP = 2, SCC_CMD_EXIT = 3 };
enum { SCK_NONE = 0, SCK_EXPLICIT = 1, SCK_SHORT = 2, SCK_IMPLICIT = 3 };

__device__ __forceinline__ uint32_t sc_ctl_ld(uint32_t sb, uint32_t k) { return rfl(*reinterpret_cast<lds_vu32*>(&g_smem[sb + SC_CTL + 4u * k])); }
__device__ __forceinline__ void sc_ctl_st(uint32_t sb, uint32_t k, uint32_t v) { if (lane_id() == 0) *reinterpret_cast<lds_vu32*>(&g_smem[sb + SC_CTL + 4u * k]) = v; }

#ifdef BROTLI_AMD_PROFILE_SCAN
__device__ unsigned long long g_scan_prof[24];
#define SCAN_PROF(k) do { if (me == 0) { uint64_t _t = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0) sp_acc[k] += _t - sp_t; sp_t = _t; } } while (0)
#define SCAN_COUNT(k, v) do { if (me == 0 && blockIdx.x == 0) sp_acc[k] += (v); } while (0)
#else
#define SCAN_PROF(k) do { } while (0)
#define SCAN_COUNT(k, v) do { } while (0)
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
__device__ __forceinline__ ScDist sc_dist(uint32_t lo, uint32_t hi, uint32_t dtree_addr, uint32_t postfix_bits, uint32_t num_direct) {
  uint32_t code, L;
  sc_lookup(dtree_addr, lo, code, L);
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
  return d;
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
  }
  // a literal run being walked by hand (commands REC does not hold): literals still to skip, then the distance and the copy
  bool in_run = false; uint32_t run_p = 0, run_rem = 0, run_copy = 0, run_implicit = 0, run_dctx = 0;
  uint32_t exit_form = SCX_BEGIN, exit_copy = 0, exit_dctx = 0; int32_t exit_dcode = 0;

  // frontiers (bits from the origin): below them the ring holds valid entries
  uint32_t f_in = 0, f_1 = 0, f_2 = 0, f_4 = 0, f_8 = 0, f_16 = 0, f_32 = 0, f_rec = 0;

  for (;;) {
    // ================= pass 1: one step =================
    const uint32_t e_rec = f_rec + SC_N;
    const uint32_t t_32 = e_rec + SC_LAG_REC, t_16 = t_32 + SC_LAG_32, t_8 = t_16 + SC_LAG_16, t_4 = t_8 + SC_LAG_8, t_2 = t_4 + SC_LAG_4,
                   t_1 = t_2 + SC_LAG_2, t_in = t_1 + SC_LAG_IN;
    if (t_in > in_limit) {  // the input ends before another step's worth: every wave sees that, wave 0 hands the stream back where it is
      break;
    }
    // input: dwords [f_in / 32, t_in / 32) into the ring (first two ring slots mirrored behind its end)
    for (uint32_t i = (f_in >> 5) + threadIdx.x; i < (t_in >> 5); i += 64u * SC_WAVES) {
      const uint32_t v = in_dw[i];
      const uint32_t slot = i & (SC_IN_DW - 1u);
      lds_st32(sb + SC_IN + (slot << 2), v);
      if (slot < 2u) lds_st32(sb + SC_IN + ((SC_IN_DW + slot) << 2), v);
    }
    f_in = t_in;
    __syncthreads();
    SCAN_PROF(0);
    // S / J1
    for (uint32_t w = (f_1 >> 6) + me; w < (t_1 >> 6); w += SC_WAVES) {
      const uint32_t p = (w << 6) + lane;
      uint32_t sym, L;
      sc_lookup(lit_tree, sc_bits32(sb, p), sym, L);
      lds_st8(sb + SC_S + (p & SC_M), sym);
      lds_st8(sb + SC_J1 + (p & SC_M), L);
    }
    f_1 = t_1;
    __syncthreads();
    SCAN_PROF(1);
#define SC_LEVEL(FROM, TO, f_to, t_to) \
    for (uint32_t w = ((f_to) >> 6) + me; w < ((t_to) >> 6); w += SC_WAVES) { \
      const uint32_t p = (w << 6) + lane; \
      const uint32_t a = lds_ld8(sb + FROM + (p & SC_M)); \
      const uint32_t c = lds_ld8(sb + FROM + ((p + a) & SC_M)); \
      lds_st8(sb + TO + (p & SC_M), a + c); \
    } \
    f_to = (t_to); \
    __syncthreads();
    SC_LEVEL(SC_J1, SC_J2, f_2, t_2)
    SC_LEVEL(SC_J2, SC_J4, f_4, t_4)
    SC_LEVEL(SC_J4, SC_J8, f_8, t_8)
    SC_LEVEL(SC_J8, SC_J16, f_16, t_16)
#undef SC_LEVEL
    for (uint32_t w = (f_32 >> 6) + me; w < (t_32 >> 6); w += SC_WAVES) {
      const uint32_t p = (w << 6) + lane;
      const uint32_t a = lds_ld8(sb + SC_J16 + (p & SC_M));
      const uint32_t c = lds_ld8(sb + SC_J16 + ((p + a) & SC_M));
      lds_st16(sb + SC_J32 + ((p & SC_M) << 1), a + c);
    }
    f_32 = t_32;
    __syncthreads();
    SCAN_PROF(2);
    // REC: the command that would start at every bit of [f_rec, e_rec)
    for (uint32_t w = (f_rec >> 6) + me; w < (e_rec >> 6); w += SC_WAVES) {
      const uint32_t p = (w << 6) + lane;
      uint32_t lo, hi;
      sc_bits64(sb, p, lo, hi);
      const ScHead h = sc_head(lo, hi, cmd_tree, lut_vgpr);
      bool ok = h.insert < 64u && h.copy < 8192u;
      uint32_t q = sc_skip(sb, p + h.bits, h.insert & 63u);
      uint32_t kind = SCK_IMPLICIT, val = 0;
      if (!h.implicit) {
        uint32_t dlo, dhi;
        sc_bits64(sb, q, dlo, dhi);
        const uint32_t dtree = h.dctx == 0u ? dt0 : h.dctx == 1u ? dt1 : h.dctx == 2u ? dt2 : dt3;
        const ScDist d = sc_dist(dlo, dhi, dtree, postfix_bits, num_direct);
        kind = d.kind; val = d.val; q += d.bits;
        ok = ok && val < (1u << 26);
      }
      const uint32_t delta = q - p;
      ok = ok && delta != 0u && delta < 2048u;
      const uint32_t rlo = delta | (h.bits << 11) | (h.insert << 17) | ((h.copy & 0x1FFu) << 23);
      const uint32_t rhi = (h.copy >> 9) | (kind << 4) | (val << 6);
      const uint32_t ra = sb + SC_REC + ((p & (SC_N - 1u)) << 3);
      lds_st32(ra, ok ? rlo : 0u);
      lds_st32(ra + 4u, ok ? rhi : 0u);
    }
    f_rec = e_rec;
    __syncthreads();
    SCAN_PROF(3);

    // ================= pass 2: batches of up to 64 commands =================
    // Per batch: wave 0 walks and resolves, then posts the batch (SCC_K entries in SC_XL, output position, and what
    // comes after it: another batch, the next step, or the end of the engine's part); barrier; every wave executes.
    bool leave = false;
    for (;;) {
      uint32_t v_ins = 0, v_copy = 0, v_dist = 0, v_off = 0, v_dep = 0, kp = 0;  // (wave 0) the batch, lane = entry
      if (me == 0) {
        uint32_t rA = 0, rB = 0, rC = 0, K = 0;
        bool stop = false, step_done = false;
        uint32_t man_lane = 64u, man_end = 0;  // lane of the copy of a command walked by hand, and the bit after its distance
#define SC_APPEND(a_, b_, c_) do { const uint32_t a__ = (a_), b__ = (b_), c__ = (c_); \
          asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0" \
                       : "+v"(rA), "+v"(rB), "+v"(rC) : "s"(a__), "s"(b__), "s"(c__), "s"(K) : "m0"); K++; } while (0)
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
          if (K >= 64u) break;
          if (b >= f_rec) { step_done = true; break; }
          const uint32_t ra = sb + SC_REC + ((b & (SC_N - 1u)) << 3);
          const uint32_t rlo = rfl(lds_ld32(ra)), rhi = rfl(lds_ld32(ra + 4u));
          const uint32_t delta = rlo & 0x7FFu;
          if (delta != 0u) { SC_APPEND(rlo, rhi, b); b += delta; continue; }
          // not in REC: a command to walk by hand.  The batch so far goes first, so that the counts below are exact.
          if (K != 0u) break;
          uint32_t lo, hi;
          sc_bits64(sb, b, lo, hi);
          const ScHead h = sc_head(lo, hi, cmd_tree, lut_vgpr);
          const uint32_t hins = rfl(h.insert), hcopy = rfl(h.copy), hbits = rfl(h.bits), himp = rfl(h.implicit), hctx = rfl(h.dctx);
          if (hbits == 0u || hins >= SC_LONG_EXIT || bl1 == 0u || hins > bl0 || (uint64_t)hins + hcopy >= (uint64_t)quota || (!himp && bl2 == 0u)) { stop = true; break; }
          in_run = true; run_p = b + hbits; run_rem = hins; run_copy = hcopy; run_implicit = himp; run_dctx = hctx;
        }
#undef SC_APPEND
        SCAN_PROF(4);
        // ---- resolve: lane k = entry k of the batch ----
        const bool active = lane < K;
        const bool manual = (rC >> 31) != 0u;
        const uint32_t ins = !active ? 0u : manual ? (rC >> 24) & 63u : (rA >> 17) & 63u;
        const uint32_t copy = !active ? 0u : manual ? rA : (rA >> 23) | ((rB & 15u) << 9);
        const uint32_t kind = !active ? (uint32_t)SCK_NONE : manual ? rB >> 30 : (rB >> 4) & 3u;
        const uint32_t val = manual ? rB & 0x3FFFFFFFu : rB >> 6;
        const uint32_t litidx = manual ? rC & SC_M : (rC + ((rA >> 11) & 63u)) & SC_M;
        const uint32_t iscmd = !active ? 0u : manual ? (rC >> 30) & 1u : 1u;
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
        uint64_t m = pm & ((1ull << lane) - 1ull);
        const uint32_t back = code == 0u ? 0u : 3u - ((0xaaafff1bu >> (code << 1)) & 3u);
        const uint32_t npush = (uint32_t)__popcll(m);
        const bool from_carry = npush <= back;
        const uint32_t ci = back - npush;  // (meaningful when from_carry)
        const int32_t carry = ci == 0u ? d0 : ci == 1u ? d1 : ci == 2u ? d2 : d3;
        if (!from_carry) for (uint32_t t = 0; t < back; t++) m &= ~(1ull << sc_msb64(m));
        const uint32_t src = from_carry ? 0u : sc_msb64(m);
        int32_t dist = kind == SCK_EXPLICIT ? (int32_t)val : 0;
        uint32_t resolved = need ? 0u : 1u;
        while (__ballot(resolved == 0u) != 0ull) {
          const int32_t sv = (int32_t)bperm(src << 2, (uint32_t)dist);
          const uint32_t sr = bperm(src << 2, resolved);
          if (resolved == 0u && (from_carry || sr != 0u)) {
            int32_t v = from_carry ? carry : sv;
            if (code != 0u) {
              const int32_t mag = (int32_t)((0xfa5fa500u >> (code << 1)) & 3u);
              if (code & 1u) v += mag; else { v -= mag; if (v <= 0) v = 0x7fffffff; }
            }
            dist = v; resolved = 1u;
          }
        }
        if (kind != SCK_NONE) {
          // max distance at the copy (decode.rs:2583-2589); beyond it the distance names a dictionary word
          const uint64_t pk = P + out_excl + ins;
          const int32_t maxd = pk < (uint64_t)(uint32_t)max_backward ? (int32_t)pk : max_backward;
          ok = ok && dist > 0 && dist <= maxd && (uint32_t)dist >= copy;
        }
        const uint64_t stopmask = __ballot(active && !ok);
        kp = stopmask ? (uint32_t)__builtin_ctzll(stopmask) : K;
        uint32_t extra_cmd = 0, extra_dst = 0;
        if (kp < K) {
          stop = true;
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
          uint64_t pk = pm & ((kp >= 64u) ? ~0ull : ((1ull << kp) - 1ull));
          uint32_t got = 0;
          int32_t nv[4] = {0, 0, 0, 0};
          for (uint32_t t = 0; t < 4u; t++) {
            if (pk == 0ull) break;
            const uint32_t l = sc_msb64(pk);
            nv[t] = (int32_t)rdlane((uint32_t)dist, l);
            pk &= ~(1ull << l); got++;
          }
          const int32_t o0 = d0, o1 = d1, o2 = d2;
          if (got == 1u) { d0 = nv[0]; d1 = o0; d2 = o1; d3 = o2; }
          else if (got == 2u) { d0 = nv[0]; d1 = nv[1]; d2 = o0; d3 = o1; }
          else if (got == 3u) { d0 = nv[0]; d1 = nv[1]; d2 = nv[2]; d3 = o0; }
          else if (got == 4u) { d0 = nv[0]; d1 = nv[1]; d2 = nv[2]; d3 = nv[3]; }
        }
        // the batch for the executing waves
        const uint32_t dep = (copy != 0u && out_excl + ins + copy > (uint32_t)dist) ? 1u : 0u;  // source reaches into this batch's output
        if (lane < kp) {
          const uint32_t xa = sb + SC_XL + (lane << 4);
          lds_st32(xa, litidx | (ins << 16) | (dep << 31)); lds_st32(xa + 4u, copy); lds_st32(xa + 8u, (uint32_t)dist); lds_st32(xa + 12u, out_excl);
        }
        v_ins = ins; v_copy = copy; v_dist = (uint32_t)dist; v_off = out_excl; v_dep = dep;
        sc_ctl_st(sb, SCC_K, kp); sc_ctl_st(sb, SCC_P_LO, (uint32_t)P); sc_ctl_st(sb, SCC_P_HI, (uint32_t)(P >> 32));
        sc_ctl_st(sb, SCC_CMD, stop ? SCC_CMD_EXIT : step_done ? SCC_CMD_STEP : SCC_CMD_EXEC);
        // state after the batch
        P += out_tot; bl0 -= lit_tot; bl1 -= cmd_tot; bl2 -= dst_tot; quota -= out_tot; mlen -= (int32_t)out_tot; ncmd += cmd_tot;
        SCAN_PROF(5);
        SCAN_COUNT(12, kp); SCAN_COUNT(13, 1);
      }
      __syncthreads();  // ---- the batch is posted ----
      const uint32_t k_exec = sc_ctl_ld(sb, SCC_K), next = sc_ctl_ld(sb, SCC_CMD);
      if (k_exec != 0u) {
        gu8* const o = out + ((uint64_t)sc_ctl_ld(sb, SCC_P_LO) | ((uint64_t)sc_ctl_ld(sb, SCC_P_HI) << 32));
        // literals: wave w takes entries w, w + 16, ...; lane i the i-th literal of the entry
        for (uint32_t k = me; k < k_exec; k += SC_WAVES) {
          const uint32_t xa = sb + SC_XL + (k << 4);
          const uint32_t x0 = rfl(lds_ld32(xa)), off = rfl(lds_ld32(xa + 12u));
          const uint32_t n = (x0 >> 16) & 63u;
          if (lane < n) {
            const uint32_t q = sc_skip(sb, x0 & SC_M, lane);
            o[off + lane] = (uint8_t)lds_ld8(sb + SC_S + (q & SC_M));
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // copies whose source lies in front of the batch: the loads of a wave's entries first, then the stores
        {
          uint32_t hold[4] = {0, 0, 0, 0};
          _Pragma("unroll") for (uint32_t t = 0; t < 4u; t++) {
            const uint32_t k = me + t * SC_WAVES;
            if (k >= k_exec) continue;
            const uint32_t xa = sb + SC_XL + (k << 4);
            const uint32_t x0 = rfl(lds_ld32(xa)), n = rfl(lds_ld32(xa + 4u)), dist = rfl(lds_ld32(xa + 8u)), off = rfl(lds_ld32(xa + 12u));
            if (n == 0u || (x0 >> 31) != 0u) continue;
            gu8* const dst = o + off + ((x0 >> 16) & 63u); gu8* const src = dst - dist;
            if (n <= 64u) { if (lane < n) hold[t] = src[lane]; }
            else {
              const uint32_t n16 = n >> 4;
              for (uint32_t c = lane; c < n16; c += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16);
              const uint32_t tail = n16 << 4;
              if (tail + lane < n) dst[tail + lane] = src[tail + lane];
            }
          }
          _Pragma("unroll") for (uint32_t t = 0; t < 4u; t++) {
            const uint32_t k = me + t * SC_WAVES;
            if (k >= k_exec) continue;
            const uint32_t xa = sb + SC_XL + (k << 4);
            const uint32_t x0 = rfl(lds_ld32(xa)), n = rfl(lds_ld32(xa + 4u)), off = rfl(lds_ld32(xa + 12u));
            if (n == 0u || (x0 >> 31) != 0u || n > 64u) continue;
            if (lane < n) o[off + ((x0 >> 16) & 63u) + lane] = (uint8_t)hold[t];
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (me == 0) {
          // copies that read what this batch wrote: one after the other (a wave's stores are visible to its later loads)
          uint64_t dm = __ballot(lane < kp && v_dep != 0u);
          while (dm) {
            const uint32_t k = (uint32_t)__builtin_ctzll(dm);
            dm &= dm - 1ull;
            const uint32_t n = rdlane(v_copy, k), dist = rdlane(v_dist, k), dpos = rdlane(v_off, k) + rdlane(v_ins, k);
            gu8* const dst = o + dpos; gu8* const src = dst - dist;
            if (n <= 64u) { uint32_t t = 0; if (lane < n) t = src[lane]; if (lane < n) dst[lane] = (uint8_t)t; }
            else {
              const uint32_t n16 = n >> 4;
              for (uint32_t c = lane; c < n16; c += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)c * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)c * 16);
              const uint32_t tail = n16 << 4;
              if (tail + lane < n) dst[tail + lane] = src[tail + lane];
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          SCAN_PROF(6);
        }
      }
      if (next == SCC_CMD_EXEC) continue;  // another batch of this step
      leave = next == SCC_CMD_EXIT;
      break;
    }
    if (leave) break;
  }
  __syncthreads();  // every store of the engine is in memory before the decoding wave goes on alone
  if (me != 0) return 0;
#ifdef BROTLI_AMD_PROFILE_SCAN
  if (blockIdx.x == 0 && lane == 0) { for (int k = 0; k < 16; k++) g_scan_prof[k] += sp_acc[k]; g_scan_prof[16] += ncmd; g_scan_prof[17] += 1; }
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
