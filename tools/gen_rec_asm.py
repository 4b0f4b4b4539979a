#!/usr/bin/env python3
"""Writes rust-brotli-decompressor_amd/csrc/brotli_rec_run_asm.h: LEAN_REC_RUN_ASM, the hand-written loop over the plain commands of a
context-modelled metablock (lean_rec_commands in brotli_kernels.hip; reference: src/decode.rs:2359-2726 for the path it takes --
ReadCommand's result out of wave 2's records, the literals of decode.rs:2463-2551, ReadDistance of decode.rs:2066-2131 with the ring of
decode.rs:2017-2049, the copy of decode.rs:2641-2680).

Why a generator: the loop is written for the cost of a LONE wave on gfx950 (tools/ubench/branch.hip, s_memtime ticks): a scalar
instruction 5 - 7, a conditional branch that is not taken 13, one that is taken 26 - 29, a v_readlane whose result the next instruction
needs 25, an LDS round trip (v_mov, ds_read, s_waitcnt, v_readfirstlane) 70.  So: the common way through a command falls through every
branch -- everything else (ring codes, implicit distances, second-level table entries, refills of the bit buffer, telling wave 2 where the
reader is) lies OUT OF LINE behind a branch taken only then and branches back --, checks are folded into one sign test where that is
cheaper than a branch each, and the blocks that appear several times (a symbol of a two-level table, the plain-copy test) are instantiated
with labels of their own.  Local labels are numbers handed out here.

Registers inside the asm (all clobbered): s80 / s81 p2 / p1 as they were (a command with literals that is not plain after all puts them
back), s82 RB = (bits of the window's first dword) - (bits of the records' origin), s83 LIMM1 = the last window bit a command may begin at,
s84 POS = the reader's position in window bits, s85 .. s89 and s95 .. s97 scratch, s90 / s91 the record, s92 copy length, s93 distance,
s94 the distance block's count behind this command, s[98:99] the bit buffer of a command with literals; v116 .. v119 scratch, v124 the copy
in flight.  %[cnt] and %[ndw] are the buffer's count and the next WINDOW dword inside a command with literals; they and %[buf] are made
from POS on the way out, whatever happened."""
import os
import sys

PROFILE_WAIT = "--profile-wait" in sys.argv   # s_memtime around every wait for the memory pipe, summed into %[wacc] (a 64-bit "+s" operand; -DBROTLI_AMD_PROFILE_RUN_WAIT)
VMWAIT = """
s_memtime s[76:77]
s_waitcnt lgkmcnt(0)
s_waitcnt vmcnt(0)
s_memtime s[74:75]
s_waitcnt lgkmcnt(0)
s_sub_u32 s74, s74, s76
s_subb_u32 s75, s75, s77
s_add_u32 s78, s78, s74
s_addc_u32 s79, s79, s75""" if PROFILE_WAIT else "s_waitcnt vmcnt(0)"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_label = [200]


def L():
    _label[0] += 1
    return str(_label[0])


class Asm:
    def __init__(self):
        self.main, self.ool = [], []   # the straight way; the blocks out of line (emitted behind the loop)

    def m(self, text, comment=None):
        for line in text.strip().split("\n"):
            self.main.append((line.strip(), comment)); comment = None

    def o(self, text, comment=None):
        for line in text.strip().split("\n"):
            self.ool.append((line.strip(), comment)); comment = None


def out_of_line(A, emit):
    """what `emit` writes -- its straight way, then what lies out of line of IT -- behind the blocks out of line so far (numeric labels are looked for forward)"""
    main_keep, ool_keep = A.main, A.ool
    A.main, A.ool = [], []
    emit()
    ool_keep.extend(A.main); ool_keep.extend(A.ool)
    A.main, A.ool = main_keep, ool_keep


def plain_test(p, q):
    """s95 < 0 unless: 0 < distance s93 <= min(P, max_backward), copy length s92 <= min(63, distance, quota - 1)   (p: the output position, q: the quota, in front of the copy)"""
    return f"""
s_min_u32 s95, {p}, %[maxb]
s_sub_u32 s95, s95, s93
s_sub_u32 s96, s93, 1
s_or_b32 s95, s95, s96
s_min_u32 s96, s93, 63
s_sub_u32 s97, {q}, 1
s_min_u32 s96, s96, s97
s_sub_u32 s96, s96, s92
s_or_b32 s95, s95, s96"""


def ring(code):
    """a ring code 1 .. 15 -> s93 (TakeDistanceFromRingBuffer, decode.rs:2017-2049); scratch s95 .. s97"""
    a, b = L(), L()
    return f"""
s_lshl_b32 s95, {code}, 1
s_lshr_b32 s96, 0xaaafff1b, s95
s_and_b32 s96, s96, 3
s_mov_b32 s93, %[d3]
s_cmp_eq_u32 s96, 1
s_cselect_b32 s93, %[d2], s93
s_cmp_eq_u32 s96, 2
s_cselect_b32 s93, %[d1], s93
s_cmp_eq_u32 s96, 3
s_cselect_b32 s93, %[d0], s93
s_lshr_b32 s97, 0xfa5fa500, s95
s_and_b32 s97, s97, 3
s_bitcmp1_b32 {code}, 0
s_cbranch_scc1 {a}f
s_sub_i32 s93, s93, s97
s_cmp_gt_i32 s93, 0
s_cselect_b32 s93, s93, 0x7fffffff
s_branch {b}f
{a}:
s_add_i32 s93, s93, s97
{b}:"""


def word_test(p, q, fail, src_lo, src_hi, within):
    """The command is not a plain copy.  A word of the static dictionary as it stands (decode.rs:2593-2640 with transform 0: every word of the
    reference's alice29 is one)?  distance s93 beyond min(P, max_backward), 4 <= length s92 <= 24, (distance - max_distance - 1) >> bits[length] == 0,
    length < quota, the distance block's count s94 there -> its address in {src_lo}:{src_hi}; a distance within the window: {within} (see overlap_test); anything else: {fail}.  (p, q as plain_test; lane n of
    %[wtab]: kDictOffsetsByLength[n] | kDictSizeBitsByLength[n] << 24; lanes 6 / 7 of the parameters: the dictionary's address)"""
    return f"""
s_min_u32 s95, {p}, %[maxb]
s_cmp_le_u32 s93, s95
s_cbranch_scc1 {within}f
s_cmp_gt_u32 s93, 0x7ffffffc
s_cbranch_scc1 {fail}f
s_sub_u32 s96, s92, 4
s_cmp_gt_u32 s96, 20
s_cbranch_scc1 {fail}f
s_cmp_ge_u32 s92, {q}
s_cbranch_scc1 {fail}f
s_cmp_lt_i32 s94, 0
s_cbranch_scc1 {fail}f
v_readlane_b32 s96, %[wtab], s92
s_sub_u32 s95, s93, s95
s_sub_u32 s95, s95, 1
s_lshr_b32 s97, s96, 24
s_lshr_b32 s97, s95, s97
s_cmp_lg_u32 s97, 0
s_cbranch_scc1 {fail}f
s_and_b32 s96, s96, 0xffffff
s_mul_i32 s95, s95, s92
s_add_u32 s96, s96, s95
v_readlane_b32 {src_lo}, %[params], 6
v_readlane_b32 {src_hi}, %[params], 7
s_add_u32 {src_lo}, {src_lo}, s96
s_addc_u32 {src_hi}, {src_hi}, 0"""


def overlap_test(q, fail):
    """... or a copy that repeats itself (decode.rs:2641-2680 copies byte by byte: byte i of it is byte i mod distance of its source): 0 < distance s93 < length
    s92 <= 63, length < quota, the distance block's count s94 there -> v115 = lane mod distance, what the copy's load takes for its lanes' offsets
    (lane d of %[mtab]: 65536 / d + 1 -- lane * that >> 16 is lane / d for lanes and distances below 64); anything else: {fail}"""
    return f"""
s_cmp_eq_u32 s93, 0
s_cbranch_scc1 {fail}f
s_cmp_gt_u32 s92, 63
s_cbranch_scc1 {fail}f
s_cmp_ge_u32 s92, {q}
s_cbranch_scc1 {fail}f
s_cmp_lt_i32 s94, 0
s_cbranch_scc1 {fail}f
s_cmp_ge_u32 s93, s92
s_cbranch_scc1 {fail}f
v_readlane_b32 s96, %[mtab], s93
v_mul_u32_u24 v115, %[lane], s96
v_lshrrev_b32 v115, 16, v115
v_mul_u32_u24 v115, v115, s93
v_sub_u32 v115, %[lane], v115"""


PUSH = """
s_mov_b32 %[d3], %[d2]
s_mov_b32 %[d2], %[d1]
s_mov_b32 %[d1], %[d0]
s_mov_b32 %[d0], s93"""


def symbol(A, need):
    """the symbol at the buffer's low end by the two-level table at LDS address s87 -> s86, its bits taken (read_symbol<true>); refill and
    second level out of line.  `need`: bits the buffer must hold (15: a code word)"""
    refill, back1, second, back2 = L(), L(), L(), L()
    A.m(f"""
s_cmp_lt_u32 %[cnt], {need}
s_cbranch_scc1 {refill}f
{back1}:
s_and_b32 s86, s98, 0xff
s_lshl_b32 s86, s86, 1
s_add_u32 s86, s86, s87
v_mov_b32 v117, s86
ds_read_u16 v118, v117
s_waitcnt lgkmcnt(0)
v_readfirstlane_b32 s86, v118
s_and_b32 s88, s86, 15
s_lshr_b32 s86, s86, 4
s_cmp_gt_u32 s88, 8
s_cbranch_scc1 {second}f
{back2}:
s_lshr_b64 s[98:99], s[98:99], s88
s_sub_u32 %[cnt], %[cnt], s88""")
    A.o(f"""
{refill}:
s_cmp_gt_u32 %[ndw], 63
s_cbranch_scc1 190f
v_readlane_b32 s96, %[cur], %[ndw]
s_mov_b32 s97, 0
s_add_u32 %[ndw], %[ndw], 1
s_lshl_b64 s[96:97], s[96:97], %[cnt]
s_add_u32 %[cnt], %[cnt], 32
s_or_b64 s[98:99], s[98:99], s[96:97]
s_branch {back1}b
{second}:
s_sub_u32 s88, s88, 8
s_lshr_b32 s96, s98, 8
s_bfm_b32 s97, s88, 0
s_and_b32 s96, s96, s97
s_add_u32 s86, s86, s96
s_lshl_b32 s86, s86, 1
s_add_u32 s86, s86, s87
v_mov_b32 v117, s86
ds_read_u16 v118, v117
s_waitcnt lgkmcnt(0)
v_readfirstlane_b32 s86, v118
s_and_b32 s88, s86, 15
s_add_u32 s88, s88, 8
s_lshr_b32 s86, s86, 4
s_branch {back2}b""")


def request_and_copy(A, p_expr_regs):
    """behind a command: the next command's record asked for, the copy in flight -- and, behind a command with literals, its literals: v118 /
    s89 lanes -- stored, this command's load issued, the loop closed.  Where wave 2 has not written that record yet (rare) the same store and load
    lie out of line and end the run with bit 0 of ok cleared.  p_expr_regs: (register that holds the output position in front of this command's copy,
    lanes of the store register or None, the register pair that holds the address of the copy's source -- in the output, or in the static dictionary)"""
    pcopy, merged, src = p_expr_regs[:3]
    off = p_expr_regs[3] if len(p_expr_regs) > 3 else "%[lane]"   # (the lanes' offsets into the copy's source: the lane, or lane mod distance)
    tell, back = L(), L()
    norec, look, have = L(), L(), L()

    def store_and_load():
        t = """
s_sub_u32 s96, %[P], %[pn]
s_add_u32 s96, %[outlo], s96
s_addc_u32 s97, %[outhi], 0
""" + VMWAIT
        if merged:
            t += """
s_bfm_b64 vcc, %[pn], 0
s_bfm_b64 exec, s89, 0
v_cndmask_b32 v118, v119, v124, vcc
global_store_byte %[lane], v118, s[96:97]"""
        else:
            t += """
s_bfm_b64 exec, %[pn], 0
global_store_byte %[lane], v124, s[96:97]"""
        return t + f"""
s_bfm_b64 exec, s92, 0
global_load_ubyte v124, {off}, {src}
s_mov_b64 exec, -1
s_mov_b32 %[pn], s92
s_add_u32 %[P], {pcopy}, s92"""
    A.m(f"""
s_add_u32 s95, s84, s82
s_andn2_b32 %[ok], %[ok], 2
s_sub_u32 s96, s95, %[said]
s_cmp_ge_u32 s96, 128
s_cbranch_scc1 {tell}f
{back}:
s_cmp_ge_u32 s95, %[front]
s_cbranch_scc1 {look}f
{have}:
s_and_b32 s95, s95, %[rmask]
s_lshl_b32 s95, s95, 3
s_add_u32 s95, s95, %[xring]
v_mov_b32 %[rx], s95
ds_read_b32 %[ry], %[rx] offset:4
ds_read_b32 %[rx], %[rx]""", "the two bytes before P are the copy's from here on; the next command's record, asked for before the memory pipe is waited for")
    A.m(store_and_load(), "the copy in flight goes to memory (no lanes: no store; behind literals: they go with it, lanes below pn the copy's bytes); this one's load (its bytes stay in v124 until the next command comes by)")
    A.m("s_branch 1b")
    A.o(f"""
{tell}:
s_mov_b32 %[said], s95
v_readlane_b32 s96, %[params], 5
v_mov_b32 v118, s95
v_mov_b32 v117, s96
ds_write_b32 v117, v118
s_branch {back}b""", "wave 2 is told where the reader is every 128 bits: it stays less than a lap ahead of that")
    A.o(f"""
{look}:
v_readlane_b32 s96, %[params], 8
v_mov_b32 v118, s96
ds_read_b64 v[116:117], v118
s_waitcnt lgkmcnt(0)
v_readfirstlane_b32 s96, v116
v_readfirstlane_b32 s97, v117
v_readlane_b32 s94, %[params], 9
s_cmp_lg_u32 s97, s94
s_cbranch_scc1 {norec}f
s_mov_b32 %[front], s96
s_cmp_lt_u32 s95, s96
s_cbranch_scc1 {have}b
{norec}:
s_andn2_b32 %[ok], %[ok], 1""", "the reader has reached the frontier it knew: where wave 2 has got to by now (XW_FRONT: positions below are written | the codes' epoch << 32); not there yet: the run ends behind this command")
    A.o(store_and_load())
    A.o("s_branch 90f")


def literal_loop(A, trivial, done):
    """the command's literals, two a round: the first takes the older context byte's register, the second the other one's -- the bytes do not
    move from register to register, and a round's first literal falls through its test.  A run that ends with a round's first literal swaps the two
    (out of line).  Ends at `done`."""
    top, odd = L(), L()

    def one(last, second, test):
        if not trivial:
            A.m(f"""
s_lshr_b32 s86, {last}, 2
s_lshr_b32 s87, {second}, 2
v_readlane_b32 s86, %[lut0], s86
v_readlane_b32 s87, %[lut1], s87
s_lshl_b32 s88, {last}, 3
s_lshl_b32 s96, {second}, 3
s_lshr_b32 s86, s86, s88
s_lshr_b32 s87, s87, s96
s_or_b32 s86, s86, s87
s_and_b32 s86, s86, 0xff
v_readlane_b32 s87, %[ctxtree], s86""", "context = lut0[the byte before] | lut1[the one before that] (four bytes a lane), its tree out of the map")
        else:
            A.m("s_mov_b32 s87, %[littree]")
        symbol(A, 15)
        A.m(f"""
s_mov_b32 {second}, s86
s_mov_b32 m0, s85
s_add_u32 s85, s85, 1
v_writelane_b32 v119, s86, m0
{test}""")
    A.m(f"{top}:")
    one("%[p1]", "%[p2]", f"s_cmp_ge_u32 s85, s89\ns_cbranch_scc1 {odd}f")
    one("%[p2]", "%[p1]", f"s_cmp_lt_u32 s85, s89\ns_cbranch_scc1 {top}b")
    A.o(f"""
{odd}:
s_mov_b32 s86, %[p1]
s_mov_b32 %[p1], %[p2]
s_mov_b32 %[p2], s86
s_branch {done}""")


def build():
    A = Asm()
    # ---------------- entry ----------------
    if PROFILE_WAIT:
        A.m("s_mov_b64 s[78:79], %[wacc]")
    A.m("""
s_bitcmp0_b32 %[ok], 0
s_cbranch_scc1 99f
s_sub_u32 s84, %[ndw], %[cb]
s_lshl_b32 s84, s84, 5
s_sub_u32 s84, s84, %[cnt]
s_sub_u32 s83, %[lim], %[cb]
s_sub_u32 s83, s83, 2
s_lshl_b32 s83, s83, 5
s_sub_u32 s83, s83, 1
s_lshl_b32 s82, %[cb], 5
s_sub_u32 s82, s82, %[org]""", "POS, LIMM1 (signed: a window that has no room left says so), RB")
    # ---------------- top of the loop ----------------
    A.m("""
.p2align 6
1:
s_waitcnt lgkmcnt(0)
v_readfirstlane_b32 s90, %[rx]
v_readfirstlane_b32 s91, %[ry]
s_sub_u32 s95, %[bl1], 1
s_sub_u32 s96, s83, s84
s_or_b32 s95, s95, s96
s_or_b32 s95, s95, s90
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 90f
s_bitcmp1_b32 s90, 24
s_cbranch_scc1 100f""", "one sign test: a command block's count that has run out, a window without room, a record that says 'not for this loop' (bit 31)")
    # ---------------- a command without literals, explicit distance (the common kind) ----------------
    kinds = L()
    A.m(f"""
s_and_b32 s92, s90, 0xffff
s_and_b32 s95, s90, 0x12000000
s_mov_b32 s93, s91
s_cmp_lg_u32 s95, 0
s_cbranch_scc1 {kinds}f
s_sub_u32 s94, %[bl2], 1""", "copy length; an implicit distance or a ring code: out of line; the explicit distance takes one of its block's count")
    A.m(plain_test("%[P]", "%[quota]"))
    word = L()
    A.m(f"""
s_or_b32 s95, s95, s94
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 {word}f
s_mov_b32 %[bl2], s94""")
    A.m(PUSH)
    A.m("""
3:
s_add_u32 s88, %[outlo], %[P]
s_addc_u32 s89, %[outhi], 0
s_sub_u32 s88, s88, s93
s_subb_u32 s89, s89, 0
4:
s_sub_u32 %[bl1], %[bl1], 1
s_sub_u32 %[quota], %[quota], s92
s_bfe_u32 s95, s90, 0x70010
s_add_u32 s84, s84, s95""", "the copy's source; counts; the reader moves on by the record's bits")
    request_and_copy(A, ("%[P]", None, "s[88:89]"))
    ov = L()
    A.o(f"{word}:")
    A.o(word_test("%[P]", "%[quota]", "90", "s88", "s89", ov))
    A.o("""
s_mov_b32 %[bl2], s94
s_branch 4b""", "a word: the distance's count, nothing pushed (decode.rs:2643-2644)")
    A.o(f"{ov}:")
    A.o(overlap_test("%[quota]", "90"))
    A.o("s_mov_b32 %[bl2], s94")
    A.o(PUSH)
    A.o("""
s_add_u32 s88, %[outlo], %[P]
s_addc_u32 s89, %[outhi], 0
s_sub_u32 s88, s88, s93
s_subb_u32 s89, s89, 0
s_sub_u32 %[bl1], %[bl1], 1
s_sub_u32 %[quota], %[quota], s92
s_bfe_u32 s95, s90, 0x70010
s_add_u32 s84, s84, s95""", "a copy that repeats itself: the same command, its load by lane mod distance")
    out_of_line(A, lambda: request_and_copy(A, ("%[P]", None, "s[88:89]", "v115")))
    # the other kinds of distance of a command without literals
    short, zero = L(), L()
    A.o(f"""
{kinds}:
s_bitcmp1_b32 s90, 25
s_cbranch_scc0 {short}f
s_mov_b32 s93, %[d0]""", "implicit: the last distance, nothing pushed, no count")
    A.o(plain_test("%[P]", "%[quota]"))
    A.o(f"""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 90f
s_branch 3b
{short}:
s_cmp_eq_u32 %[bl2], 0
s_cbranch_scc1 90f
s_cmp_eq_u32 s91, 0
s_cbranch_scc1 {zero}f""", "a ring code s91 = 0 .. 15")
    A.o(ring("s91"))
    A.o(plain_test("%[P]", "%[quota]"))
    A.o("""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 90f
s_sub_u32 %[bl2], %[bl2], 1""")
    A.o(PUSH)
    A.o(f"""
s_branch 3b
{zero}:
s_mov_b32 s93, %[d0]""")
    A.o(plain_test("%[P]", "%[quota]"))
    A.o("""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 90f
s_sub_u32 %[bl2], %[bl2], 1
s_branch 3b""")
    # ---------------- a command with literals: s91 = their number ----------------
    have_ctx, triv, long, long_back, keep_ctx = L(), L(), L(), L(), L()
    A.m(f"""
100:
s_cmp_ge_u32 s91, %[quota]
s_cbranch_scc1 90f
s_cmp_gt_u32 s91, %[bl0]
s_cbranch_scc1 90f
s_add_u32 s89, s91, %[pn]
s_bfe_u32 s95, s90, 0x70010
s_add_u32 s85, s84, s95
s_lshr_b32 s86, s85, 5
s_add_u32 s87, s86, 1
v_readlane_b32 s96, %[cur], s86
v_readlane_b32 s97, %[cur], s87
s_and_b32 s88, s85, 31
s_add_u32 %[ndw], s86, 2
s_lshr_b64 s[98:99], s[96:97], s88
s_sub_u32 %[cnt], 64, s88
s_bitcmp1_b32 %[ok], 1
s_cbranch_scc1 {keep_ctx}f
s_cmp_lt_u32 %[pn], 2
s_cbranch_scc1 90f
s_sub_u32 s96, %[pn], 1
s_sub_u32 s97, %[pn], 2
{VMWAIT}
v_readlane_b32 %[p1], v124, s96
v_readlane_b32 %[p2], v124, s97
{have_ctx}:
s_cmp_gt_u32 s89, 63
s_cbranch_scc1 {long}f
{long_back}:
s_mov_b32 s85, %[pn]
s_cmp_lg_u32 %[trivial], 0
s_cbranch_scc1 {triv}f""", "s89 = lanes of the store (the copy in flight and the literals behind it); the bit buffer from behind the head's bits on (made BEFORE the copy in flight is waited for: its last two bytes are the context); the two bytes before P: in p1 / p2 or that copy's tail; s85 = the lane of the next literal")
    A.o(f"""
{keep_ctx}:
s_mov_b32 s81, %[p1]
s_mov_b32 s80, %[p2]
s_branch {have_ctx}b""", "p1 / p2 are the two bytes before P already: kept, for a command that is not plain after all (where they are not, what a roll-back puts into them does not matter)")
    A.o(f"""
{long}:
s_cmp_gt_u32 s91, 63
s_cbranch_scc1 190f
s_cmp_eq_u32 %[pn], 0
s_cbranch_scc1 190f
s_sub_u32 s96, %[P], %[pn]
s_add_u32 s96, %[outlo], s96
s_addc_u32 s97, %[outhi], 0
s_waitcnt vmcnt(0)
s_bfm_b64 exec, %[pn], 0
global_store_byte %[lane], v124, s[96:97]
s_mov_b64 exec, -1
s_mov_b32 %[pn], 0
s_mov_b32 s89, s91
s_or_b32 %[ok], %[ok], 2
s_mov_b32 s81, %[p1]
s_mov_b32 s80, %[p2]
s_branch {long_back}b""", "more literals than fit one store behind the copy in flight: that copy goes to memory first (its last two bytes are in p1 / p2 by now, and stay the context whatever becomes of this command)")
    dist = L()
    literal_loop(A, False, dist + "b")
    A.m(f"{dist}:")
    A.o(f"{triv}:")
    # the trivial loop lives out of line: emit it into ool by swapping buffers
    main_keep, ool_keep = A.main, A.ool
    A.main, A.ool = [], []
    literal_loop(A, True, dist + "b")
    A.m(f"s_branch {dist}b")
    ool_keep.extend(A.main); ool_keep.extend(A.ool)   # (the loop, then what lies out of line of IT: its labels are looked for forward)
    A.main, A.ool = main_keep, ool_keep
    # the distance behind the literals
    implicit, small, join = L(), L(), L()
    extra_refill, extra_back = L(), L()
    A.m(f"""
s_bitcmp1_b32 s90, 25
s_cbranch_scc1 {implicit}f
s_cmp_eq_u32 %[bl2], 0
s_cbranch_scc1 190f
s_bfe_u32 s86, s90, 0x2001a
s_sub_u32 s94, %[bl2], 1
v_readlane_b32 s87, %[params], s86""", "the distance code: its context picks the table (lanes 0 .. 3 of the parameters)")
    symbol(A, 15)
    A.m(f"""
s_cmp_lt_u32 s86, 16
s_cbranch_scc1 {small}f
s_cmp_ge_u32 s86, 64
s_cbranch_scc1 190f
v_readlane_b32 s86, %[dlut], s86
s_cmp_lt_u32 %[cnt], 24
s_cbranch_scc1 {extra_refill}f
{extra_back}:
s_cmp_eq_u32 s86, 0
s_cbranch_scc1 190f
s_and_b32 s88, s86, 31
s_lshr_b32 s93, s86, 5
s_bfm_b32 s96, s88, 0
s_and_b32 s96, s98, s96
s_add_u32 s93, s93, s96
s_lshr_b64 s[98:99], s[98:99], s88
s_sub_u32 %[cnt], %[cnt], s88
s_and_b32 s92, s90, 0xffff
s_add_u32 s85, %[P], s91
s_sub_u32 s88, %[quota], s91""", "a code of the plain alphabet: base and number of extra bits out of the lane table (all zero where the alphabet is another)")
    A.m(plain_test("s85", "s88"))
    word2, join2 = L(), L()
    A.m(f"""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 {word2}f
s_mov_b32 %[bl2], s94""")
    A.m(PUSH)
    A.m(f"""
{join}:
s_add_u32 s86, %[outlo], s85
s_addc_u32 s87, %[outhi], 0
s_sub_u32 s86, s86, s93
s_subb_u32 s87, s87, 0
{join2}:
s_sub_u32 %[quota], s88, s92
s_sub_u32 %[bl0], %[bl0], s91
s_sub_u32 %[bl1], %[bl1], 1
s_lshl_b32 s95, %[ndw], 5
s_sub_u32 s84, s95, %[cnt]""", "it is a plain command: the copy's source, the counts, the reader's position")
    request_and_copy(A, ("s85", True, "s[86:87]"))
    ov2 = L()
    A.o(f"{word2}:")
    A.o(word_test("s85", "s88", "190", "s86", "s87", ov2))
    A.o(f"""
s_mov_b32 %[bl2], s94
s_branch {join2}b""")
    A.o(f"{ov2}:")
    A.o(overlap_test("s88", "190"))
    A.o("s_mov_b32 %[bl2], s94")
    A.o(PUSH)
    A.o("""
s_add_u32 s86, %[outlo], s85
s_addc_u32 s87, %[outhi], 0
s_sub_u32 s86, s86, s93
s_subb_u32 s87, s87, 0
s_sub_u32 %[quota], s88, s92
s_sub_u32 %[bl0], %[bl0], s91
s_sub_u32 %[bl1], %[bl1], 1
s_lshl_b32 s95, %[ndw], 5
s_sub_u32 s84, s95, %[cnt]""")
    out_of_line(A, lambda: request_and_copy(A, ("s85", True, "s[86:87]", "v115")))
    A.o(f"""
{extra_refill}:
s_cmp_gt_u32 %[ndw], 63
s_cbranch_scc1 190f
v_readlane_b32 s96, %[cur], %[ndw]
s_mov_b32 s97, 0
s_add_u32 %[ndw], %[ndw], 1
s_lshl_b64 s[96:97], s[96:97], %[cnt]
s_add_u32 %[cnt], %[cnt], 32
s_or_b64 s[98:99], s[98:99], s[96:97]
s_branch {extra_back}b""")
    # implicit distance / ring codes behind literals
    A.o(f"""
{implicit}:
s_mov_b32 s93, %[d0]
s_and_b32 s92, s90, 0xffff
s_add_u32 s85, %[P], s91
s_sub_u32 s88, %[quota], s91""")
    A.o(plain_test("s85", "s88"))
    zero2 = L()
    A.o(f"""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 190f
s_branch {join}b
{small}:
s_and_b32 s92, s90, 0xffff
s_add_u32 s85, %[P], s91
s_sub_u32 s88, %[quota], s91
s_cmp_eq_u32 s86, 0
s_cbranch_scc1 {zero2}f""")
    A.o(ring("s86"))
    A.o(plain_test("s85", "s88"))
    A.o("""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 190f
s_mov_b32 %[bl2], s94""")
    A.o(PUSH)
    A.o(f"""
s_branch {join}b
{zero2}:
s_mov_b32 s93, %[d0]""")
    A.o(plain_test("s85", "s88"))
    A.o(f"""
s_cmp_lt_i32 s95, 0
s_cbranch_scc1 190f
s_mov_b32 %[bl2], s94
s_branch {join}b""")
    # ---------------- the ways out ----------------
    tail = """
190:
s_mov_b32 %[p1], s81
s_mov_b32 %[p2], s80
90:
s_lshr_b32 s95, s84, 5
s_add_u32 s85, s95, 1
v_readlane_b32 s96, %[cur], s95
v_readlane_b32 s97, %[cur], s85
s_and_b32 s85, s84, 31
s_add_u32 %[ndw], s95, %[cb]
s_lshr_b64 s[98:99], s[96:97], s85
s_sub_u32 %[cnt], 64, s85
s_add_u32 %[ndw], %[ndw], 2
s_cmp_eq_u32 s85, 0
s_cselect_b32 %[cnt], 32, %[cnt]
s_cselect_b32 s99, 0, s99
s_cselect_b32 s96, 1, 0
s_sub_u32 %[ndw], %[ndw], s96
s_mov_b64 %[buf], s[98:99]
99:
s_waitcnt lgkmcnt(0)""" + ("\ns_mov_b64 %[wacc], s[78:79]" if PROFILE_WAIT else "")
    lines = A.main + A.ool + [(t.strip(), None) for t in tail.strip().split("\n")]
    return lines


def main():
    lines = build()
    out = ["// GENERATED by tools/gen_rec_asm.py -- do not edit; see that file for what the loop does, why it is laid out like this and which registers it keeps.",
           "#define LEAN_REC_RUN_ASM \\"]
    for i, (text, comment) in enumerate(lines):
        sep = "\\n" if text.endswith(":") else "\\n\\t"
        c = ("  /* " + comment + " */") if comment else ""
        out.append('  "%s%s"%s \\' % (text, sep, c))
    out[-1] = out[-1][:-2]
    path = os.path.join(ROOT, "rust-brotli-decompressor_amd", "csrc", "brotli_rec_run_asm_wait.h" if PROFILE_WAIT else "brotli_rec_run_asm.h")
    for a in sys.argv[1:]:
        if a.startswith("--out="):   # (tests: the committed header is what this script writes)
            path = a[len("--out="):]
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", path, len(lines), "lines")


if __name__ == "__main__":
    main()
