#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_encode_loop_pc.inc: the CODER half of the producer / consumer form of the (32,64),
P <= 12 ANS encoder (cst_ans_pc.hip, round 4).

65 536 streams are one wave per SIMD, and a lone wave issues one instruction per ~4.4 cycles whatever it is: in
cst_encode_loop.inc the coder chain's wave also stages the symbol tiles (8 x 16-byte LDS stores of one wave: 36 cycles
each), reads complete word groups back from its ring, stores them and requests the next tiles -- a quarter of the kernel
(scripts/ablate_encoder.sh: 0.228 ms without the tile work against 0.275 ms with it, all of it issue and LDS-store time of
the one wave, none of it memory time).  Here a second wave of the same workgroup -- the HELPER, compiler-scheduled C++ in
cst_ans_pc.hip, on the same SIMD -- does all of that, and the wave generated below runs nothing but the coder steps:

    quad g:  request the symbols of quad g-2 (one 16-B LDS read of the lane's tile row), fetch the four 16-B table
             entries of quad g-1, fold quad g's symbols into smin / smax (the range check), run quad g's four steps.
    words:   every step writes its candidate word to the lane's 64-slot LDS ring (layout [slot][lane]) and advances the
             write position if the word was really emitted (stack.rs:1035-1040), as in cst_encode_loop.inc.
    hand-off, once per tile at the top of quad 1 (every read of the current tile's row has returned by then):
             publish the write position, s_waitcnt lgkmcnt(0), s_barrier.
             Behind the barrier the helper has finished staging the NEXT tile into the other tile buffer (the coder's
             quads 1 and 0 already read it), may overwrite THIS tile's buffer with the tile after that, and moves complete
             64-byte groups below the published position from the ring to the slab.
             (A first version gave the coder per-tile OUT windows -- one VALU instruction less per step -- that the helper
             copied into its own ring: the SIMD's VALU is what both waves share, and the helper's 13 pushes per tile cost
             more of it than the coder saved: 0.296 ms against 0.278 for the one-wave kernel.)
Two tile buffers alternate; the loop body holds two tiles.

Run:  python scripts/gen_encode_loop_pc.py
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

CSRC = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc")
OUT = CSRC / "cst_encode_loop_pc.inc"
OUT_CK = CSRC / "cst_encode_loop_pc_ck.inc"         # ... noting jump points (round 5: see ck_hook below)
NO_BARRIER = bool(os.environ.get("GEN_NO_BARRIER"))     # timing experiment only (races with the helper)
PRIO = int(os.environ.get("GEN_PRIO", "2"))             # s_setprio of the coder's wave (0 = leave it alone)


def regs(base, n=4):
    return [f"v{base + i}" for i in range(n)]


def tup(base, n=4):
    return f"v[{base}:{base + n - 1}]"


S = [regs(100 + 4 * i) for i in range(4)]
S_T = [tup(100 + 4 * i) for i in range(4)]
E = [[regs(116 + 16 * e + 4 * i) for i in range(4)] for e in range(2)]
E_T = [[tup(116 + 16 * e + 4 * i) for i in range(4)] for e in range(2)]
A0, A1, W0, W1, U0, U1, T0, T1, SM0, SM1, Q0, Q1 = (f"v{r}" for r in range(148, 160))
A_T, W_T, U_T, T_T, SM_T, Q_T = (tup(148 + 2 * i, 2) for i in range(6))
RR, KK, CK, RA, EA0, EA1, WR = (f"v{r}" for r in range(160, 167))
SD = "s[84:85]"
CLOBBERS = [f"v{r}" for r in range(100, 170)] + ["s82", "s84", "s85", "vcc", "scc", "memory"]
ROW = ["%[row0]", "%[row1]"]
SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD"


LO, HI = "v168", "v169"          # the coder state lives in a register pair inside the statement: the step's last instruction writes both halves
ST_T = tup(168, 2)
OLD_TAIL = bool(os.environ.get("GEN_OLD_TAIL"))     # the step's tail as it was until round 4 (A + q k + c: two instructions more)


def step(a, e0, e1, m0, m1):
    """One coder step (stack.rs:1035-1045) on a packed table entry (see scripts/gen_encode_loop.py: step()).
    The new state is  q 2^P + r + c  with q = floor(A / p), r = A mod p.  From the estimate q_est in {q - 1, q} and
    r_est = A - q_est p in [0, 2 p):   q 2^P + r + c = q_est 2^P + r_est + (fix ? c + 2^P - p : c),   fix <=> r_est >= p
    -- ONE 64-bit mad (q_est_lo x 2^P + [r_est + c', q_est_hi << P]) where round 3 computed A + q (2^P - p) + c' with a mad, a
    24-bit mad and a 64-bit add: two instructions less on a wave that pays for every instruction it issues."""
    a.i(f"v_cmp_ge_u32_sdwa vcc, {HI}, {e1} src0_sel:WORD_1 src1_sel:WORD_1", "emit <=> (state >> (64 - P)) >= p")
    if OLD_TAIL:
        a.i(f"v_sub_u32_sdwa {KK}, %[twoP], {e1} {SDWA} src0_sel:DWORD src1_sel:WORD_0", "k = 2^P - p")
    a.i(f"v_lshlrev_b32 {RA}, 8, {WR}")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[c3f00], %[lanebase]")
    a.i(f"v_cndmask_b32_e64 {A0}, {LO}, {HI}, vcc")
    a.i(f"v_cndmask_b32_e64 {A1}, {HI}, 0, vcc")
    a.ds(f"ds_write_b32 {RA}, {LO}", "W", "candidate word, always written")
    a.i(f"v_addc_co_u32 {WR}, vcc, 0, {WR}, vcc")
    a.i(f"v_mul_hi_u32 {W0}, {A0}, {m0}")
    a.i(f"v_mad_u64_u32 {U_T}, vcc, {A1}, {m0}, {W_T}", "U = a1*m0 + hi32(a0*m0)   (< 2^64)")
    a.i(f"v_mad_u64_u32 {T_T}, vcc, {A0}, {m1}, {U_T}", "T = a0*m1 + U, carry -> vcc")
    a.i(f"v_mov_b32 {SM0}, {T1}")
    a.i(f"v_addc_co_u32 {SM1}, vcc, 0, {W1}, vcc", "[T_hi, carry]")
    a.i(f"v_mad_u64_u32 {Q_T}, vcc, {A1}, {m1}, {SM_T}", "q_est in {q - 1, q}")
    a.i(f"v_mul_u32_u24_sdwa {RR}, {Q0}, {e1} {SDWA} src0_sel:DWORD src1_sel:WORD_0", "low 24 bits of q_est times p")
    a.i(f"v_sub_u32 {RR}, {A0}, {RR}", "r_est modulo 2^24")
    a.i(f"v_cmp_ge_u32_sdwa vcc, {RR}, {e1} src0_sel:WORD_0 src1_sel:WORD_0", "fix <=> q = q_est + 1")
    if OLD_TAIL:
        a.i(f"v_mad_u64_u32 {U_T}, {SD}, {Q0}, {KK}, {A_T}", "A + q_lo * k")
        a.i(f"v_mad_u32_u24 {U1}, {Q1}, {KK}, {U1}", "      + (q_hi * k) << 32")
        a.i(f"v_cndmask_b32_sdwa {CK}, {e0}, {e0}, vcc {SDWA} src0_sel:WORD_0 src1_sel:WORD_1", "c, or c + k")
        a.i(f"v_add_co_u32 {LO}, vcc, {U0}, {CK}")
        a.i(f"v_addc_co_u32 {HI}, vcc, 0, {U1}, vcc")
        return
    a.i(f"v_cndmask_b32_sdwa {CK}, {e0}, {e0}, vcc {SDWA} src0_sel:WORD_0 src1_sel:WORD_1", "c, or c + 2^P - p")
    a.i(f"v_lshlrev_b32 {A1}, %[P], {Q1}", "q_est_hi << P   (q < 2^(64 - P))")
    a.i(f"v_add_u32_sdwa {A0}, {RR}, {CK} {SDWA} src0_sel:WORD_0 src1_sel:DWORD", "r_est + c'   (< 2^15)")
    a.i(f"v_mad_u64_u32 {ST_T}, {SD}, {Q0}, %[twoP], {A_T}", "state = q_est 2^P + r_est + c'")


# 12 < P <= 24 (round 5: the int8 / int16 coders at the reference's default precision): gen_encode_loop_wide.py's step on UNPACKED
# entries {c, p, floor(2^64 / p)} with this generator's registers -- 24 VALU instructions + the ring write.
WIDE = False
PSHL = KK


def step_wide(a, c, p, m0, m1):
    a.i(f"v_lshlrev_b32 {PSHL}, %[sh], {p}", "p << (32 - P)")
    a.i(f"v_cmp_ge_u32 vcc, {HI}, {PSHL}", "emit <=> (state >> (64 - P)) >= p")
    a.i(f"v_lshlrev_b32 {RA}, 8, {WR}")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[c3f00], %[lanebase]")
    a.i(f"v_cndmask_b32_e64 {A0}, {LO}, {HI}, vcc")
    a.i(f"v_cndmask_b32_e64 {A1}, {HI}, 0, vcc")
    a.ds(f"ds_write_b32 {RA}, {LO}", "W", "candidate word, always written")
    a.i(f"v_addc_co_u32 {WR}, vcc, 0, {WR}, vcc")
    a.i(f"v_mul_hi_u32 {W0}, {A0}, {m0}")
    a.i(f"v_mad_u64_u32 {U_T}, vcc, {A1}, {m0}, {W_T}", "U = a1*m0 + hi32(a0*m0)   (< 2^64)")
    a.i(f"v_mad_u64_u32 {T_T}, vcc, {A0}, {m1}, {U_T}", "T = a0*m1 + U, carry -> vcc")
    a.i(f"v_mov_b32 {SM0}, {T1}")
    a.i(f"v_addc_co_u32 {SM1}, vcc, 0, {W1}, vcc", "[T_hi, carry]")
    a.i(f"v_mad_u64_u32 {Q_T}, vcc, {A1}, {m1}, {SM_T}", "q_est in {q - 1, q}")
    a.i(f"v_mul_lo_u32 {RR}, {Q0}, {p}")
    a.i(f"v_sub_u32 {RR}, {A0}, {RR}", "r_est (< 2p: its low 32 bits)")
    a.i(f"v_sub_u32 {CK}, {RR}, {p}", "r_est - p   (wraps if r_est < p)")
    a.i(f"v_cmp_ge_u32 vcc, {RR}, {p}", "fix <=> q = q_est + 1")
    a.i(f"v_min_u32 {RR}, {RR}, {CK}", "r")
    a.i(f"v_cndmask_b32 {CK}, 0, %[twoP], vcc", "fix 2^P")
    a.i(f"v_lshlrev_b32 {A1}, %[P], {Q1}", "q_est_hi << P   (q < 2^(64 - P))")
    a.i(f"v_add3_u32 {A0}, {RR}, {c}, {CK}", "r + c + fix 2^P   (< 2^26)")
    a.i(f"v_mad_u64_u32 {ST_T}, {SD}, {Q0}, %[twoP], {A_T}", "state = q_est 2^P + r + c + fix 2^P")


def coder_step(a, c, p, m0, m1):
    (step_wide if WIDE else step)(a, c, p, m0, m1)


def read_syms(a, g, buf, quad):
    a.ds(f"ds_read_b128 {S_T[g % 4]}, {ROW[buf]} offset:{16 * quad}", f"S{g}")


def fetch_entries(a, g):
    x, y, z, w = S[g % 4]
    for i, sym in enumerate((w, z, y, x)):       # consumption order: .w first
        ea = (EA0, EA1)[i & 1]
        a.i(f"v_lshl_add_u32 {ea}, {sym}, 4, %[tbl]")
        a.ds(f"ds_read_b128 {E_T[g % 2][i]}, {ea}", f"E{g}")


def fold_minmax(a, g):
    """the range check: a symbol outside the model's support reads a garbage entry (harmless: LDS never faults) and flags its
    stream at the end.  (Tried: one v_max3_u32 per two table ADDRESSES with the table at LDS address 0 -- half the
    instructions, but (symbol - min) << 4 wraps for |symbol - min| >= 2^28 and such a symbol then passes as a valid one.)"""
    x, y, z, w = S[g % 4]
    a.i(f"v_max3_i32 %[smax], %[smax], {x}, {y}")
    a.i(f"v_max3_i32 %[smax], %[smax], {z}, {w}")
    a.i(f"v_min3_i32 %[smin], %[smin], {x}, {y}")
    a.i(f"v_min3_i32 %[smin], %[smin], {z}, {w}")


def hand_off(a):
    a.ds(f"ds_write_b32 %[pub], {WR}", "cnt", "words emitted so far")
    a.wait_lds_all("they are in the ring, and every read of this tile's row has returned")
    if not NO_BARRIER:
        a.i("s_barrier")


# Jump points (round 5: cst_encode_loop_pc_ck.inc / cst_encode_loop_pc_n8_ck.inc).  The reference's `AnsCoder::pos()` (stack.rs:1107-1139)
# in front of every chunk of K symbols, K a multiple of 32: a scalar countdown per tile; where a chunk starts the coder's wave
# leaves the statement's straight line for six instructions -- (words emitted so far, state) to d_ckpt_pos[s][j] / d_ckpt_state[s][j],
# j counting down -- and comes back.  The coder wave has no other vector-memory instruction, so there is no wait to keep.
JUMP = False
PO, SO = "v170", "v171"            # byte offsets of this lane's next jump point in the two arrays


def ck_hook(a, site):
    if not JUMP:
        return
    a.i("s_sub_u32 s89, s89, 1")
    a.i("s_cmp_eq_u32 s89, 0")
    a.i(f"s_cbranch_scc1 7{site}f", "a chunk starts here: note the jump point")
    a.i(f"8{site}:")


def ck_blocks(a, sites):
    if not JUMP:
        return
    a.i("s_branch 9f")
    for site in sites:
        a.i(f"7{site}:")
        a.i(f"global_store_dword {PO}, {WR}, %[ckpos]", "AnsCoder::pos(): words in the bulk ...")
        a.i(f"global_store_dwordx2 {SO}, {ST_T}, %[ckstate]", "... and the coder state")
        a.i(f"v_subrev_u32 {PO}, 4, {PO}")
        a.i(f"v_subrev_u32 {SO}, 8, {SO}")
        a.i("s_mov_b32 s89, %[cktiles]")
        a.i(f"s_branch 8{site}b")
    a.i("9:")


def ck_prologue(a):
    if not JUMP:
        return
    a.i(f"v_mov_b32 {PO}, %[ckposoff]")
    a.i(f"v_mov_b32 {SO}, %[ckstateoff]")
    a.i("s_mov_b32 s89, %[cktiles]", "tiles until the next jump point")


def ck_clobbers():
    return [PO, SO, "s89"] if JUMP else []


def ck_operands():
    return ', [ckpos] "s"(ckpt_pos), [ckstate] "s"(ckpt_state), [cktiles] "s"(ckpt_tiles), [ckposoff] "v"(ckpt_pos_off), [ckstateoff] "v"(ckpt_state_off)' if JUMP else ""


def half(a, h, g0, site=0):
    """one tile in tile buffer h, global quad indices g0 .. g0+7 stand for quads 7 .. 0"""
    a.i(f"; ---- tile in buffer {h}")
    for j in range(8):
        g, quad = g0 + j, 7 - j
        if quad == 1:
            hand_off(a)
        if f"S{g + 1}" in a.lds:
            a.wait_lds(f"S{g + 1}", f"quad {quad}: symbols of the next quad are back", cap=True)
        far = quad - 2
        read_syms(a, g + 2, h if far >= 0 else 1 - h, far if far >= 0 else far + 8)
        fetch_entries(a, g + 1)
        if f"E{g}" in a.lds:
            a.wait_lds(f"E{g}", f"entries of quad {quad} are back", cap=True)
        fold_minmax(a, g)
        for c, p, m0, m1 in E[g % 2]:
            coder_step(a, c, p, m0, m1)
    ck_hook(a, site)


def gen():
    a = Asm()
    a.i(f"v_mov_b32 {W1}, 0")
    a.i(f"v_mov_b32 {LO}, %[lo]")
    a.i(f"v_mov_b32 {HI}, %[hi]")
    if PRIO:
        a.i(f"s_setprio {PRIO}", "the coder chain's wave goes first on its SIMD; the helper fills the gaps")
    a.i(f"v_mov_b32 {WR}, 0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    ck_prologue(a)
    read_syms(a, 0, 0, 7)
    read_syms(a, 1, 0, 6)
    a.wait_lds("S0")
    fetch_entries(a, 0)
    a.i("1:")
    first = len(a.events)
    half(a, 0, 0, site=0)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    half(a, 1, 8, site=1)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    ren = {"S16": "S0", "S17": "S1", "E16": "E0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert lds_end == lds_back and vm_end == a.vm, (lds_end, lds_back)
    a.i("2:")
    a.ds(f"ds_write_b32 %[pub], {WR}", "cnt", "all words of the main loop")
    a.i(f"v_mov_b32 %[lo], {LO}")
    a.i(f"v_mov_b32 %[hi], {HI}")
    a.wait_lds_all()
    ck_blocks(a, (0, 1))
    return a, notes


def main():
    global JUMP, WIDE
    for WIDE, JUMP, out in ((False, False, OUT), (False, True, OUT_CK), (True, False, OUT_W), (True, True, OUT_W_CK)):
        emit_coder(out)
    JUMP = WIDE = False


def emit_coder(out):
    a, notes = gen()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Coder half of the producer / consumer (32,64) ANS encoder: see ans_encode_pc_coder_loop in cst_ans_pc.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [smin] "+v"(smin), [smax] "+v"(smax)',
           '    : [row0] "v"(tile_row_addr[0]), [row1] "v"(tile_row_addr[1]), [lanebase] "v"(ring_lane_addr), [pub] "v"(publish_addr), [tbl] "s"(table_bias),',
           '      [twoP] "v"(1u << P), [P] "s"(P), [c3f00] "s"(0x3f00u), [ntiles] "s"(n_tiles)' + (', [sh] "s"(32u - P)' if WIDE else '') + ck_operands(),
           "    : " + ", ".join(f'"{c}"' for c in CLOBBERS + ck_clobbers()) + ");"]
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


# ---------------------------------------------------------------------------------------------------------------------
# the HELPER half (cst_encode_loop_pc_helper.inc): everything but the coder steps, with as few VALU instructions as possible
# (the SIMD's VALU is what the two waves share; LDS, vector-memory and scalar instructions issue beside the coder's VALU
# stream).  hipcc's version of the same C++ took 107 VALU instructions per tile (copies of the prefetched registers, 64-bit
# address arithmetic, flat_* stores into the tile buffers) and waited vmcnt(0) in front of every staging.
#   window i (between the coder's barriers i - 1 and i):
#       tile i + 1: registers -> tile buffer (i + 1) & 1 (requested three windows earlier; free since barrier i - 1),
#       request tile i + 4 into the registers just freed (three register sets, so a window is 6 = lcm(2, 3) long),
#       one complete 64-byte group below the write position published at barrier i - 1: ring -> slab,
#       s_waitcnt lgkmcnt(0), s_barrier.
# Past the last tile the statement keeps re-requesting and re-staging tile n - 1 (nobody reads those buffers any more).
# ---------------------------------------------------------------------------------------------------------------------
OUT_HELPER = CSRC / "cst_encode_loop_pc_helper.inc"
NSETS = int(os.environ.get("GEN_NSETS", "3"))           # register sets of prefetched tiles: a tile is requested NSETS windows before it is staged
HBASE = 234 - 32 * NSETS
SETS = "ABCDE"[:NSETS]
HR = {n: [tup(HBASE + 32 * i + 4 * k) for k in range(8)] for i, n in enumerate(SETS)}
HFD = [(tup(234 + 4 * k, 2), tup(236 + 4 * k, 2), tup(234 + 4 * k)) for k in range(4)]
HWR, HNCH, HLIM, HFADDR, HFOFF = (f"v{r}" for r in range(250, 255))
HSAVE = "s[86:87]"
H_CLOBBERS = [f"v{r}" for r in range(HBASE, 255)] + [f"s{r}" for r in range(80, 90)] + ["vcc", "scc", "memory"]
HTR = ["%[tr0]", "%[tr1]"]


HLOAD_MOD = os.environ.get("GEN_HLOAD_MOD", "nt")       # cache policy of the symbol loads (experiment: "", "sc0", "sc1", "sc0 sc1", "nt sc1" ...)
HSTORE_MOD = os.environ.get("GEN_HSTORE_MOD", "")       # ... and of the word stores


def h_advance_base(a):
    """s[80:81] -> symbols of the next tile to request; stays on the last one (the FIRST 32 symbols of the rows) once every
    tile has been requested"""
    a.i("s_cmp_lg_u32 s83, 0")
    a.i("s_cselect_b32 s88, 0x80, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_sub_u32 s80, s80, s88")
    a.i("s_subb_u32 s81, s81, 0")
    a.i("s_sub_u32 s83, s83, s89")


def h_load_set(a, name):
    for k in range(8):
        a.vmem(f"global_load_dwordx4 {HR[name][k]}, %[goff{k}], s[80:81] {HLOAD_MOD}".rstrip(), f"ld{name}")
    h_advance_base(a)


def h_stage_set(a, name, buf):
    a.wait_vm(f"ld{name}", f"symbols in set {name} have arrived")
    for k in range(8):
        a.ds(f"ds_write_b128 {HTR[buf]}, {HR[name][k]} offset:{1152 * k}", "tl")


def h_flush(a):
    """one complete 64-byte group below the published write position: ring -> slab (64-byte aligned slabs of whole groups)"""
    a.ds(f"ds_read_b32 {HWR}, %[pub]", "pub")
    a.wait_lds("pub")
    a.i(f"v_sub_u32 {HNCH}, {HWR}, %[flushed]")
    a.i(f"v_lshrrev_b32 {HNCH}, 4, {HNCH}", "whole 16-word groups pending: 0 or 1")
    for k in range(4):
        a.i(f"v_add_lshl_u32 {HFADDR}, %[flushed], {4 * k}, 8")
        a.i(f"v_and_or_b32 {HFADDR}, {HFADDR}, %[c3f00], %[lanebase]")
        a.ds(f"ds_read2st64_b32 {HFD[k][0]}, {HFADDR} offset1:1", "fl")
        a.ds(f"ds_read2st64_b32 {HFD[k][1]}, {HFADDR} offset0:2 offset1:3", "fl")
    a.i(f"v_add_u32 {HLIM}, 16, %[flushed]")
    a.i(f"v_lshl_add_u32 {HFOFF}, %[flushed], 2, %[slaboff]")
    a.i(f"v_cmp_le_u32 vcc, {HLIM}, %[cap]", "group inside the slab")
    a.i(f"v_cmp_ne_u32 {HSAVE}, 0, {HNCH}")
    a.i(f"s_and_b64 vcc, vcc, {HSAVE}")
    a.wait_lds("fl")
    a.i(f"s_and_saveexec_b64 {HSAVE}, vcc")
    for k in range(4):
        if "nostores" in HABL:
            break
        a.vmem(f"global_store_dwordx4 {HFOFF}, {HFD[k][2]}, %[wbase] offset:{16 * k} {HSTORE_MOD}".rstrip(), "st")
    a.i(f"s_mov_b64 exec, {HSAVE}")
    a.i(f"v_lshl_add_u32 %[flushed], {HNCH}, 4, %[flushed]")


HABL = set((os.environ.get("GEN_HABL") or "").split("+")) - {""}      # timing experiments (results wrong): noflush, nostage, noloads, nostores


PAIRS = bool(os.environ.get("GEN_PAIRS"))     # experiment: request TWO tiles (256 contiguous bytes per row) every other window (needs NSETS = 5)


def h_window(a, w):
    name, buf = SETS[(w + 1) % NSETS], (w + 1) & 1
    if PAIRS:
        a.i(f"; ---- window {w}: set {name} -> tile buffer {buf}")
        h_stage_set(a, name, buf)
        if w % 2 == 0:
            h_load_set(a, SETS[w % NSETS])
            h_load_set(a, name)
        h_flush(a)
        a.wait_lds_all("the tile is staged")
        a.i("s_barrier")
        return
    a.i(f"; ---- window {w}: set {name} -> tile buffer {buf}")
    if "nostage" not in HABL:
        if "noloads" in HABL:
            for k in range(8):
                a.ds(f"ds_write_b128 {HTR[buf]}, {HR[name][k]} offset:{1152 * k}", "tl")
        else:
            h_stage_set(a, name, buf)
    if "noloads" not in HABL:
        h_load_set(a, name)
    if "noflush" not in HABL:
        h_flush(a)
    a.wait_lds_all("the tile is staged")
    a.i("s_barrier")


def gen_helper():
    a = Asm()
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST full tile of stream s0: tile 0")
    a.i("s_mov_b32 s82, %[ntiles]", "windows left")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    for n in SETS:
        h_load_set(a, n)
    h_stage_set(a, "A", 0)
    if not PAIRS:
        h_load_set(a, "A")
    a.wait_lds_all("tile 0 is staged (and the table, by everybody)")
    a.i("s_barrier")
    a.i("1:")
    first = len(a.events)
    period = NSETS * 2 if NSETS % 2 else NSETS
    for w in range(period):
        h_window(a, w)
        a.i("s_sub_u32 s82, s82, 1")
        if w < period - 1:
            a.i("s_cmp_eq_u32 s82, 0")
            a.i("s_cbranch_scc1 2f")
        else:
            a.i("s_cmp_lg_u32 s82, 0")
            a.i("s_cbranch_scc1 1b")
    lds_end, vm_end, notes = a.verify_loop(first, a.lds, a.vm, passes=1)
    assert HABL or (lds_end == a.lds and vm_end == a.vm), (vm_end, a.vm)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


def main_helper():
    a, notes = gen_helper()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Helper half of the producer / consumer (32,64) ANS encoder: see ans_encode_pc_helper_loop in cst_ans_pc.hip."]
    ops = ['    : [flushed] "+v"(flushed)',
           '    : [tr0] "v"(tile_tr_addr[0]), [tr1] "v"(tile_tr_addr[1]), [lanebase] "v"(ring_lane_addr), [pub] "v"(publish_addr),',
           '      [cap] "v"(cap), [slaboff] "v"(slab_off), [c3f00] "s"(0x3f00u), [wbase] "s"(words_base), [sbase] "s"(symbols_base), [ntiles] "s"(n_tiles),',
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in H_CLOBBERS) + ");"]
    OUT_HELPER.write_text(a.render(header, ops))
    print(f"wrote {OUT_HELPER} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


# ---------------------------------------------------------------------------------------------------------------------
# SPLIT helpers (cst_encode_loop_pc_loader.inc / cst_encode_loop_pc_storer.inc).  Loads and stores of one wave retire through
# ONE in-order counter (vmcnt): the combined helper's wait for a tile requested three windows ago also waits for the word
# stores it issued in between, and a store's acknowledgement under load takes longer than a window (measured: helper without
# its loads 0.251 ms, without its stores 0.254 ms, with both 0.275 ms).  So the four helper waves of a workgroup split by ROLE
# instead of by stream: waves 4 and 5 only LOAD and stage (each for two coder waves), waves 6 and 7 only FLUSH (each for two
# coder waves).  A loader's vmcnt sees loads only; a storer never waits for its stores inside the loop.
# ---------------------------------------------------------------------------------------------------------------------
OUT_LOADER = CSRC / "cst_encode_loop_pc_loader.inc"
OUT_STORER = CSRC / "cst_encode_loop_pc_storer.inc"
LSETS = int(os.environ.get("GEN_LSETS", "2"))           # register sets per coder wave in a loader (a tile is requested LSETS windows before it is staged)
LSLEEP = int(os.environ.get("GEN_LSLEEP", "0"))         # experiment: spread a window's 16 requests over the window (64 LSLEEP cycles after each)
LBASE = 228 - 64 * LSETS
LR = {(c, i): [tup(LBASE + 32 * (LSETS * c + i) + 4 * k) for k in range(8)] for c in range(2) for i in range(LSETS)}
L_CLOBBERS = [f"v{r}" for r in range(LBASE, 228)] + [f"s{r}" for r in range(80, 90)] + ["vcc", "scc", "memory"]
PAIR_TILE_OFF = 2 * 64 * 36 * 4                          # the partner coder's two tile buffers follow this one's


def l_load(a, i):
    for c in range(2):
        base = "s[80:81]" if c == 0 else "s[84:85]"
        for k in range(8):
            if "noloads" not in HABL:
                a.vmem(f"global_load_dwordx4 {LR[(c, i)][k]}, %[goff{c}_{k}], {base} {HLOAD_MOD}".rstrip(), f"ld{c}{i}")
                if LSLEEP:
                    a.i(f"s_sleep {LSLEEP}")
    a.i("s_cmp_lg_u32 s83, 0")
    a.i(f"s_cselect_b32 s88, {'0' if 'loadsame' in HABL else '0x80'}, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_sub_u32 s80, s80, s88")
    a.i("s_subb_u32 s81, s81, 0")
    a.i("s_sub_u32 s84, s84, s88")
    a.i("s_subb_u32 s85, s85, 0")
    a.i("s_sub_u32 s83, s83, s89")


def l_stage(a, i, buf):
    for c in range(2):
        if "noloads" not in HABL:
            a.wait_vm(f"ld{c}{i}", f"coder {c}: symbols in set {i} have arrived")
        for k in range(8):
            if "nostage" in HABL:                 # (timing experiments, results wrong: the loads are still waited for)
                continue
            if "stage32" in HABL:                 # a quarter of the register-file reads and LDS writes of the staging
                a.ds(f"ds_write_b32 {HTR[buf]}, v{LBASE + 32 * (LSETS * c + i) + 4 * k} offset:{PAIR_TILE_OFF * c + 1152 * k}", "tl")
                continue
            a.ds(f"ds_write_b128 {HTR[buf]}, {LR[(c, i)][k]} offset:{PAIR_TILE_OFF * c + 1152 * k}", "tl")


def gen_loader():
    a = Asm()
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST full tile of the pair's first stream: tile 0")
    a.i("s_add_u32 s84, s80, %[rowblock]", "... and of the second coder wave's first stream")
    a.i("s_addc_u32 s85, s81, 0")
    a.i("s_mov_b32 s82, %[ntiles]", "windows left")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    for i in range(LSETS):
        l_load(a, i)
    l_stage(a, 0, 0)
    l_load(a, 0)
    a.wait_lds_all("tile 0 is staged (and the table, by everybody)")
    a.i("s_barrier")
    a.i("1:")
    first = len(a.events)
    period = LSETS * 2 if LSETS % 2 else LSETS
    for w in range(period):
        i, buf = (w + 1) % LSETS, (w + 1) & 1
        a.i(f"; ---- window {w}: sets {i} -> tile buffers {buf}")
        l_stage(a, i, buf)
        l_load(a, i)
        a.wait_lds_all("the tiles are staged")
        a.i("s_barrier")
        a.i("s_sub_u32 s82, s82, 1")
        if w < period - 1:
            a.i("s_cmp_eq_u32 s82, 0")
            a.i("s_cbranch_scc1 2f")
        else:
            a.i("s_cmp_lg_u32 s82, 0")
            a.i("s_cbranch_scc1 1b")
    lds_end, vm_end, notes = a.verify_loop(first, a.lds, a.vm, passes=1)
    assert HABL or (lds_end == a.lds and vm_end == a.vm), (vm_end, a.vm)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


SQUAD = not os.environ.get("GEN_SROW")                  # lanes 4 j .. 4 j + 3 store the four 16-byte chunks of stream 16 k + j's group (one 64-byte request), k = 0 .. 3
SFD = {c: [(tup(196 + 16 * c + 4 * k, 2), tup(198 + 16 * c + 4 * k, 2), tup(196 + 16 * c + 4 * k)) for k in range(4)] for c in range(2)}
SWR, SNCH, SLIM, SFADDR, SFOFF, SXQ, SCOL0, SQOFF0 = ({c: f"v{r + 12 * c}" for c in range(2)} for r in range(228, 236))
SXS = {c: [f"v{236 + 12 * c + k}" for k in range(4)] for c in range(2)}
SC4I, SBPA = "v252", "v253"
SSAVE = {0: "s[86:87]", 1: "s[88:89]"}
SQBASE = ["%[wbase]", "s[90:91]", "s[92:93]", "s[94:95]"]
S_CLOBBERS = [f"v{r}" for r in range(196, 254)] + [f"s{r}" for r in range(80, 96)] + ["vcc", "scc", "memory"]


def s_flush_rows(a):
    """for both coder waves of the pair: one complete 64-byte group below the published write position, ring -> slab;
    every lane its own stream's group (four 16-byte pieces of ~26 different cache lines per store instruction)"""
    for c in range(2):
        a.ds(f"ds_read_b32 {SWR[c]}, %[pub{c}]", f"pub{c}")
    for c in range(2):
        a.wait_lds(f"pub{c}")
        a.i(f"v_sub_u32 {SNCH[c]}, {SWR[c]}, %[flushed{c}]")
        a.i(f"v_lshrrev_b32 {SNCH[c]}, 4, {SNCH[c]}", "whole 16-word groups pending: 0 or 1")
        for k in range(4):
            a.i(f"v_add_lshl_u32 {SFADDR[c]}, %[flushed{c}], {4 * k}, 8")
            a.i(f"v_and_or_b32 {SFADDR[c]}, {SFADDR[c]}, %[c3f00], %[lanebase{c}]")
            a.ds(f"ds_read2st64_b32 {SFD[c][k][0]}, {SFADDR[c]} offset1:1", f"fl{c}")
            a.ds(f"ds_read2st64_b32 {SFD[c][k][1]}, {SFADDR[c]} offset0:2 offset1:3", f"fl{c}")
        a.i(f"v_add_u32 {SLIM[c]}, 16, %[flushed{c}]")
        a.i(f"v_lshl_add_u32 {SFOFF[c]}, %[flushed{c}], 2, %[slaboff{c}]")
        if "storesame" in HABL:
            a.i(f"v_mov_b32 {SFOFF[c]}, %[slaboff{c}]")
    for c in range(2):
        a.i(f"v_cmp_le_u32 vcc, {SLIM[c]}, %[cap{c}]", "group inside the slab (capacity 0: a lane without a stream of its own)")
        a.i(f"v_cmp_ne_u32 {SSAVE[c]}, 0, {SNCH[c]}")
        a.i(f"s_and_b64 vcc, vcc, {SSAVE[c]}")
        a.wait_lds(f"fl{c}")
        a.i(f"s_and_saveexec_b64 {SSAVE[c]}, vcc")
        for k in range(4):
            if "nostores" in HABL:
                break
            a.vmem(f"global_store_dwordx4 {SFOFF[c]}, {SFD[c][k][2]}, %[wbase] offset:{16 * k} {HSTORE_MOD}".rstrip(), "st")
        a.i(f"s_mov_b64 exec, {SSAVE[c]}")
        a.i(f"v_lshl_add_u32 %[flushed{c}], {SNCH[c]}, 4, %[flushed{c}]")


def s_quad_invariants(a):
    """what depends on the lane only: lane 4 j + i moves chunk i of the streams 16 k + j"""
    a.i(f"v_mbcnt_lo_u32_b32 {SC4I}, -1, 0")
    a.i(f"v_mbcnt_hi_u32_b32 {SC4I}, -1, {SC4I}", "lane")
    a.i(f"v_and_b32 {SBPA}, 0xfc, {SC4I}", "4 (lane >> 2): ds_bpermute address of stream (lane >> 2)")
    for c in range(2):
        a.i(f"v_lshlrev_b32 {SCOL0[c]}, 2, {SC4I}")
        a.i(f"v_sub_u32 {SCOL0[c]}, %[lanebase{c}], {SCOL0[c]}", f"ring of coder wave {c}")
        a.i(f"v_add_u32 {SCOL0[c]}, {SCOL0[c]}, {SBPA}", "ring column of stream (lane >> 2)")
        a.ds(f"ds_bpermute_b32 {SQOFF0[c]}, {SBPA}, %[slaboff{c}]", "bp0")
    a.i(f"v_and_b32 {SC4I}, 3, {SC4I}")
    a.i(f"v_lshlrev_b32 {SC4I}, 2, {SC4I}", "4 (lane & 3): first of this lane's four words in a group")
    a.i("v_readlane_b32 s88, %[slaboff0], 16")
    a.i("v_readlane_b32 s89, %[slaboff0], 0")
    a.i("s_sub_u32 s88, s88, s89", "bytes from stream s to stream s + 16 (slabs are equally spaced)")
    a.i("s_mov_b64 s[90:91], %[wbase]")
    a.i("s_add_u32 s90, s90, s88")
    a.i("s_addc_u32 s91, s91, 0")
    a.i("s_add_u32 s92, s90, s88")
    a.i("s_addc_u32 s93, s91, 0")
    a.i("s_add_u32 s94, s92, s88")
    a.i("s_addc_u32 s95, s93, 0")
    a.wait_lds("bp0")
    for c in range(2):
        a.i(f"v_lshl_add_u32 {SQOFF0[c]}, {SC4I}, 2, {SQOFF0[c]}", "slab offset of stream (lane >> 2) + 16 (lane & 3)")


def s_flush_quads(a):
    """for both coder waves of the pair: the complete 64-byte groups below the published write positions, ring -> slab;
    lanes 4 j .. 4 j + 3 move the group of stream 16 k + j (one 64-byte request), k = 0 .. 3"""
    for c in range(2):
        a.ds(f"ds_read_b32 {SWR[c]}, %[pub{c}]", f"pub{c}")
    for c in range(2):
        a.wait_lds(f"pub{c}")
        a.i(f"v_sub_u32 {SNCH[c]}, {SWR[c]}, %[flushed{c}]")
        a.i(f"v_add_u32 {SLIM[c]}, 16, %[flushed{c}]")
        a.i(f"v_lshrrev_b32 {SNCH[c]}, 4, {SNCH[c]}", "whole 16-word groups pending: 0 or 1")
        if STORER_GROUPS > 1:
            a.i(f"v_min_u32 {SNCH[c]}, 1, {SNCH[c]}", "(12 < P <= 24: up to 39 words wait, ONE group leaves per pass)")
        a.i(f"v_cmp_le_u32 vcc, {SLIM[c]}, %[cap{c}]", "group inside the slab (capacity 0: a lane without a stream of its own)")
        a.i(f"v_cndmask_b32_e64 {SLIM[c]}, 0, {SNCH[c]}, vcc")
        a.i(f"v_lshl_or_b32 {SXQ[c]}, {SLIM[c]}, 31, %[flushed{c}]", "flush position | (a group leaves) << 31")
        for k in range(4):
            a.ds(f"ds_bpermute_b32 {SXS[c][k]}, {SBPA}, {SXQ[c]} offset:{64 * k}", f"bp{c}", f"... of stream {16 * k} + (lane >> 2)")
        a.i(f"v_lshl_add_u32 %[flushed{c}], {SNCH[c]}, 4, %[flushed{c}]", "(a group outside the slab is dropped: the stream ends CAPACITY)")
    for c in range(2):
        a.wait_lds(f"bp{c}")
        for k in range(4):
            a.i(f"v_add_lshl_u32 {SFADDR[c]}, {SXS[c][k]}, {SC4I}, 8", "(bit 31 leaves)")
            a.i(f"v_and_or_b32 {SFADDR[c]}, {SFADDR[c]}, %[c3f00], {SCOL0[c]}")
            a.ds(f"ds_read2_b32 {SFD[c][k][0]}, {SFADDR[c]} offset0:{16 * k} offset1:{64 + 16 * k}", f"fl{c}")
            a.ds(f"ds_read2_b32 {SFD[c][k][1]}, {SFADDR[c]} offset0:{128 + 16 * k} offset1:{192 + 16 * k}", f"fl{c}")
    for c in range(2):
        a.wait_lds(f"fl{c}")
        for k in range(4):
            a.i(f"v_lshl_add_u32 {SFOFF[c]}, {SXS[c][k]}, 2, {SQOFF0[c]}")
            if "storesame" in HABL:
                a.i(f"v_mov_b32 {SFOFF[c]}, {SQOFF0[c]}")
            a.i(f"v_cmp_gt_i32 vcc, 0, {SXS[c][k]}")
            a.i(f"s_and_saveexec_b64 {SSAVE[c]}, vcc")
            if "nostores" not in HABL:
                a.vmem(f"global_store_dwordx4 {SFOFF[c]}, {SFD[c][k][2]}, {SQBASE[k]} {HSTORE_MOD}".rstrip(), "st")
            a.i(f"s_mov_b64 exec, {SSAVE[c]}")


STORER_GROUPS = 1        # 64-byte groups a storer moves per tile and coder wave (2: the coders at 12 < P <= 24)


def gen_storer():
    a = Asm()
    a.i("s_mov_b32 s82, %[ntiles]", "windows left")
    if SQUAD:
        s_quad_invariants(a)
    a.wait_lds_all("nothing published yet (and the table, by everybody)")
    a.i("s_barrier")
    burst = int(os.environ.get("GEN_SBURST", "0"))      # experiment: flush only every `burst`-th window, `burst` groups then
    a.i("1:")
    if burst:
        for w in range(burst):
            if w == burst - 1:
                for _ in range(burst):
                    (s_flush_quads if SQUAD else s_flush_rows)(a)
            a.wait_lds_all()
            a.i("s_barrier")
            a.i("s_sub_u32 s82, s82, 1")
            if w < burst - 1:
                a.i("s_cmp_eq_u32 s82, 0")
                a.i("s_cbranch_scc1 2f")
        a.i("s_cmp_lg_u32 s82, 0")
        a.i("s_cbranch_scc1 1b")
        a.i("2:")
    else:
        for _ in range(STORER_GROUPS):
            (s_flush_quads if SQUAD else s_flush_rows)(a)
        a.wait_lds_all()
        a.i("s_barrier")
        a.i("s_sub_u32 s82, s82, 1")
        a.i("s_cmp_lg_u32 s82, 0")
        a.i("s_cbranch_scc1 1b")
    a.wait_vm_all("the stores of the loop (the wave's last groups follow from C++)")
    return a, []


def main_split():
    a, notes = gen_loader()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Loader wave of the producer / consumer (32,64) ANS encoder: see ans_encode_pc_loader_loop in cst_ans_pc.hip."]
    ops = ['    :',
           '    : [tr0] "v"(tile_tr_addr[0]), [tr1] "v"(tile_tr_addr[1]), [sbase] "s"(symbols_base), [rowblock] "s"(row_block_bytes), [ntiles] "s"(n_tiles),',
           '      ' + ", ".join(f'[goff0_{k}] "v"(goff0[{k}])' for k in range(8)) + ",",
           '      ' + ", ".join(f'[goff1_{k}] "v"(goff1[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in L_CLOBBERS) + ");"]
    OUT_LOADER.write_text(a.render(header, ops))
    print(f"wrote {OUT_LOADER} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)
    global STORER_GROUPS
    for STORER_GROUPS, out_storer in ((1, OUT_STORER), (2, OUT_STORER2)):
        emit_storer(out_storer)
    STORER_GROUPS = 1


def emit_storer(out_storer):
    a, notes = gen_storer()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Storer wave of the producer / consumer (32,64) ANS encoder: see ans_encode_pc_storer_loop in cst_ans_pc.hip."]
    ops = ['    : [flushed0] "+v"(flushed[0]), [flushed1] "+v"(flushed[1])',
           '    : [lanebase0] "v"(ring_lane_addr[0]), [lanebase1] "v"(ring_lane_addr[1]), [pub0] "v"(publish_addr[0]), [pub1] "v"(publish_addr[1]),',
           '      [cap0] "v"(cap[0]), [cap1] "v"(cap[1]), [slaboff0] "v"(slab_off[0]), [slaboff1] "v"(slab_off[1]), [c3f00] "s"(0x3f00u), [wbase] "s"(words_base), [ntiles] "s"(n_tiles)',
           "    : " + ", ".join(f'"{c}"' for c in S_CLOBBERS) + ");"]
    out_storer.write_text(a.render(header, ops))
    print(f"wrote {out_storer} ({a.n_instr()} instructions incl. prologue)")


# ---------------------------------------------------------------------------------------------------------------------
# INT8 symbol matrices (round 5: cst_encode_loop_pc_n8.inc / cst_encode_loop_pc_loader_n8.inc, ans_encode_pc_n8_kernel).
# A row of the matrix is int8, so a 128-byte line holds FOUR tiles.  The loader stages whole lines into byte tiles (rows of 128
# symbols + 4 bytes of padding: 33 words, conflict-free for the b32 accesses of both sides), two row blocks per window; the coder
# reads a quad as ONE dword and forms a table address per symbol with one SDWA shift of the sign-extended byte,
#     v_lshlrev_b32_sdwa ea, 4, sext(quad) src1_sel:BYTE_k     ->     ds_read_b128 entry, ea offset:2048
# against a 256-entry table centred at LDS address 2048 (the LDS address adder wraps: scripts/microbench/ds_wrap.hip) -- the
# instruction the int32 form spends on  symbol * 16 + table.  The range check folds the four ADDRESSES (16 * symbol, signed)
# instead of the four symbols.  Same steps, same ring, same hand-off; the storer waves are those of the int32 kernel.
# The loop body still holds two tiles: the tile pointers live in two registers that leapfrog (one v_add per tile; every fourth
# step jumps to the other line buffer: the jump alternates between J and 192 - J and comes from the scalar side).
# ---------------------------------------------------------------------------------------------------------------------
OUT_N8 = CSRC / "cst_encode_loop_pc_n8.inc"
OUT_LOADER_N8 = CSRC / "cst_encode_loop_pc_loader_n8.inc"
N8_ROW = 132                       # kN8RowBytes
N8_LINEBUF = 64 * N8_ROW           # one line buffer of a coder wave: 8448 bytes
S8 = [f"v{100 + i}" for i in range(4)]
EA8 = [[f"v{104 + 4 * e + i}" for i in range(4)] for e in range(2)]
PA, PB = "v112", "v113"
N8_CLOBBERS = [f"v{r}" for r in range(100, 114)] + [f"v{r}" for r in range(116, 170)] + ["s82", "s83", "s84", "s85", "s86", "s87", "s88", "vcc", "scc", "memory"]


def n8_read_syms(a, g, ptr, quad):
    a.ds(f"ds_read_b32 {S8[g % 4]}, {ptr} offset:{4 * quad}", f"S{g}")


def n8_fetch_entries(a, g):
    for i, b in enumerate((3, 2, 1, 0)):       # consumption order: the quad's last symbol first
        a.i(f"v_lshlrev_b32_sdwa {EA8[g % 2][i]}, %[four], sext({S8[g % 4]}) {SDWA} src0_sel:DWORD src1_sel:BYTE_{b}", "16 * symbol")
        a.ds(f"ds_read_b128 {E_T[g % 2][i]}, {EA8[g % 2][i]} offset:2048", f"E{g}")


def n8_fold_minmax(a, g):
    """the range check.  The table has an entry for EVERY int8 value, so the check is a flag: entries outside the model's support
    carry bit 15 in their first word (c < 2^12 in a real one) -- two v_or3_b32 per quad over the entries being CODED (not those
    fetched ahead: past the last tile they are whatever the line buffer holds) where the int32 form spends four min3 / max3 on
    the symbols.  The flagged stream's state is garbage from there on (its words are never used; nothing it addresses depends on
    the state).  GEN_N8_MINMAX=1: the first form, min / max over the table addresses 16 * symbol (0.239 against 0.234 ms)."""
    if os.environ.get("GEN_N8_MINMAX"):
        x, y, z, w = EA8[g % 2]
        a.i(f"v_max3_i32 %[smax], %[smax], {x}, {y}")
        a.i(f"v_max3_i32 %[smax], %[smax], {z}, {w}")
        a.i(f"v_min3_i32 %[smin], %[smin], {x}, {y}")
        a.i(f"v_min3_i32 %[smin], %[smin], {z}, {w}")
        return
    e = [E[g % 2][i][0] for i in range(4)]
    a.i(f"v_or3_b32 %[smax], %[smax], {e[0]}, {e[1]}", "bit 15: a symbol outside the support")
    a.i(f"v_or3_b32 %[smax], %[smax], {e[2]}, {e[3]}")


def n8_half(a, cur, nxt, g0, delta, site=0):
    """one tile at pointer `cur` (the next tile's at `nxt`), global quad indices g0 .. g0+7 stand for quads 7 .. 0"""
    a.i(f"; ---- tile at {cur}")
    for j in range(8):
        g, quad = g0 + j, 7 - j
        if quad == 1:
            hand_off(a)
        if f"S{g + 1}" in a.lds:
            a.wait_lds(f"S{g + 1}", f"quad {quad}: symbols of the next quad are back", cap=True)
        far = quad - 2
        n8_read_syms(a, g + 2, cur if far >= 0 else nxt, far if far >= 0 else far + 8)
        n8_fetch_entries(a, g + 1)
        if f"E{g}" in a.lds:
            a.wait_lds(f"E{g}", f"entries of quad {quad} are back", cap=True)
        n8_fold_minmax(a, g)
        for c, p, m0, m1 in E[g % 2]:
            coder_step(a, c, p, m0, m1)
    a.i(f"v_add_u32 {cur}, {delta}, {nxt}", "leapfrog: the tile after the next")
    ck_hook(a, site)


def gen_n8():
    a = Asm()
    a.i(f"v_mov_b32 {W1}, 0")
    a.i(f"v_mov_b32 {LO}, %[lo]")
    a.i(f"v_mov_b32 {HI}, %[hi]")
    if PRIO:
        a.i(f"s_setprio {PRIO}", "the coder chain's wave goes first on its SIMD; the helpers fill the gaps")
    a.i(f"v_mov_b32 {WR}, 0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i(f"v_add_u32 {PA}, 96, %[row0]", "tile 0: the last 32 symbols of the line in buffer 0")
    a.i(f"v_add_u32 {PB}, 64, %[row0]")
    a.i(f"s_mov_b32 s83, {N8_LINEBUF + 96}", "J: from the first tile of a line in buffer 0 to the last tile of the line in buffer 1 (192 - J: back)")
    a.i("s_mov_b32 s86, 0", "tile pairs done, mod 2")
    a.i("s_mov_b32 s87, -32")
    ck_prologue(a)
    n8_read_syms(a, 0, PA, 7)
    n8_read_syms(a, 1, PA, 6)
    a.wait_lds("S0")
    n8_fetch_entries(a, 0)
    a.i("1:")
    first = len(a.events)
    # tile A (even index i): the step behind it, P(i + 2) = P(i + 1) + d(i + 1), is -32 for i + 1 = 1 mod 4 and the jump for 3 mod 4
    n8_half(a, PA, PB, 0, "s87", site=0)
    a.i("s_xor_b32 s86, s86, 1")
    a.i("s_sub_u32 s88, 192, s83")
    a.i("s_cmp_eq_u32 s86, 1", "the NEXT pair's second tile is the first of its line: its successor lies in the other buffer")
    a.i("s_cselect_b32 s87, s83, -32")
    a.i("s_cmp_eq_u32 s86, 0", "a jump was just taken: the next one goes the other way")
    a.i("s_cselect_b32 s83, s88, s83")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    n8_half(a, PB, PA, 8, "-32", site=1)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    ren = {"S16": "S0", "S17": "S1", "E16": "E0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert lds_end == lds_back and vm_end == a.vm, (lds_end, lds_back)
    a.i("2:")
    a.ds(f"ds_write_b32 %[pub], {WR}", "cnt", "all words of the main loop")
    a.i(f"v_mov_b32 %[lo], {LO}")
    a.i(f"v_mov_b32 %[hi], {HI}")
    a.wait_lds_all()
    ck_blocks(a, (0, 1))
    return a, notes


def l8_load(a, i):
    for c in range(2):
        base = "s[80:81]" if c == 0 else "s[84:85]"
        for k in range(8):
            a.vmem(f"global_load_dwordx4 {LR[(c, i)][k]}, %[goff{c}_{k}], {base} {HLOAD_MOD}".rstrip(), f"ld{c}{i}")
    a.i("s_cmp_lg_u32 s83, 0")
    a.i("s_cselect_b32 s88, 0x80, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_sub_u32 s80, s80, s88")
    a.i("s_subb_u32 s81, s81, 0")
    a.i("s_sub_u32 s84, s84, s88")
    a.i("s_subb_u32 s85, s85, 0")
    a.i("s_sub_u32 s83, s83, s89")


def l8_stage(a, i, buf, blocks):
    """row blocks `blocks` of the line in register set i -> line buffer `buf` of both coder waves"""
    for c in range(2):
        for k in blocks:
            base = int(LR[(c, i)][k][2:].split(":")[0])
            for j in range(4):
                a.ds(f"ds_write_b32 {HTR[buf]}, v{base + j} offset:{2 * N8_LINEBUF * c + 8 * N8_ROW * k + 4 * j}", "tl")


def gen_loader_n8():
    assert LSETS == 2
    a = Asm()
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST line of the pair's first stream: line 0")
    a.i("s_add_u32 s84, s80, %[rowblock]", "... and of the second coder wave's first stream")
    a.i("s_addc_u32 s85, s81, 0")
    a.i("s_mov_b32 s82, %[ntiles]", "windows left")
    a.i("s_lshr_b32 s83, %[ntiles], 2")
    a.i("s_sub_u32 s83, s83, 1", "lines left to request")
    l8_load(a, 0)
    l8_load(a, 1)
    for c in range(2):
        a.wait_vm(f"ld{c}0", f"coder {c}: line 0 has arrived")
    l8_stage(a, 0, 0, range(8))
    l8_load(a, 0)
    a.wait_lds_all("line 0 is staged (and the table, by everybody)")
    a.i("s_barrier")
    a.i("1:")
    first = len(a.events)
    for w in range(8):
        line, v = w // 4 + 1, w % 4
        i = buf = line & 1
        a.i(f"; ---- window {w}: set {i}, row blocks {2 * v}, {2 * v + 1} -> line buffers {buf}")
        if v == 0:
            for c in range(2):
                a.wait_vm(f"ld{c}{i}", f"coder {c}: the line in set {i} has arrived")
        l8_stage(a, i, buf, (2 * v, 2 * v + 1))
        if v == 3:
            l8_load(a, i)
        a.wait_lds_all("the row blocks are staged")
        a.i("s_barrier")
        a.i("s_sub_u32 s82, s82, 1")
        if w < 7:
            a.i("s_cmp_eq_u32 s82, 0")
            a.i("s_cbranch_scc1 2f")
        else:
            a.i("s_cmp_lg_u32 s82, 0")
            a.i("s_cbranch_scc1 1b")
    lds_end, vm_end, notes = a.verify_loop(first, a.lds, a.vm, passes=1)
    assert lds_end == a.lds and vm_end == a.vm, (vm_end, a.vm)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


OUT_N8_CK = CSRC / "cst_encode_loop_pc_n8_ck.inc"
OUT_W = CSRC / "cst_encode_loop_pc_w.inc"                 # the int32 coders at 12 < P <= 24
OUT_W_CK = CSRC / "cst_encode_loop_pc_w_ck.inc"
OUT_N8W = CSRC / "cst_encode_loop_pc_n8w.inc"             # the same coders at 12 < P <= 24 (step_wide)
OUT_N8W_CK = CSRC / "cst_encode_loop_pc_n8w_ck.inc"
OUT_N16W = CSRC / "cst_encode_loop_pc_n16w.inc"
OUT_N16W_CK = CSRC / "cst_encode_loop_pc_n16w_ck.inc"
OUT_STORER2 = CSRC / "cst_encode_loop_pc_storer2.inc"     # ... and their storer: two 64-byte groups per tile (32 symbols can emit 24 words)


def main_n8():
    global JUMP, WIDE
    for JUMP, out in ((False, OUT_N8), (True, OUT_N8_CK)):
        emit_n8_coder(out)
    WIDE = True                                             # 12 < P <= 24
    for JUMP, out in ((False, OUT_N8W), (True, OUT_N8W_CK)):
        emit_n8_coder(out)
    JUMP = WIDE = False
    emit_n8_loader()


def emit_n8_coder(out):
    a, notes = gen_n8()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Coder half of the producer / consumer (32,64) ANS encoder for int8 symbol matrices: see ans_encode_pc_n8_coder_loop in cst_ans_pc.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [smin] "+v"(smin), [smax] "+v"(smax)',
           '    : [row0] "v"(line_row_addr), [lanebase] "v"(ring_lane_addr), [pub] "v"(publish_addr), [four] "v"(4u),',
           '      [twoP] "v"(1u << P), [P] "s"(P), [c3f00] "s"(0x3f00u), [ntiles] "s"(n_tiles)' + (', [sh] "s"(32u - P)' if WIDE else '') + ck_operands(),
           "    : " + ", ".join(f'"{c}"' for c in N8_CLOBBERS + ck_clobbers()) + ");"]
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


def emit_n8_loader():
    a, notes = gen_loader_n8()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Loader wave of the producer / consumer (32,64) ANS encoder for int8 symbol matrices: see ans_encode_pc_n8_loader_loop in cst_ans_pc.hip."]
    ops = ['    :',
           '    : [tr0] "v"(line_tr_addr[0]), [tr1] "v"(line_tr_addr[1]), [sbase] "s"(symbols_base), [rowblock] "s"(row_block_bytes), [ntiles] "s"(n_tiles),',
           '      ' + ", ".join(f'[goff0_{k}] "v"(goff0[{k}])' for k in range(8)) + ",",
           '      ' + ", ".join(f'[goff1_{k}] "v"(goff1[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in L_CLOBBERS) + ");"]
    OUT_LOADER_N8.write_text(a.render(header, ops))
    print(f"wrote {OUT_LOADER_N8} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


# ---------------------------------------------------------------------------------------------------------------------
# INT16 symbol matrices (cst_encode_loop_pc_n16{,_ck}.inc / cst_encode_loop_pc_loader_n16.inc, ans_encode_pc_n16_kernel).  A 128-byte
# line is 64 symbols: TWO tiles.  The coder reads a quad as two dwords (one ds_read2_b32) and forms a table address per symbol with
#     v_mad_i32_i16 ea, quad_half, 16, table - 16 min_symbol  op_sel:[k,0,0,0]        (k: the high or the low half of the dword)
# -- one instruction again, against the int32 form's table of at most 1024 entries at its ordinary place; the range check folds the
# ADDRESSES with unsigned min3 / max3 (an address below the table wraps to a huge one).  The tile pointers leapfrog as for int8; with
# two tiles per line EVERY second step jumps to the other line buffer (J, then 128 - J).  The loader stages a line in two windows.
# ---------------------------------------------------------------------------------------------------------------------
OUT_N16 = CSRC / "cst_encode_loop_pc_n16.inc"
OUT_N16_CK = CSRC / "cst_encode_loop_pc_n16_ck.inc"
OUT_LOADER_N16 = CSRC / "cst_encode_loop_pc_loader_n16.inc"
S16 = [(f"v{100 + 2 * i}", f"v{101 + 2 * i}", tup(100 + 2 * i, 2)) for i in range(4)]        # a quad: two dwords
N16_CLOBBERS = [f"v{r}" for r in range(100, 170)] + ["s82", "s83", "s84", "s85", "s87", "s88", "vcc", "scc", "memory"]
EA16 = [[f"v{108 + 4 * e + i}" for i in range(4)] for e in range(2)]                          # v108 .. v115 (PA, PB move to v170 / v171 ... see below)
PA16, PB16 = "v172", "v173"


def n16_read_syms(a, g, ptr, quad):
    a.ds(f"ds_read2_b32 {S16[g % 4][2]}, {ptr} offset0:{2 * quad} offset1:{2 * quad + 1}", f"S{g}")


def n16_fetch_entries(a, g):
    lo_reg, hi_reg, _ = S16[g % 4]
    for i, (reg, half) in enumerate(((hi_reg, 1), (hi_reg, 0), (lo_reg, 1), (lo_reg, 0))):      # consumption order: the quad's last symbol first
        a.i(f"v_mad_i32_i16 {EA16[g % 2][i]}, {reg}, 16, %[tbl] op_sel:[{half},0,0,0]", "table + 16 (symbol - min_symbol)")
        a.ds(f"ds_read_b128 {E_T[g % 2][i]}, {EA16[g % 2][i]}", f"E{g}")


def n16_fold_minmax(a, g):
    x, y, z, w = EA16[g % 2]
    a.i(f"v_max3_u32 %[smax], %[smax], {x}, {y}", "the range check, on the table addresses of the quad being coded")
    a.i(f"v_max3_u32 %[smax], %[smax], {z}, {w}")
    a.i(f"v_min3_u32 %[smin], %[smin], {x}, {y}")
    a.i(f"v_min3_u32 %[smin], %[smin], {z}, {w}")


def n16_half(a, cur, nxt, g0, delta, site=0):
    a.i(f"; ---- tile at {cur}")
    for j in range(8):
        g, quad = g0 + j, 7 - j
        if quad == 1:
            hand_off(a)
        if f"S{g + 1}" in a.lds:
            a.wait_lds(f"S{g + 1}", f"quad {quad}: symbols of the next quad are back", cap=True)
        far = quad - 2
        n16_read_syms(a, g + 2, cur if far >= 0 else nxt, far if far >= 0 else far + 8)
        n16_fetch_entries(a, g + 1)
        if f"E{g}" in a.lds:
            a.wait_lds(f"E{g}", f"entries of quad {quad} are back", cap=True)
        n16_fold_minmax(a, g)
        for c, p, m0, m1 in E[g % 2]:
            coder_step(a, c, p, m0, m1)
    a.i(f"v_add_u32 {cur}, {delta}, {nxt}", "leapfrog: the tile after the next")
    ck_hook(a, site)


def gen_n16():
    a = Asm()
    a.i(f"v_mov_b32 {W1}, 0")
    a.i(f"v_mov_b32 {LO}, %[lo]")
    a.i(f"v_mov_b32 {HI}, %[hi]")
    if PRIO:
        a.i(f"s_setprio {PRIO}", "the coder chain's wave goes first on its SIMD; the helpers fill the gaps")
    a.i(f"v_mov_b32 {WR}, 0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i(f"v_add_u32 {PA16}, 64, %[row0]", "tile 0: the last 32 symbols (64 bytes) of the line in buffer 0")
    a.i(f"v_mov_b32 {PB16}, %[row0]")
    a.i(f"s_mov_b32 s87, {N8_LINEBUF + 64}", "J: from the first tile of a line in buffer 0 to the last tile of the line in buffer 1 (128 - J: back)")
    ck_prologue(a)
    n16_read_syms(a, 0, PA16, 7)
    n16_read_syms(a, 1, PA16, 6)
    a.wait_lds("S0")
    n16_fetch_entries(a, 0)
    a.i("1:")
    first = len(a.events)
    # tile A (even index i) is the last tile of its line, tile B the first: P(i + 2) = P(i + 1) + jump, P(i + 3) = P(i + 2) - 64
    n16_half(a, PA16, PB16, 0, "s87", site=0)
    a.i("s_sub_u32 s87, 128, s87", "the next jump goes the other way")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    n16_half(a, PB16, PA16, 8, "-64", site=1)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    ren = {"S16": "S0", "S17": "S1", "E16": "E0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert lds_end == lds_back and vm_end == a.vm, (lds_end, lds_back)
    a.i("2:")
    a.ds(f"ds_write_b32 %[pub], {WR}", "cnt", "all words of the main loop")
    a.i(f"v_mov_b32 %[lo], {LO}")
    a.i(f"v_mov_b32 %[hi], {HI}")
    a.wait_lds_all()
    ck_blocks(a, (0, 1))
    return a, notes


def gen_loader_n16():
    assert LSETS == 2
    a = Asm()
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST line of the pair's first stream: line 0")
    a.i("s_add_u32 s84, s80, %[rowblock]", "... and of the second coder wave's first stream")
    a.i("s_addc_u32 s85, s81, 0")
    a.i("s_mov_b32 s82, %[ntiles]", "windows left")
    a.i("s_lshr_b32 s83, %[ntiles], 1")
    a.i("s_sub_u32 s83, s83, 1", "lines left to request")
    l8_load(a, 0)
    l8_load(a, 1)
    for c in range(2):
        a.wait_vm(f"ld{c}0", f"coder {c}: line 0 has arrived")
    l8_stage(a, 0, 0, range(8))
    l8_load(a, 0)
    a.wait_lds_all("line 0 is staged (and the table, by everybody)")
    a.i("s_barrier")
    a.i("1:")
    first = len(a.events)
    for w in range(4):
        line, v = w // 2 + 1, w % 2
        i = buf = line & 1
        a.i(f"; ---- window {w}: set {i}, row blocks {4 * v} .. {4 * v + 3} -> line buffers {buf}")
        if v == 0:
            for c in range(2):
                a.wait_vm(f"ld{c}{i}", f"coder {c}: the line in set {i} has arrived")
        l8_stage(a, i, buf, range(4 * v, 4 * v + 4))
        if v == 1:
            l8_load(a, i)
        a.wait_lds_all("the row blocks are staged")
        a.i("s_barrier")
        a.i("s_sub_u32 s82, s82, 1")
        if w < 3:
            a.i("s_cmp_eq_u32 s82, 0")
            a.i("s_cbranch_scc1 2f")
        else:
            a.i("s_cmp_lg_u32 s82, 0")
            a.i("s_cbranch_scc1 1b")
    lds_end, vm_end, notes = a.verify_loop(first, a.lds, a.vm, passes=1)
    assert lds_end == a.lds and vm_end == a.vm, (vm_end, a.vm)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


def main_n16():
    global JUMP, WIDE
    for WIDE, JUMP, out in ((False, False, OUT_N16), (False, True, OUT_N16_CK), (True, False, OUT_N16W), (True, True, OUT_N16W_CK)):
        a, notes = gen_n16()
        header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
                  "// Coder half of the producer / consumer (32,64) ANS encoder for int16 symbol matrices: see ans_encode_pc_n16_coder_loop in cst_ans_pc.hip."]
        ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [smin] "+v"(smin), [smax] "+v"(smax)',
               '    : [row0] "v"(line_row_addr), [lanebase] "v"(ring_lane_addr), [pub] "v"(publish_addr), [tbl] "s"(table_bias),',
               '      [twoP] "v"(1u << P), [P] "s"(P), [c3f00] "s"(0x3f00u), [ntiles] "s"(n_tiles)' + (', [sh] "s"(32u - P)' if WIDE else '') + ck_operands(),
               "    : " + ", ".join(f'"{c}"' for c in N16_CLOBBERS + [PA16, PB16] + ck_clobbers()) + ");"]
        out.write_text(a.render(header, ops))
        print(f"wrote {out} ({a.n_instr()} instructions incl. prologue)")
    JUMP = WIDE = False
    a, notes = gen_loader_n16()
    header = ["// GENERATED by scripts/gen_encode_loop_pc.py -- do not edit by hand (edit the generator and re-run it).",
              "// Loader wave of the producer / consumer (32,64) ANS encoder for int16 symbol matrices: see ans_encode_pc_n16_loader_loop in cst_ans_pc.hip."]
    ops = ['    :',
           '    : [tr0] "v"(line_tr_addr[0]), [tr1] "v"(line_tr_addr[1]), [sbase] "s"(symbols_base), [rowblock] "s"(row_block_bytes), [ntiles] "s"(n_tiles),',
           '      ' + ", ".join(f'[goff0_{k}] "v"(goff0[{k}])' for k in range(8)) + ",",
           '      ' + ", ".join(f'[goff1_{k}] "v"(goff1[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in L_CLOBBERS) + ");"]
    OUT_LOADER_N16.write_text(a.render(header, ops))
    print(f"wrote {OUT_LOADER_N16} ({a.n_instr()} instructions incl. prologue)")


def main_all():
    main()
    main_helper()
    main_split()
    main_n8()
    main_n16()


if __name__ == "__main__":
    main_all()
