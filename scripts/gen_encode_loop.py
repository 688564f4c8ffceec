#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_encode_loop.inc: the hand-scheduled gfx950 main loop of the (32,64) ANS
encoder -- ONE asm statement that encodes all full 32-symbol tiles of a wave's 64 streams, last tile first.

The steps run as one continuous software pipeline of quads (4 symbols) that does not drain at tile boundaries:
    quad g:  request the symbols of quad g-2 (one 16-B LDS read of the lane's tile row),
             fetch the four 16-B table entries of quad g-1,
             fold quad g's symbols into smin/smax, run its four 23-instruction coder steps (packed table entries:
             see step()).
Two LDS tile buffers alternate, so quads 1 and 0 of a tile already read the NEXT tile's row.  Everything else is
hung into fixed places of a tile ("half": the loop body holds two, one per register set of prefetched symbols):
    before quad 6 : read the 64-byte group of compressed words that may be complete in the lane's LDS ring;
    after quad 5  : wait for the NEXT tile's symbols (requested two tiles earlier, 8 x 16 B per lane, transposed
                    mapping), write them to the other tile buffer, store the word group (exec-masked; slabs are
                    64-byte aligned on this path), request the symbols of tile - 3 into the registers just freed.
All stores of a tile are issued BEFORE its loads, so the only operations younger than the loads a tile waits for are
the other register set's eight loads: the same s_waitcnt operand is right for the first pass and the steady state.
asmgen.Asm keeps the lgkmcnt / vmcnt book and verify_loop() re-checks every operand against the steady state.

Run:  python scripts/gen_encode_loop.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

CSRC = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc")
OUT = CSRC / "cst_encode_loop.inc"
OUT_SINGLE = CSRC / "cst_encode_loop_1buf.inc"
OUT_SM = CSRC / "cst_encode_loop_sm.inc"
# SINGLE: ONE tile buffer per wave and a 32-slot word ring (cst_encode_loop_1buf.inc): 17 KiB of LDS per wave instead of
# 34, so that two workgroups share a CU when a batch has more than one wave per SIMD.  The next tile is staged between
# the last read of the current tile (quad 0's symbols, requested in quad 2) and the first read of the next one (its
# quad 7, requested in quad 1) -- one wave's LDS operations execute in order.
SINGLE = False
# SYMBOL_MAJOR (cst_encode_loop_sm.inc): symbols[t][stream].  Only the staging differs: a load instruction reads 16 rows
# (t) of 64 bytes (16 streams), lane l = row l >> 2, streams 4 (l & 3) .. + 3 -- coalesced 64-byte pieces whose other
# halves the next instruction reads -- and the four symbols of a register go to four ROWS of the LDS tile
# (tile[stream][t], stride 36 words): with this lane mapping the 64 lanes of a ds_write_b32 hit 64 different banks.
# The base moves by 32 * n_streams * 4 bytes per tile (an operand) instead of 128.
SYMBOL_MAJOR = False


# Timing-only ablations (results are wrong), scripts/exp_variants.sh: what the coder chain's wave would cost if a helper
# wave did the tile work.  GEN_NO_TILEWORK: both tile buffers staged once, no loads / staging / ring reads / stores in the
# loop; GEN_LOCAL_RING: the write position restarts at 0 in every tile (one v_lshl_add for the slot address, no mask);
# GEN_ADDR_MINMAX: range check by one v_max3_u32 per two table addresses; GEN_BARRIER: one s_barrier per tile.
NO_TILEWORK = bool(os.environ.get("GEN_NO_TILEWORK"))
LOCAL_RING = bool(os.environ.get("GEN_LOCAL_RING"))
ADDR_MINMAX = bool(os.environ.get("GEN_ADDR_MINMAX"))
BARRIER = bool(os.environ.get("GEN_BARRIER"))
# Code placement experiments (results stay right): GEN_SHIFT4 puts one s_nop in front of the loop (every instruction of the
# body 4 bytes later); GEN_E64 encodes every VOP2 instruction as VOP3 (8 bytes: all VALU / DS instructions then start on
# 8-byte boundaries relative to each other).
# QUAD_FLUSH (round 4): a complete 64-byte group of a stream's words leaves for HBM through FOUR lanes (16 bytes each), sixteen
# streams per store instruction, instead of through its own lane as "chunk k of every stream" (64 lanes, 64 different lines,
# 16 bytes each: 300-390 cycles of memory-pipeline time per instruction for a lone wave, microbench/scatter_cost.hip; the
# ablation scripts/ablate_encoder.sh prices the old flush at a quarter of the kernel).  Lane 4j + i moves chunk i of stream
# 16 k + j in pass k: it fetches that stream's flush position with ds_bpermute, reads its four words from the ring column of
# the stream and stores them -- whole 64-byte segments per quad of lanes.  GEN_OLD_FLUSH=1 generates the previous form (A/B).
QUAD_FLUSH = not os.environ.get("GEN_OLD_FLUSH")
# finer timing-only ablations of the tile work (results wrong): GEN_ABL = comma list of stores,loads,stage,flushreads
ABL = set((os.environ.get("GEN_ABL") or "").split(",")) - {""}
ADDR_MODE = os.environ.get("GEN_ADDR_MODE")        # ring-address experiments: bfe | vmask
WR_RESET = bool(os.environ.get("GEN_WR_RESET"))    # zero the write position once per tile although the ring address is masked
RING_MASK = os.environ.get("GEN_RING_MASK")        # e.g. 0x1f00: a 32-slot ring (experiment)
SHIFT4 = int(os.environ.get("GEN_SHIFT4") or 0)
E64 = bool(os.environ.get("GEN_E64"))


def regs(base, n=4):
    return [f"v{base + i}" for i in range(n)]


def tup(base, n=4):
    return f"v[{base}:{base + n - 1}]"


R = {"A": [tup(100 + 4 * k) for k in range(8)], "B": [tup(132 + 4 * k) for k in range(8)]}
S = [regs(164 + 4 * i) for i in range(4)]                    # four symbol quads (three live + one being filled)
S_T = [tup(164 + 4 * i) for i in range(4)]
E = [[regs(180 + 16 * e + 4 * i) for i in range(4)] for e in range(2)]
E_T = [[tup(180 + 16 * e + 4 * i) for i in range(4)] for e in range(2)]
A0, A1, W0, W1, U0, U1, T0, T1, SM0, SM1, Q0, Q1 = (f"v{r}" for r in range(212, 224))
A_T, W_T, U_T, T_T, SM_T, Q_T = (tup(212 + 2 * i, 2) for i in range(6))
RR, PSHL, LO, HI, RA, EA = (f"v{r}" for r in range(224, 230))     # (LO, HI): the coder state, a register pair inside the statement
ST_T = tup(226, 2)
CK = "v255"
KK = None                            # (k = 2^P - p had a register until round 4: the step no longer needs it)
FD = [(tup(230 + 4 * k, 2), tup(232 + 4 * k, 2), tup(230 + 4 * k)) for k in range(4)]
NCH, LIM, FADDR, FOFF = "v246", "v247", "v248", "v249"
# quad flush: X = flush position | pending << 31 of the lane's own stream; XS[k] = the same of stream 16 k + (lane >> 2);
# loop invariants: C4I = 4 (lane & 3), BPA = 4 (lane >> 2), COL0 = ring address of column (lane >> 2), QOFF0 = slab offset of
# stream (lane >> 2) + 16 (lane & 3); s[90:95] = words base + 16 k slabs (k = 1..3).  At quad boundaries the step's
# temporaries are free: FADDR / FOFF / LIM live there.
XQ, C4I, BPA, COL0, QOFF0 = "v225", "v250", "v251", "v252", "v253"
XS = ["v247", "v248", "v249", "v254"]
QFADDR, QFOFF, QLIM = A0, A1, W0
QBASE = ["%[wbase]", "s[90:91]", "s[92:93]", "s[94:95]"]
SD, SAVE = "s[84:85]", "s[86:87]"
CLOBBERS = [f"v{r}" for r in range(100, 256)] + [f"s{r}" for r in range(80, 96)] + ["vcc", "scc", "memory"]
ROW = ["%[row0]", "%[row1]"]          # the lane's own row in tile buffer 0 / 1
TR = ["%[tr0]", "%[tr1]"]             # transposed write address in tile buffer 0 / 1


SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD"


def step(a, e0, e1, m0, m1):
    """One coder step (stack.rs:1035-1045) on a PACKED table entry  e0 = c | (c + 2^P - p) << 16,  e1 = p | p << (32 - P),
    (m0, m1) = floor(2^64 / p): the operands derived from c and p are SDWA half-word selects.
        emit  <=>  (state >> (64 - P)) >= p  <=>  (hi >> 16) >= p << (16 - P)    (P <= 12)
        A = emit ? state >> 32 : state;  q_est = mulhi64(A, m) in {q - 1, q};  r_est = A - q_est * p  (< 2p < 2^13: its low 16 bits
        follow from the low 24 bits of q_est alone, one v_mul_u32_u24);  fix <=> r_est >= p
        state' = (q << P) + c + r = (q_est << P) + r_est + (fix ? c + 2^P - p : c)
    20 VALU instructions + the ring write.  A lone wave executes one instruction after another, dependent or not
    (4.0 cycles of issue + 0.3-1 of operand fetch each: DESIGN.md 3.8), so what counts is the NUMBER of instructions."""
    a.i(f"v_cmp_ge_u32_sdwa vcc, {HI}, {e1} src0_sel:WORD_1 src1_sel:WORD_1", "emit <=> (state >> (64 - P)) >= p")
    if LOCAL_RING:
        a.i(f"v_lshl_add_u32 {RA}, %[wr], 8, %[lanebase]")
    elif ADDR_MODE == "bfe":                    # two instructions, no SGPR operand
        a.i(f"v_bfe_u32 {RA}, %[wr], 0, 4")
        a.i(f"v_lshl_add_u32 {RA}, {RA}, 8, %[lanebase]")
    elif ADDR_MODE == "vmask":                  # the mask from a VGPR
        a.i(f"v_lshlrev_b32 {RA}, 8, %[wr]")
        a.i(f"v_and_or_b32 {RA}, {RA}, v254, %[lanebase]")
    else:
        a.i(f"v_lshlrev_b32 {RA}, 8, %[wr]")
        a.i(f"v_and_or_b32 {RA}, {RA}, %[c3f00], %[lanebase]")
    a.i(f"v_cndmask_b32_e64 {A0}, {LO}, {HI}, vcc")
    a.i(f"v_cndmask_b32_e64 {A1}, {HI}, 0, vcc")
    a.ds(f"ds_write_b32 {RA}, {LO}", "W", "candidate word, always written")
    a.i(f"v_addc_co_u32 %[wr], vcc, 0, %[wr], vcc")
    # q_est = floor(A * m / 2^64) = a1*m1 + floor((a1*m0 + a0*m1 + hi32(a0*m0)) / 2^32), the middle sum taken to 65 bits
    a.i(f"v_mul_hi_u32 {W0}, {A0}, {m0}")
    a.i(f"v_mad_u64_u32 {U_T}, vcc, {A1}, {m0}, {W_T}", "U = a1*m0 + hi32(a0*m0)   (< 2^64)")
    a.i(f"v_mad_u64_u32 {T_T}, vcc, {A0}, {m1}, {U_T}", "T = a0*m1 + U, carry -> vcc")
    a.i(f"v_mov_b32 {SM0}, {T1}")
    a.i(f"v_addc_co_u32 {SM1}, vcc, 0, {W1}, vcc", "[T_hi, carry]")
    a.i(f"v_mad_u64_u32 {Q_T}, vcc, {A1}, {m1}, {SM_T}", "q_est in {q - 1, q}")
    a.i(f"v_mul_u32_u24_sdwa {RR}, {Q0}, {e1} {SDWA} src0_sel:DWORD src1_sel:WORD_0", "low 24 bits of q_est times p")
    a.i(f"v_sub_u32 {RR}, {A0}, {RR}", "r_est modulo 2^24")
    a.i(f"v_cmp_ge_u32_sdwa vcc, {RR}, {e1} src0_sel:WORD_0 src1_sel:WORD_0", "fix <=> q = q_est + 1")
    # state' = q 2^P + r + c = q_est 2^P + r_est + (fix ? c + 2^P - p : c): ONE 64-bit mad on [r_est + c', q_est_hi << P]
    # (until round 4: A + q_est (2^P - p) + c' with a 64-bit mad, a 24-bit mad, a 64-bit add and the subtraction for 2^P - p:
    # two instructions more)
    a.i(f"v_cndmask_b32_sdwa {CK}, {e0}, {e0}, vcc {SDWA} src0_sel:WORD_0 src1_sel:WORD_1", "c, or c + 2^P - p")
    a.i(f"v_lshlrev_b32 {A1}, %[P], {Q1}", "q_est_hi << P   (q < 2^(64 - P))")
    a.i(f"v_add_u32_sdwa {A0}, {RR}, {CK} {SDWA} src0_sel:WORD_0 src1_sel:DWORD", "r_est + c'   (< 2^15)")
    a.i(f"v_mad_u64_u32 {ST_T}, {SD}, {Q0}, %[twoP], {A_T}", "state = q_est 2^P + r_est + c'")


def read_syms(a, g, buf, quad):
    a.ds(f"ds_read_b128 {S_T[g % 4]}, {ROW[0 if SINGLE else buf]} offset:{16 * quad}", f"S{g}")


def fetch_entries(a, g):
    x, y, z, w = S[g % 4]
    for i, sym in enumerate((w, z, y, x)):       # consumption order: .w first
        ea = f"v{250 + i}" if ADDR_MINMAX else EA
        a.i(f"v_lshl_add_u32 {ea}, {sym}, 4, %[tbl]")
        a.ds(f"ds_read_b128 {E_T[g % 2][i]}, {ea}", f"E{g}")
    if ADDR_MINMAX:
        a.i("v_max3_u32 %[smax], %[smax], v250, v251")
        a.i("v_max3_u32 %[smax], %[smax], v252, v253")


def fold_minmax(a, g):
    if ADDR_MINMAX:
        return
    x, y, z, w = S[g % 4]
    a.i(f"v_max3_i32 %[smax], %[smax], {x}, {y}")
    a.i(f"v_max3_i32 %[smax], %[smax], {z}, {w}")
    a.i(f"v_min3_i32 %[smin], %[smin], {x}, {y}")
    a.i(f"v_min3_i32 %[smin], %[smin], {z}, {w}")


def advance_base(a):
    """s[80:81] -> symbols of the next tile to request; stays on tile 0 once every tile has been requested"""
    a.i("s_cmp_lg_u32 s83, 0")
    a.i("s_cselect_b32 s88, %[tilestep], 0" if SYMBOL_MAJOR else "s_cselect_b32 s88, 0x80, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_sub_u32 s80, s80, s88")
    a.i("s_subb_u32 s81, s81, 0")
    a.i("s_sub_u32 s83, s83, s89")


def load_set(a, name):
    if "loads" in ABL and len(a.lines) > 120:
        advance_base(a)
        return
    for k in range(8):
        a.vmem(f"global_load_dwordx4 {R[name][k]}, %[goff{k}], s[80:81] nt", f"ld{name}")
    advance_base(a)


import os


def stage_set(a, name, buf):
    if "stage" in ABL and len(a.lines) > 120:
        return
    if not ("loads" in ABL and len(a.lines) > 120):
        a.wait_vm(f"ld{name}", f"symbols in set {name} have arrived")
    if os.environ.get("GEN_NO_VMWAIT") and len(a.lines) > 100:      # timing experiment only: results are wrong
        a.lines.pop()
    if SYMBOL_MAJOR:
        base = {"A": 100, "B": 132}[name]
        for k in range(8):
            for c in range(4):
                a.ds(f"ds_write_b32 {TR[buf]}, v{base + 4 * k + c} offset:{(16 * (k >> 1) + c) * 144 + 64 * (k & 1)}", f"tl{k}")
            if k >= 2:
                a.wait_lds(f"tl{k - 2}", cap=True)      # (lgkmcnt only counts to 15: at most 8 of these writes stay in flight)
        return
    for k in range(8):
        a.ds(f"ds_write_b128 {TR[0 if SINGLE else buf]}, {R[name][k]} offset:{1152 * k}", "tl")


def flush_own_lane(a, quad, part):
    """round 1-3: every lane moves the complete 64-byte group of its own stream (4 x 16 bytes, "chunk k of every stream")"""
    if part == "reads" and quad in (7, 6):
        # ring reads of the 64-byte group (4 chunks, two per quad: lgkmcnt counts only to 15) that may be complete.
        # Words leave for HBM 64 bytes at a time: 16-byte stores reach DRAM as partial bursts (measured 1.6x write
        # amplification); at most 15 + 12 words are ever pending, so one group per tile is enough and the 64-slot
        # ring holds the backlog.
        if quad == 7:
            # decide NOW whether the group is complete: words written after these reads must not count
            a.i(f"v_sub_u32 {NCH}, %[wr], %[flushed]")
            a.i(f"v_lshrrev_b32 {NCH}, 4, {NCH}", "whole 16-word groups pending: 0 or 1")
        for k in ((0, 1) if quad == 7 else (2, 3)):
            a.i(f"v_add_lshl_u32 {FADDR}, %[flushed], {4 * k}, 8")
            a.i(f"v_and_or_b32 {FADDR}, {FADDR}, %[c3f00], %[lanebase]")
            a.ds(f"ds_read2st64_b32 {FD[k][0]}, {FADDR} offset1:1", "fl")
            a.ds(f"ds_read2st64_b32 {FD[k][1]}, {FADDR} offset0:2 offset1:3", "fl")
    if part == "stores":
        a.i(f"v_add_u32 {LIM}, 16, %[flushed]")
        a.i(f"v_lshl_add_u32 {FOFF}, %[flushed], 2, %[slaboff]")
        a.i(f"v_cmp_le_u32 vcc, {LIM}, %[cap]", "group inside the slab (cap % 16 == 0 on this path)")
        a.i(f"v_cmp_ne_u32 {SAVE}, 0, {NCH}")
        a.i(f"s_and_b64 vcc, vcc, {SAVE}")
        a.wait_lds("fl", cap=True)
        a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
        for k in range(4):
            a.vmem(f"global_store_dwordx4 {FOFF}, {FD[k][2]}, %[wbase] offset:{16 * k}", "st")
        a.i(f"s_mov_b64 exec, {SAVE}")
        a.i(f"v_lshl_add_u32 %[flushed], {NCH}, 4, %[flushed]")


def quad_flush_invariants(a):
    """prologue of the quad flush: what depends on the lane only"""
    a.i(f"v_mbcnt_lo_u32_b32 {C4I}, -1, 0")
    a.i(f"v_mbcnt_hi_u32_b32 {C4I}, -1, {C4I}", "lane")
    a.i(f"v_and_b32 {BPA}, 0xfc, {C4I}", "4 (lane >> 2): ds_bpermute address of stream (lane >> 2)")
    a.i(f"v_lshlrev_b32 {COL0}, 2, {C4I}")
    a.i(f"v_sub_u32 {COL0}, %[lanebase], {COL0}", "the wave's ring")
    a.i(f"v_add_u32 {COL0}, {COL0}, {BPA}", "ring column of stream (lane >> 2)")
    a.i(f"v_and_b32 {C4I}, 3, {C4I}")
    a.i(f"v_lshlrev_b32 {C4I}, 2, {C4I}", "4 (lane & 3): first of this lane's four words in a group")
    a.ds(f"ds_bpermute_b32 {QOFF0}, {BPA}, %[slaboff]", "bp0")
    a.i("v_readlane_b32 s88, %[slaboff], 16")
    a.i("v_readlane_b32 s89, %[slaboff], 0")
    a.i("s_sub_u32 s88, s88, s89", "bytes from stream s to stream s + 16 (slabs are equally spaced)")
    a.i("s_mov_b64 s[90:91], %[wbase]")
    a.i("s_add_u32 s90, s90, s88")
    a.i("s_addc_u32 s91, s91, 0")
    a.i("s_add_u32 s92, s90, s88")
    a.i("s_addc_u32 s93, s91, 0")
    a.i("s_add_u32 s94, s92, s88")
    a.i("s_addc_u32 s95, s93, 0")
    a.wait_lds("bp0")
    a.i(f"v_lshl_add_u32 {QOFF0}, {C4I}, 2, {QOFF0}", "slab offset of stream (lane >> 2) + 16 (lane & 3)")


def flush_quad(a, quad, part):
    """the complete 64-byte group of stream 16 k + (lane >> 2) through lanes 4 (lane >> 2) .. + 3, k = 0 .. 3"""
    if part == "reads" and quad == 7:
        # decide NOW whether the group is complete: words written after the ring reads must not count
        a.i(f"v_sub_u32 {NCH}, %[wr], %[flushed]")
        a.i(f"v_add_u32 {QLIM}, 16, %[flushed]")
        a.i(f"v_lshrrev_b32 {NCH}, 4, {NCH}", "whole 16-word groups pending: 0 or 1")
        a.i(f"v_cmp_le_u32 vcc, {QLIM}, %[cap]", "group inside the slab (cap % 16 == 0 on this path)")
        a.i(f"v_cndmask_b32_e64 {QLIM}, 0, {NCH}, vcc")
        a.i(f"v_lshl_or_b32 {XQ}, {QLIM}, 31, %[flushed]", "flush position | (a group leaves) << 31")
        for k in range(4):
            a.ds(f"ds_bpermute_b32 {XS[k]}, {BPA}, {XQ} offset:{64 * k}", "bp", f"... of stream {16 * k} + (lane >> 2)")
        a.i(f"v_lshl_add_u32 %[flushed], {NCH}, 4, %[flushed]")
    if part == "reads" and quad == 6 and "flushreads" not in ABL:
        a.wait_lds("bp", cap=True)
        for k in range(4):
            a.i(f"v_add_lshl_u32 {QFADDR}, {XS[k]}, {C4I}, 8", "(bit 31 leaves)")
            a.i(f"v_and_or_b32 {QFADDR}, {QFADDR}, %[c3f00], {COL0}")
            a.ds(f"ds_read2_b32 {FD[k][0]}, {QFADDR} offset0:{16 * k} offset1:{64 + 16 * k}", "fl")
            a.ds(f"ds_read2_b32 {FD[k][1]}, {QFADDR} offset0:{128 + 16 * k} offset1:{192 + 16 * k}", "fl")
    if part == "stores":
        if "flushreads" not in ABL:
            a.wait_lds("fl", cap=True)
        for k in range(4):
            a.i(f"v_lshl_add_u32 {QFOFF}, {XS[k]}, 2, {QOFF0}")
            a.i(f"v_cmp_gt_i32 vcc, 0, {XS[k]}")
            a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
            if "stores" not in ABL:
                a.vmem(f"global_store_dwordx4 {QFOFF}, {FD[k][2]}, {QBASE[k]}", "st")
            a.i(f"s_mov_b64 exec, {SAVE}")


def half(a, h, g0):
    """one tile: register set / tile buffer h (0 = A), global quad indices g0 .. g0+7 stand for quads 7 .. 0"""
    own, other = "AB"[h], "AB"[1 - h]
    a.i(f"; ---- tile in buffer {h} (symbols came from set {own})")
    for j in range(8):
        g, quad = g0 + j, 7 - j
        # the pipeline runs on into the next tile: quads "-1" and "-2" are quads 7 and 6 of the other buffer
        if f"S{g + 1}" in a.lds:        # (quad 4: already retired by the wait in front of the tile staging)
            a.wait_lds(f"S{g + 1}", f"quad {quad}: symbols of the next quad are back", cap=True)
        far = quad - 2
        read_syms(a, g + 2, h if far >= 0 else 1 - h, far if far >= 0 else far + 8)
        if SINGLE and quad == 2:
            stage_set(a, other, 1 - h)
            load_set(a, other)
        fetch_entries(a, g + 1)
        if f"E{g}" in a.lds:
            a.wait_lds(f"E{g}", f"entries of quad {quad} are back", cap=True)
        if not NO_TILEWORK or ("add_q7" in ABL and quad == 7):
            (flush_quad if QUAD_FLUSH else flush_own_lane)(a, quad, "reads")
        fold_minmax(a, g)
        for c, p, m0, m1 in E[g % 2]:
            step(a, c, p, m0, m1)
        if quad == 5 and NO_TILEWORK:
            if "add_q5" in ABL:
                for k in range(4):
                    a.i(f"v_lshl_add_u32 {QFOFF}, {XS[k]}, 2, {QOFF0}")
                    a.i(f"v_cmp_gt_i32 vcc, 0, {XS[k]}")
                    a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
                    a.i(f"s_mov_b64 exec, {SAVE}")
            if "add_wait" in ABL:
                a.wait_lds(f"E{g + 1}", "(early)")
            if "add_base" in ABL:
                advance_base(a)
            if LOCAL_RING or WR_RESET:
                a.i("v_mov_b32 %[wr], 0")
            if BARRIER:
                a.i("s_barrier")
        if quad == 5 and "resetwr" in ABL:
            a.i("v_mov_b32 %[wr], 0")
            a.i("v_mov_b32 %[flushed], 0")
        if quad == 5 and not NO_TILEWORK:
            # word group -> slab; next tile's symbols -> the other tile buffer; request tile - 3 into the freed registers
            (flush_quad if QUAD_FLUSH else flush_own_lane)(a, quad, "stores")
            if not SINGLE:
                a.wait_lds(f"E{g + 1}", "(early: keeps the eight tile writes below within lgkmcnt's range of 15)")
                stage_set(a, other, 1 - h)
                load_set(a, other)


def gen():
    a = Asm()
    a.i(f"v_mov_b32 {W1}, 0")
    a.i(f"v_mov_b32 {LO}, %[lo]")
    a.i(f"v_mov_b32 {HI}, %[hi]")
    if QUAD_FLUSH:
        quad_flush_invariants(a)
    if ADDR_MODE == "vmask":
        a.i("v_mov_b32 v254, 0x3f00")
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST full tile of stream s0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    load_set(a, "A")                  # last tile
    load_set(a, "B")                  # the one before
    stage_set(a, "A", 0)
    if NO_TILEWORK:
        stage_set(a, "B", 1)
        a.wait_lds_all()
    else:
        load_set(a, "A")                  # two before
    read_syms(a, 0, 0, 7)
    read_syms(a, 1, 0, 6)
    a.wait_lds("S0")
    fetch_entries(a, 0)
    for _ in range(SHIFT4):
        a.i("s_nop 0")
    a.i("1:")
    first = len(a.events)
    half(a, 0, 0)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    half(a, 1, 8)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    # the body is 16 quads: symbol sets (mod 4) and entry sets (mod 2) are back in phase; rename the tags of the
    # quads in flight to what the top of the body expects and verify every wait against the steady state
    ren = {"S16": "S0", "S17": "S1", "E16": "E0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert ABL or (lds_end == lds_back and vm_end == a.vm), (lds_end, lds_back, vm_end, a.vm)
    a.i("2:")
    a.i(f"v_mov_b32 %[lo], {LO}")
    a.i(f"v_mov_b32 %[hi], {HI}")
    if "shortdrain" in ABL:
        a.i("v_and_b32 %[flushed], -16, %[wr]")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


def to_e64(text):
    """VOP2 -> VOP3 encoding of the same instruction (8 bytes instead of 4)"""
    op = text.split()[0]
    if op in ("v_lshlrev_b32", "v_mov_b32", "v_sub_u32", "v_add_u32", "v_add_co_u32", "v_addc_co_u32", "v_lshrrev_b32"):
        return text.replace(op, op + "_e64", 1)
    return text


def emit(out, single, symbol_major=False):
    global SINGLE, SYMBOL_MAJOR
    SINGLE, SYMBOL_MAJOR = single, symbol_major
    a, notes = gen()
    if E64:
        a.lines = [(to_e64(t), c) for t, c in a.lines]
    header = ["// GENERATED by scripts/gen_encode_loop.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (32,64) ANS encoder: see ans_encode_tiles_loop in cst_ans_kernels.hpp."]
    if single:
        header[1] = "// Main loop of the hand-scheduled (32,64) ANS encoder, ONE tile buffer and a 32-slot ring per wave: see cst_ans_small.hip."
        ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr), [flushed] "+v"(flushed), [smin] "+v"(smin), [smax] "+v"(smax)',
               '    : [row0] "v"(tile_row_addr), [tr0] "v"(tile_tr_addr),']
    else:
        ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr), [flushed] "+v"(flushed), [smin] "+v"(smin), [smax] "+v"(smax)',
               '    : [row0] "v"(tile_row_addr[0]), [row1] "v"(tile_row_addr[1]), [tr0] "v"(tile_tr_addr[0]), [tr1] "v"(tile_tr_addr[1]),']
    ops += ['      [lanebase] "v"(ring_lane_addr), [cap] "v"(cap), [slaboff] "v"(slab_off),',
            '      [tbl] "s"(table_addr_biased), [twoP] "v"(1u << P), [P] "s"(P), [c3f00] "s"(' + ("ring_mask" if single else (RING_MASK + "u" if RING_MASK else "0x3f00u")) + '), [wbase] "s"(words_base),',
            '      [sbase] "s"(symbols_base), [ntiles] "s"(n_tiles),' + (' [tilestep] "s"(tile_step_bytes),' if symbol_major else ''),
            '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
            "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


def main():
    emit(OUT, False)
    if NO_TILEWORK or LOCAL_RING or ADDR_MINMAX or BARRIER or SHIFT4 or E64 or RING_MASK or ADDR_MODE or WR_RESET or ABL or not QUAD_FLUSH:      # (ablations of the main loop only)
        return
    emit(OUT_SINGLE, True)
    emit(OUT_SM, False, True)


if __name__ == "__main__":
    main()
