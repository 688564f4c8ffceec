#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_pt_encode_loop.inc: the hand-scheduled gfx950 main loop of the ANS encoder
with ONE TABLE PER STREAM in its compact form (cst_ans_pt.hip, BASELINE config C3), (W,S) = (32,64), P <= 12.

Same skeleton as gen_encode_loop.py (one asm statement for all full 32-symbol tiles of a wave's 64 streams, 64-byte
word groups to the slab, symbols requested two tiles ahead, the 24-instruction coder step; ONE LDS tile buffer), but
the table entry of a symbol takes two dependent LDS reads, so the quad pipeline is one stage deeper:
    quad g:  request the symbols of quad g-3 (one 16-B LDS read of the lane's tile row),
             quad g-2: symbol -> index i, t = clamp(i, a, b), d = i - t, request c[t] and c[t+1] (two 16-bit reads:
                       a 32-bit read at a 2-byte aligned address is served 5x slower),
             quad g-1: c = c[t] + d, p = c[t+1] - c[t], request floor(2^64 / p) from the workgroup's reciprocal table,
             run the four coder steps of quad g.
A symbol outside the support clamps to a valid row position (garbage c, valid p) and is reported through smin/smax.

Run:  python scripts/gen_pt_encode_loop.py   (rewrites the .inc; the .inc is checked in)
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

import os

# GEN_PT_CK=1 (round 5): the same loop with JUMP POINTS -- the reference's Pos side information (stack.rs:1107-1139): every
# `cktiles` tiles (one chunk of the stream) the lanes note (words in the bulk, coder state) = AnsCoder::pos() in front of that
# chunk.  A scalar countdown and a branch per tile; the three instructions of a checkpoint run k times per stream, end with
# vmcnt(0) (they are not in the generator's book: waiting for MORE than the book knows is always safe) and cost nothing
# measurable.  The compressed words are those of the plain loop.
CKPT = bool(os.environ.get("GEN_PT_CK"))
# GEN_PT_N8=1 (round 6, with GEN_PT_CK): the same loop over an INT8 symbol matrix (C3's support -127 .. 127 fits the type; the reference's
# Symbol is generic, quantize.rs:229-255).  A tile is 32 BYTES of a row: one global_load_dword per lane and quad into the quad's last
# register, sign-extended into the quad (three v_bfe_i32 and a shift) in front of the transposed ds_write_b128.  cst_pt_encode_loop_ck_n8.inc
N8 = bool(os.environ.get("GEN_PT_N8"))
assert not N8 or CKPT
OUT = Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc" / \
    ("cst_pt_encode_loop_ck_n8.inc" if N8 else "cst_pt_encode_loop_ck.inc" if CKPT else "cst_pt_encode_loop.inc")


def regs(base, n=4):
    return [f"v{base + i}" for i in range(n)]


def tup(base, n=4):
    return f"v[{base}:{base + n - 1}]"


BASE = 88
R = {"A": [tup(BASE + 4 * k) for k in range(8)], "B": [tup(BASE + 32 + 4 * k) for k in range(8)]}
S = [regs(BASE + 64 + 4 * i) for i in range(4)]              # four symbol quads; a quad's registers later hold d = i - t
S_T = [tup(BASE + 64 + 4 * i) for i in range(4)]
E = [[regs(BASE + 80 + 16 * e + 4 * i) for i in range(4)] for e in range(2)]      # {c, p, m_lo, m_hi} per symbol
A0, A1, W0, W1, U0, U1, T0, T1, SM0, SM1, Q0, Q1 = (f"v{r}" for r in range(BASE + 112, BASE + 124))
A_T, W_T, U_T, T_T, SM_T, Q_T = (tup(BASE + 112 + 2 * i, 2) for i in range(6))
RR, PSHL, LO, HI, RA, EA = (f"v{r}" for r in range(BASE + 124, BASE + 130))      # (LO, HI): the coder state, a register pair inside the statement
ST_T = tup(BASE + 126, 2)
CK = "v254"
FD = [(tup(BASE + 130 + 4 * k, 2), tup(BASE + 132 + 4 * k, 2), tup(BASE + 130 + 4 * k)) for k in range(4)]
NCH, LIM, FADDR, FOFF = (f"v{r}" for r in range(BASE + 146, BASE + 150))
EW = [regs(BASE + 150 + 8 * e, 8) for e in range(2)]         # (c[t], c[t+1]) of the four symbols of two quads
SD, SAVE = "s[84:85]", "s[86:87]"
CLOBBERS = [f"v{r}" for r in range(BASE, BASE + 167)] + ["s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89"] + \
    (["s90"] if CKPT else []) + ["vcc", "scc", "memory"]
# ONE tile buffer per wave (the second one of gen_encode_loop.py costs 36 KiB of LDS the rows need): the next tile is
# staged between the last read of the current tile (quad 0's symbols, requested in quad 3) and the first read of the
# next one (its quad 7, requested in quad 2) -- one wave's LDS operations execute in order.
ROW = ["%[row0]", "%[row0]"]          # the lane's own row in the tile buffer
TR = ["%[tr0]", "%[tr0]"]             # transposed write address in the tile buffer
SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD"


def step(a, c, p, m0, m1):
    """gen_encode_loop_wide.py's step: state' = q_est 2^P + min(r_est, r_est - p) + c + (fix ? 2^P : 0) as one 64-bit mad"""
    a.i(f"v_lshlrev_b32 {PSHL}, %[shP], {p}", "p << (32 - P)")
    a.i(f"v_lshlrev_b32 {RA}, 8, %[wr]")
    a.i(f"v_cmp_ge_u32 vcc, {HI}, {PSHL}", "emit <=> (state >> (64 - P)) >= p")
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
    a.i(f"v_mul_lo_u32 {RR}, {Q0}, {p}")
    a.i(f"v_sub_u32 {RR}, {A0}, {RR}", "r_est")
    a.i(f"v_sub_u32 {CK}, {RR}, {p}", "r_est - p   (wraps if r_est < p)")
    a.i(f"v_cmp_ge_u32 vcc, {RR}, {p}", "fix <=> q = q_est + 1")
    a.i(f"v_min_u32 {RR}, {RR}, {CK}", "r")
    a.i(f"v_cndmask_b32 {CK}, 0, %[twoPv], vcc", "fix 2^P")
    a.i(f"v_lshlrev_b32 {A1}, %[P], {Q1}", "q_est_hi << P")
    a.i(f"v_add3_u32 {A0}, {RR}, {c}, {CK}", "r + c + fix 2^P")
    a.i(f"v_mad_u64_u32 {ST_T}, {SD}, {Q0}, %[twoP], {A_T}", "state = q_est 2^P + r + c + fix 2^P")


def read_syms(a, g, buf, quad):
    a.ds(f"ds_read_b128 {S_T[g % 4]}, {ROW[buf]} offset:{16 * quad}", f"S{g}")


def stage_a(a, g):
    """symbols of quad g -> row positions; requests the row words"""
    x, y, z, w = S[g % 4]
    a.i(f"v_max3_i32 %[smax], %[smax], {x}, {y}")
    a.i(f"v_max3_i32 %[smax], %[smax], {z}, {w}")
    a.i(f"v_min3_i32 %[smin], %[smin], {x}, {y}")
    a.i(f"v_min3_i32 %[smin], %[smin], {z}, {w}")
    for i, sym in enumerate((w, z, y, x)):       # consumption order: .w first
        a.i(f"v_subrev_u32 {sym}, %[minsym], {sym}", "index i")
        a.i(f"v_med3_i32 {EA}, {sym}, %[alo], %[bhi]", "t = clamp(i, a, b)")
        a.i(f"v_sub_u32 {sym}, {sym}, {EA}", "d = i - t")
        a.i(f"v_lshl_add_u32 {EA}, {EA}, 1, %[rowb]")
        # two aligned 16-bit reads: a misaligned 32-bit one is served 5x slower (scripts/microbench/lds_tput.hip), and
        # with SRAM ECC a d16_hi read does not preserve the other half of its register
        a.ds(f"ds_read_u16 {EW[g % 2][2 * i]}, {EA}", f"R{g}", "c[t]")
        a.ds(f"ds_read_u16 {EW[g % 2][2 * i + 1]}, {EA} offset:2", f"R{g}", "c[t+1]")


def stage_b(a, g):
    """row words of quad g -> (c, p); requests the reciprocals"""
    x, y, z, w = S[g % 4]
    for i, d in enumerate((w, z, y, x)):
        c, p, m0, m1 = E[g % 2][i]
        c0, c1 = EW[g % 2][2 * i], EW[g % 2][2 * i + 1]
        a.i(f"v_add_u32 {c}, {d}, {c0}", "c = c[t] + d")
        a.i(f"v_sub_u32 {p}, {c1}, {c0}", "p = c[t+1] - c[t]")
        a.i(f"v_lshl_add_u32 {EA}, {p}, 3, %[recip]")
        a.ds(f"ds_read_b64 v[{m0[1:]}:{m1[1:]}], {EA}", f"M{g}", "floor(2^64 / p)")


def advance_base(a):
    """s[80:81] -> symbols of the next tile to request; stays on tile 0 once every tile has been requested"""
    a.i("s_cmp_lg_u32 s83, 0")
    a.i("s_cselect_b32 s88, 0x20, 0" if N8 else "s_cselect_b32 s88, 0x80, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_sub_u32 s80, s80, s88")
    a.i("s_subb_u32 s81, s81, 0")
    a.i("s_sub_u32 s83, s83, s89")


def _quad_base(name, k):
    """first register of quad k of symbol set `name` (R[name][k] = "v[b:b+3]")"""
    return int(R[name][k][2:].split(":")[0])


def load_set(a, name):
    for k in range(8):
        if N8:
            a.vmem(f"global_load_dword v{_quad_base(name, k) + 3}, %[goff{k}], s[80:81] nt", f"ld{name}")
        else:
            a.vmem(f"global_load_dwordx4 {R[name][k]}, %[goff{k}], s[80:81] nt", f"ld{name}")
    advance_base(a)


import os


def stage_set(a, name, buf):
    a.wait_vm(f"ld{name}", f"symbols in set {name} have arrived")
    if os.environ.get("GEN_NO_VMWAIT") and len(a.lines) > 100:      # timing experiment only: results are wrong
        a.lines.pop()
    for k in range(8):
        if N8:
            b = _quad_base(name, k)
            for c in range(3):
                a.i(f"v_bfe_i32 v{b + c}, v{b + 3}, {8 * c}, 8", "four int8 symbols -> the quad" if c == 0 else None)
            a.i(f"v_ashrrev_i32 v{b + 3}, 24, v{b + 3}")
        a.ds(f"ds_write_b128 {TR[buf]}, {R[name][k]} offset:{1152 * k}", "tl")


def half(a, h, g0):
    """one tile: register set / tile buffer h (0 = A), global quad indices g0 .. g0+7 stand for quads 7 .. 0"""
    own, other = "AB"[h], "AB"[1 - h]
    a.i(f"; ---- tile in buffer {h} (symbols came from set {own})")
    for j in range(8):
        g, quad = g0 + j, 7 - j
        # the pipeline runs on into the next tile: quads "-1" .. "-3" are quads 7 .. 5 of the other buffer
        if f"S{g + 2}" in a.lds:
            a.wait_lds(f"S{g + 2}", f"quad {quad}: symbols of quad {quad - 2} are back", cap=True)
        far = quad - 3
        read_syms(a, g + 3, h if far >= 0 else 1 - h, far if far >= 0 else far + 8)
        if quad == 3:
            # next tile's symbols -> the tile buffer (behind the read of quad 0); request tile - 3 into the freed registers
            stage_set(a, other, 1 - h)
            load_set(a, other)
        stage_a(a, g + 2)
        if f"R{g + 1}" in a.lds:
            a.wait_lds(f"R{g + 1}", f"row words of quad {quad - 1} are back", cap=True)
        stage_b(a, g + 1)
        if f"M{g}" in a.lds:
            a.wait_lds(f"M{g}", f"reciprocals of quad {quad} are back", cap=True)
        if quad in (7, 6):
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
        for c, p, m0, m1 in E[g % 2]:
            step(a, c, p, m0, m1)
        if quad == 5:
            # word group -> slab
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


def checkpoint(a, label):
    """after the last step of a tile: is this tile the first of a chunk?  then (wr, state) is that chunk's jump point"""
    if not CKPT:
        return
    a.i("s_sub_u32 s90, s90, 1")
    a.i("s_cmp_lg_u32 s90, 0")
    a.i(f"s_cbranch_scc1 {label}f")
    a.i(f"v_lshlrev_b32 {CK}, 2, %[ckidx]")
    a.i(f"global_store_dword {CK}, %[wr], %[ckpos]", "pos: words in the bulk")
    a.i(f"v_lshlrev_b32 {CK}, 3, %[ckidx]")
    a.i(f"global_store_dwordx2 {CK}, {ST_T}, %[ckstate]", "the coder state there")
    a.i("v_add_u32 %[ckidx], -1, %[ckidx]", "the chunk in front of this one is next")
    a.i("s_mov_b32 s90, %[cktiles]")
    a.i("s_waitcnt vmcnt(0)")
    a.i(f"{label}:")


def gen():
    a = Asm()
    a.i(f"v_mov_b32 {W1}, 0")
    a.i(f"v_mov_b32 {LO}, %[lo]")
    a.i(f"v_mov_b32 {HI}, %[hi]")
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST full tile of stream s0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    if CKPT:
        a.i("s_mov_b32 s90, %[cktiles]", "tiles until the next jump point")
    load_set(a, "A")                  # last tile
    load_set(a, "B")                  # the one before
    stage_set(a, "A", 0)
    load_set(a, "A")                  # two before
    read_syms(a, 0, 0, 7)
    read_syms(a, 1, 0, 6)
    read_syms(a, 2, 0, 5)
    a.wait_lds("S0")
    stage_a(a, 0)
    a.wait_lds("S1")
    stage_a(a, 1)
    a.wait_lds("R0")
    stage_b(a, 0)
    a.i("1:")
    first = len(a.events)
    half(a, 0, 0)
    checkpoint(a, 5)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    half(a, 1, 8)
    checkpoint(a, 6)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    # the body is 16 quads: symbol sets (mod 4) and entry sets (mod 2) are back in phase; rename the tags of the
    # quads in flight to what the top of the body expects and verify every wait against the steady state
    ren = {"S16": "S0", "S17": "S1", "S18": "S2", "R16": "R0", "R17": "R1", "M16": "M0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert lds_end == lds_back and vm_end == a.vm, (lds_end, lds_back, vm_end, a.vm)
    a.i("2:")
    a.i(f"v_mov_b32 %[lo], {LO}")
    a.i(f"v_mov_b32 %[hi], {HI}")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


def main():
    a, notes = gen()
    header = ["// GENERATED by scripts/gen_pt_encode_loop.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled per-stream-table ANS encoder" + (", int8 symbol matrix" if N8 else "") + ": see pt_encode_tiles_loop in cst_ans_pt.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr), [flushed] "+v"(flushed), [smin] "+v"(smin), [smax] "+v"(smax)' +
           (', [ckidx] "+v"(ck_index)' if CKPT else ""),
           '    : [row0] "v"(tile_row_addr), [tr0] "v"(tile_tr_addr),',
           '      [lanebase] "v"(ring_lane_addr), [cap] "v"(cap), [slaboff] "v"(slab_off),',
           '      [alo] "v"(sym_lo), [bhi] "v"(sym_hi), [rowb] "v"(row_addr_biased),',
           '      [recip] "s"(recip_addr), [minsym] "s"(min_symbol), [shP] "s"(32u - P), [P] "s"(P), [twoP] "s"(1u << P), [twoPv] "v"(1u << P), [c3f00] "s"(ring_mask), [wbase] "s"(words_base),',
           '      [sbase] "s"(symbols_base), [ntiles] "s"(n_tiles),' +
           (' [ckpos] "s"(ck_pos_base), [ckstate] "s"(ck_state_base), [cktiles] "s"(ck_tiles),' if CKPT else ""),
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
    OUT.write_text(a.render(header, ops))
    print(f"wrote {OUT} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


if __name__ == "__main__":
    main()
