#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_encode_loop_w16.inc: the hand-scheduled gfx950 main loop of the (16,32) ANS encoder
(SmallAnsCoder, stack.rs:153), 8 <= P <= 12 -- gen_encode_loop.py's pipeline (it is imported: symbol tiles, entry fetch, word
ring, register sets) around the 32-bit step, with TWO word groups leaving per tile (32 symbols can emit 24 sixteen-bit
words, each in a 32-bit slot).

The step (stack.rs:1035-1045 on a 32-bit state) on a table entry  e0 = c | (c + 2^P - p) << 16,  e1 = p,
e2 = floor(2^32 / p),  e3 = p << (32 - P):
    emit  <=>  (state >> (32 - P)) >= p  <=>  state >= e3;   A = emit ? state >> 16 : state   (< p * 2^20)
    q_est = mulhi(A, e2) in {q - 1, q}  (< 2^20: one v_mul_u32_u24 gives q_est * p);  r_est = A - q_est * p;  fix <=> r_est >= p
    state' = (q << P) + c + r = (q_est << P) + r_est + (fix ? c + 2^P - p : c)
14 VALU + the ring write.  The ring is zeroed once and written with ds_write_b16, so that the upper halves of its slots
stay zero without masking the word.

Run:  python scripts/gen_encode_loop_w16.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import gen_encode_loop as G  # noqa: E402
from asmgen import Asm  # noqa: E402

# GEN_W16_PACKED=1 (round 5, CST_FLAG_PACKED_W16): the slab holds the words as the reference does -- a Vec<u16>, two words per
# 32-bit slot (stack.rs:153) -- so a 64-byte group is THIRTY-TWO words and at most one leaves per tile (32 symbols emit at most
# 24).  The ring keeps one word per 32-bit slot (upper halves zero: ds_write_b16 into a zeroed ring); the flush reads eight
# 4-word chunks per group (quads 7, 6 and 4, 3 as before), packs pairs with v_lshl_or_b32 (16 VALU per group: 0.4 per symbol)
# and stores 32 bytes at quad 5 and 32 at quad 2.  A chunk read late is never overwritten: word flushed + 4 c is written again
# when wr reaches flushed + 64 + 4 c, and wr - flushed stays below 55 + 12.  Written to cst_encode_loop_w16_pk.inc.
PACKED = bool(os.environ.get("GEN_W16_PACKED"))
# GEN_W16_CK=1 (round 6, with GEN_W16_PACKED): the same loop with JUMP POINTS -- the reference's Pos side information (stack.rs:1107-1139:
# words in the bulk, state), which depends on neither the word type nor how the words are stored: after the last step of every
# ck_tiles-th tile the lanes store (wr, state) at element ck_index of the two jump arrays and step to the chunk in front (as
# gen_pt_encode_loop.py's GEN_PT_CK does).  Written to cst_encode_loop_w16_pk_ck.inc.
CKPT = bool(os.environ.get("GEN_W16_CK"))
assert not CKPT or PACKED
OUT = G.CSRC / ("cst_encode_loop_w16_pk_ck.inc" if CKPT else "cst_encode_loop_w16_pk.inc" if PACKED else "cst_encode_loop_w16.inc")
CKST, CKZ, CKA, CKCNT = "v218", "v219", "v220", "s92"          # v[218:219] = [state, 0]: the jump table holds 64-bit states
A_, SH, QE, R_, T_, CKS = (f"v{r}" for r in range(212, 218))
RA, NCH, LIM, FADDR, FOFF, FD, SAVE = G.RA, G.NCH, G.LIM, G.FADDR, G.FOFF, G.FD, G.SAVE


def step(a, e0, e1, e2, e3):
    a.i(f"v_cmp_ge_u32 vcc, %[st], {e3}", "emit <=> (state >> (32 - P)) >= p")
    a.i(f"v_lshlrev_b32 {RA}, 8, %[wr]")
    a.i(f"v_lshrrev_b32 {SH}, 16, %[st]")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[c3f00], %[lanebase]")
    a.i(f"v_cndmask_b32 {A_}, %[st], {SH}, vcc")
    a.ds(f"ds_write_b16 {RA}, %[st]", "W", "candidate word, always written (the slot's upper half stays zero)")
    a.i(f"v_addc_co_u32 %[wr], vcc, 0, %[wr], vcc")
    a.i(f"v_mul_hi_u32 {QE}, {A_}, {e2}", "q_est in {q - 1, q}")
    a.i(f"v_mul_u32_u24 {R_}, {QE}, {e1}")
    a.i(f"v_sub_u32 {R_}, {A_}, {R_}", "r_est")
    a.i(f"v_cmp_ge_u32 vcc, {R_}, {e1}", "fix <=> q = q_est + 1")
    a.i(f"v_lshl_add_u32 {T_}, {QE}, %[P], {R_}", "(q_est << P) + r_est")
    a.i(f"v_cndmask_b32_sdwa {CKS}, {e0}, {e0}, vcc {G.SDWA} src0_sel:WORD_0 src1_sel:WORD_1", "c, or c + 2^P - p")
    a.i(f"v_add_u32 %[st], {T_}, {CKS}")


def packed_reads(a, ks, decide):
    """chunks `ks` (4 words each, two per quad) of the 32-word group at `flushed`"""
    if decide:
        a.i(f"v_sub_u32 {NCH}, %[wr], %[flushed]")
        a.i(f"v_lshrrev_b32 {NCH}, 5, {NCH}")
        a.i(f"v_min_u32 {NCH}, 1, {NCH}", "whole 32-word groups to move now: 0 or 1")
    for k in ks:
        a.i(f"v_add_lshl_u32 {FADDR}, %[flushed], {4 * k}, 8")
        a.i(f"v_and_or_b32 {FADDR}, {FADDR}, %[c3f00], %[lanebase]")
        a.ds(f"ds_read2st64_b32 {FD[k % 4][0]}, {FADDR} offset1:1", "fl")
        a.ds(f"ds_read2st64_b32 {FD[k % 4][1]}, {FADDR} offset0:2 offset1:3", "fl")


def packed_store(a, second):
    """16 words in FD (four chunks of four) -> eight dwords -> two 16-byte stores at byte 0 / 32 of the group"""
    if not second:
        a.i(f"v_add_u32 {LIM}, 32, %[flushed]")
        a.i(f"v_lshl_add_u32 {FOFF}, %[flushed], 1, %[slaboff]", "(two bytes per word)")
        a.i(f"v_cmp_le_u32 vcc, {LIM}, %[cap]", "group inside the slab (cap % 32 == 0 on this path)")
        a.i(f"v_cmp_ne_u32 {SAVE}, 0, {NCH}")
        a.i(f"s_and_b64 {MASK}, vcc, {SAVE}")
    a.wait_lds("fl", cap=True)
    for pair in range(2):
        b = 230 + 8 * pair                   # chunks 2 pair and 2 pair + 1 sit in v[b : b + 7]
        for j in range(4):
            a.i(f"v_lshl_or_b32 v{b + j}, v{b + 2 * j + 1}, 16, v{b + 2 * j}", "two words per slot" if pair == 0 and j == 0 else None)
    a.i(f"s_mov_b64 {SAVE}, exec")
    a.i(f"s_mov_b64 exec, {MASK}")
    for pair in range(2):
        a.vmem(f"global_store_dwordx4 {FOFF}, v[{230 + 8 * pair}:{233 + 8 * pair}], %[wbase] offset:{32 * int(second) + 16 * pair}", "st")
    a.i(f"s_mov_b64 exec, {SAVE}")
    if second:
        a.i(f"v_lshl_add_u32 %[flushed], {NCH}, 5, %[flushed]")


MASK = "s[90:91]"


def group_reads(a, ks, decide):
    if decide:
        # decide NOW whether the group is complete: words written after these reads must not count
        a.i(f"v_sub_u32 {NCH}, %[wr], %[flushed]")
        a.i(f"v_lshrrev_b32 {NCH}, 4, {NCH}")
        a.i(f"v_min_u32 {NCH}, 1, {NCH}", "whole 16-word groups to move now: 0 or 1")
    for k in ks:
        a.i(f"v_add_lshl_u32 {FADDR}, %[flushed], {4 * k}, 8")
        a.i(f"v_and_or_b32 {FADDR}, {FADDR}, %[c3f00], %[lanebase]")
        a.ds(f"ds_read2st64_b32 {FD[k][0]}, {FADDR} offset1:1", "fl")
        a.ds(f"ds_read2st64_b32 {FD[k][1]}, {FADDR} offset0:2 offset1:3", "fl")


def group_store(a):
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


def half(a, h, g0):
    """one tile: register set / tile buffer h (0 = A), global quad indices g0 .. g0+7 stand for quads 7 .. 0"""
    own, other = "AB"[h], "AB"[1 - h]
    a.i(f"; ---- tile in buffer {h} (symbols came from set {own})")
    for j in range(8):
        g, quad = g0 + j, 7 - j
        if f"S{g + 1}" in a.lds:
            a.wait_lds(f"S{g + 1}", f"quad {quad}: symbols of the next quad are back", cap=True)
        far = quad - 2
        G.read_syms(a, g + 2, h if far >= 0 else 1 - h, far if far >= 0 else far + 8)
        G.fetch_entries(a, g + 1)
        if f"E{g}" in a.lds:
            a.wait_lds(f"E{g}", f"entries of quad {quad} are back", cap=True)
        if quad in (7, 6, 4, 3) and PACKED:
            packed_reads(a, {7: (0, 1), 6: (2, 3), 4: (4, 5), 3: (6, 7)}[quad], quad == 7)
        elif quad in (7, 6, 4, 3):
            # two 64-byte word groups may leave per tile: decided at quads 7 and 4 (12 and 20 symbols apart: at most 9 and
            # 15 words are produced in between)
            group_reads(a, (0, 1) if quad in (7, 4) else (2, 3), quad in (7, 4))
        G.fold_minmax(a, g)
        for e in G.E[g % 2]:
            step(a, *e)
        if quad == 5:
            packed_store(a, False) if PACKED else group_store(a)
        if quad == 2:
            # second word group -> slab; next tile's symbols -> the other tile buffer; request tile - 3 into the freed
            # registers (every store of a tile is issued before its loads)
            packed_store(a, True) if PACKED else group_store(a)
            a.wait_lds(f"E{g + 1}", "(early: keeps the eight tile writes below within lgkmcnt's range of 15)")
            G.stage_set(a, other, 1 - h)
            G.load_set(a, other)


def checkpoint(a, label):
    """after the last step of a tile: is this tile the first of a chunk?  then (wr, state) is that chunk's jump point"""
    if not CKPT:
        return
    a.i(f"s_sub_u32 {CKCNT}, {CKCNT}, 1")
    a.i(f"s_cmp_lg_u32 {CKCNT}, 0")
    a.i(f"s_cbranch_scc1 {label}f")
    a.i(f"v_lshlrev_b32 {CKA}, 2, %[ckidx]")
    a.i(f"global_store_dword {CKA}, %[wr], %[ckpos]", "pos: 16-bit words in the bulk")
    a.i(f"v_mov_b32 {CKST}, %[st]")
    a.i(f"v_lshlrev_b32 {CKA}, 3, %[ckidx]")
    a.i(f"global_store_dwordx2 {CKA}, v[218:219], %[ckstate]", "the coder state there")
    a.i("v_add_u32 %[ckidx], -1, %[ckidx]", "the chunk in front of this one is next")
    a.i(f"s_mov_b32 {CKCNT}, %[cktiles]")
    a.i("s_waitcnt vmcnt(0)")
    a.i(f"{label}:")


def gen():
    G.SINGLE = False
    a = Asm()
    if CKPT:
        a.i(f"v_mov_b32 {CKZ}, 0")
        a.i(f"s_mov_b32 {CKCNT}, %[cktiles]", "tiles until the next jump point")
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST full tile of stream s0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    G.load_set(a, "A")                  # last tile
    G.load_set(a, "B")                  # the one before
    G.stage_set(a, "A", 0)
    G.load_set(a, "A")                  # two before
    G.read_syms(a, 0, 0, 7)
    G.read_syms(a, 1, 0, 6)
    a.wait_lds("S0")
    G.fetch_entries(a, 0)
    a.i("1:")
    first = len(a.events)
    half(a, 0, 0)
    checkpoint(a, "5")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    half(a, 1, 8)
    checkpoint(a, "6")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    ren = {"S16": "S0", "S17": "S1", "E16": "E0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert lds_end == lds_back, (lds_end, lds_back)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, notes


OUT_SM = OUT.with_name("cst_encode_loop_w16_sm.inc")


def main():
    emit(OUT, False)
    if not PACKED:
        emit(OUT_SM, True)          # symbols[t][stream]: gen_encode_loop.py's SYMBOL_MAJOR staging


def emit(out, symbol_major):
    G.SYMBOL_MAJOR = symbol_major
    a, notes = gen()
    G.SYMBOL_MAJOR = False
    header = ["// GENERATED by scripts/gen_encode_loop_w16.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (16,32) ANS encoder: see cst_ans_w16.hip."]
    ops = ['    : [st] "+v"(st), [wr] "+v"(wr), [flushed] "+v"(flushed), [smin] "+v"(smin), [smax] "+v"(smax)' + (', [ckidx] "+v"(ck_index)' if CKPT else ""),
           '    : [row0] "v"(tile_row_addr[0]), [row1] "v"(tile_row_addr[1]), [tr0] "v"(tile_tr_addr[0]), [tr1] "v"(tile_tr_addr[1]),',
           '      [lanebase] "v"(ring_lane_addr), [cap] "v"(cap), [slaboff] "v"(slab_off),',
           '      [tbl] "s"(table_addr_biased), [P] "s"(P), [c3f00] "s"(0x3f00u), [wbase] "s"(words_base),',
           '      [sbase] "s"(symbols_base), [ntiles] "s"(n_tiles),' + (' [tilestep] "s"(tile_step_bytes),' if symbol_major else '')
           + (' [ckpos] "s"(ck_pos_base), [ckstate] "s"(ck_state_base), [cktiles] "s"(ck_tiles),' if CKPT else ''),
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in G.CLOBBERS) + ");"]
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


if __name__ == "__main__":
    main()
