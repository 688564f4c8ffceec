#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_decode_loop_w16.inc: the hand-scheduled gfx950 main loop of the (16,32) ANS decoder
(SmallAnsCoder, stack.rs:153), 8 <= P <= 12, quantile table in LDS.

The step of stack.rs:1084-1097 on a 32-bit state is short -- (state >> P) * p + (q - c) is ONE v_mad_u32_u24, the refill
`state << 16 | word` one v_lshl_or -- 14 VALU per symbol.  What the wave waits for (PMC: 73 of 228 cycles per symbol in
`SQ_WAIT_INST_ANY`) is the vector-memory unit: twice as many 16-byte chunk loads per symbol as the 32-bit presets, every
one of them 64 lane addresses in 64 different cache lines.  The rest is
gen_decode_loop.py's skeleton (word ring, symbol tile streamed out one tile later) with two changes.  32 symbols can
take 24 sixteen-bit words (one per 32-bit slot in HBM, as everywhere in this library), so the window moves every HALF
tile; and half a tile of these short steps (~1600 cycles) is less than an HBM round trip, so a window's chunks land a
whole tile after they were requested (two sets of pending registers: the one requested at step 0 lands after step 31,
the one requested at step 16 after step 15 of the next tile).  That takes a ring of 64 words; it holds them as 16-bit
values ([position][lane] halfwords, 8 KiB per wave as before), written by four ds_write_b16 per chunk.

Run:  python scripts/gen_decode_loop_w16.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

# GEN_W16_PACKED=1 (round 5, CST_FLAG_PACKED_W16): two words per 32-bit slot in HBM, as the reference's Vec<u16> (stack.rs:153).
# A 16-byte chunk then carries EIGHT words: half as many chunk loads per symbol -- the loads were what this decoder waited for
# (SQ_WAIT_INST_ANY 55 - 60 cycles per symbol).  Two chunk slots per half tile instead of three, lo_issued moves by 8, the byte
# offset of a chunk is 2 lo_issued, and a chunk lands with eight ds_write_b16 (low halves with ds_write_b16, high halves with
# ds_write_b16_d16_hi).  Written to cst_decode_loop_w16_pk.inc (stream-major).
PACKED = bool(os.environ.get("GEN_W16_PACKED"))
OUT = Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc" / ("cst_decode_loop_w16_pk.inc" if PACKED else "cst_decode_loop_w16.inc")

CHUNK = 8 if PACKED else 4     # words per 16-byte chunk
K_CHUNKS = 2 if PACKED else 3  # window chunks requested per HALF tile (16 symbols * 12 bits = 12 words)
AHEAD_M1 = 43 if PACKED else 39    # kW16Ahead - 1: 12 words of this half, 24 of the two halves until the chunks have landed, + a chunk


def tup(base, n=4):
    return f"v[{base}:{base + n - 1}]"


# SYMBOL_MAJOR (cst_decode_loop_w16_sm.inc): symbols[t][stream], staged like gen_decode_loop_b16.py's SYMBOL_MAJOR (full waves only)
SYMBOL_MAJOR = False
OUT_SM = OUT.with_name("cst_decode_loop_w16_sm.inc")


def gen():
    a = Asm()
    N, D, PR, T, LA, CP, WD, RA, R1, Q, NS = (f"v{120 + k}" for k in range(11))
    SYM = [f"v{134 + k}" for k in range(8)]      # two quads
    X = tup(144)
    PEND = {"A": [(tup(148 + 4 * k), [f"v{148 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)],
            "B": [(tup(176 + 4 * k), [f"v{176 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]}
    LAND = {"A": [f"v{160 + k}" for k in range(K_CHUNKS)], "B": [f"v{188 + k}" for k in range(K_CHUNKS)]}
    WANT, TMP, TADDR, TOFF = "v163", "v164", "v165", "v166"
    GOFF = [f"v{168 + k}" for k in range(8)]
    clobbers = [f"v{r}" for r in range(120, 192)] + [f"s{r}" for r in range(80, 88)] + ["vcc", "scc", "memory"]
    SD, SAVE = "s[84:85]", "s[86:87]"

    def window_requests(st):
        a.i(f"v_add_u32 {WANT}, %[rd], %[shm1]")
        a.i(f"v_sub_u32_e64 {WANT}, {WANT}, {AHEAD_M1} clamp", "want_lo = max(rd + shift - kW16Ahead, 0)")
        for k in range(K_CHUNKS):
            a.i(f"v_cmp_gt_u32 vcc, %[lo_issued], {WANT}", f"set {st}, chunk slot {k}: needed?")
            a.i(f"v_cndmask_b32_e64 {TMP}, 0, {CHUNK}, vcc")
            a.i(f"v_sub_u32 %[lo_issued], %[lo_issued], {TMP}")
            a.i(f"v_lshlrev_b32 {TADDR}, 7, %[lo_issued]")
            a.i(f"v_and_or_b32 {TADDR}, {TADDR}, %[cmask], %[lanebase]")
            a.i(f"v_cndmask_b32 {LAND[st][k]}, %[dump], {TADDR}, vcc", "landing address: ring position or the dump rows")
            a.i(f"v_lshl_add_u32 {TOFF}, %[lo_issued], {1 if PACKED else 2}, %[woff]")
            a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
            a.vmem(f"global_load_dwordx4 {PEND[st][k][0]}, {TOFF}, %[wbase]", f"chunk{st}{k}")
            a.i(f"s_mov_b64 exec, {SAVE}")

    def window_landing(st, comment):
        a.wait_lds_all(comment)
        a.wait_vm(f"chunk{st}{K_CHUNKS - 1}", f"set {st} was requested a tile ago")
        for k in range(K_CHUNKS):
            for i, r in enumerate(PEND[st][k][1]):
                if PACKED:
                    a.ds(f"ds_write_b16 {LAND[st][k]}, {r} offset:{256 * i}", "land")
                    a.ds(f"ds_write_b16_d16_hi {LAND[st][k]}, {r} offset:{256 * i + 128}", "land")
                else:
                    a.ds(f"ds_write_b16 {LAND[st][k]}, {r} offset:{128 * i}", "land")

    def lookup(sym_reg):
        a.i(f"v_and_b32 {Q}, %[mask], %[st]", "quantile")
        a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
        a.ds(f"ds_read_b32 {CP}, {LA}", "cp", "c | p << 16  <- end of the serial chain")
        a.ds(f"ds_read_b32 {sym_reg}, {LA} offset:16384", "sym")

    def word_request():
        a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 7")
        a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
        a.ds(f"ds_read_u16 {WD}, {RA}", "w")

    def sym_reg(j):
        return SYM[(j // 4 % 2) * 4 + j % 4]

    # the eight store offsets wait in the lane's row of the CURRENT tile buffer (gen_decode_loop_b16.py: rows of any length,
    # partial waves and the symbol-major mapping are the kernel's business)
    a.ds(f"ds_read_b128 {tup(168)}, %[rowcur]", "goff")
    a.ds(f"ds_read_b128 {tup(172)}, %[rowcur] offset:16", "goff")
    a.wait_lds_all("the store offsets")
    a.i("s_mov_b64 s[80:81], %[gbase]", "store base of the PREVIOUS tile, bumped by 128 B per iteration")
    a.i("s_mov_b32 s82, %[ntiles]")
    window_requests("B")      # (the set that lands after step 15)
    a.i("1:", None)

    for j in range(32):
        quad, pos = divmod(j, 4)
        if j in (0, 16):
            # ---- window: request the chunks the next half tile may need, then this half tile's first lookup ----
            window_requests("A" if j == 0 else "B")
            lookup(sym_reg(j))
            word_request()
            a.i(f"v_min_u32 {R1}, 1, %[rd]")
            a.i(f"v_lshrrev_b32 {T}, %[P], %[st]")
        a.wait_lds("cp", f"---- step {j}: the entry is back")
        a.i(f"v_sub_u32_sdwa {D}, {Q}, {CP} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "q - c")
        a.i(f"v_lshrrev_b32 {PR}, 16, {CP}", "p")
        a.i(f"v_mad_u32_u24 {N}, {T}, {PR}, {D}", "N = (state >> P) * p + (q - c)")
        a.i(f"v_lshrrev_b32 {NS}, 16, {N}")
        a.i(f"v_cmp_lt_u32 vcc, {NS}, {R1}", "refill <=> N < 2^16 and words remain")
        a.wait_lds_all("candidate word (and everything older) is back")
        a.i(f"v_lshl_or_b32 {NS}, {N}, 16, {WD}")
        a.i(f"v_cndmask_b32 %[st], {N}, {NS}, vcc")
        last_of_half = j in (15, 31)
        if not last_of_half:
            lookup(sym_reg(j + 1))
        a.i(f"v_subbrev_co_u32 %[rd], {SD}, 0, %[rd], vcc")
        if not last_of_half:
            word_request()
        if pos == 1 and SYMBOL_MAJOR:
            for c in range(4):
                a.ds(f"ds_read_b32 v{144 + c}, %[trprev] offset:{(32 * (quad & 1) + c) * 144 + 32 * (quad >> 1)}", "x")
        elif pos == 1:
            a.ds(f"ds_read_b128 {X}, %[trprev] offset:{1152 * quad}", "x", f"previous tile, rows (lane>>3)+{8 * quad}")
        if not last_of_half:
            a.i(f"v_min_u32 {R1}, 1, %[rd]")
            a.i(f"v_lshrrev_b32 {T}, %[P], %[st]")
        if pos == 2:
            a.vmem(f"global_store_dwordx4 {GOFF[quad]}, {X}, s[80:81] \" CST_STORE_MOD \"", f"store{quad}")
        if pos == 3:
            base = 134 + (quad % 2) * 4
            a.ds(f"ds_write_b128 %[rowcur], v[{base}:{base + 3}] offset:{16 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3}")
        if j == 15:
            window_landing("B", "---- middle of the tile")
            a.wait_lds_all("landed chunks visible")

    window_landing("A", "---- end of tile")
    a.i("v_swap_b32 %[rowcur], %[rowprev]")
    a.i("v_swap_b32 %[trcur], %[trprev]")
    a.i("s_add_u32 s80, s80, %[tilestep]" if SYMBOL_MAJOR else "s_add_u32 s80, s80, 0x80")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds_all("landed chunks visible to the next tile")
    a.i("s_cbranch_scc1 1b")
    window_landing("B", "---- after the last tile: the chunks still under way")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    return a, clobbers


def main():
    global SYMBOL_MAJOR
    for SYMBOL_MAJOR, out in (((False, OUT),) if PACKED else ((False, OUT), (True, OUT_SM))):
        emit(out)
    SYMBOL_MAJOR = False


def emit(out):
    a, clobbers = gen()
    header = ["// GENERATED by scripts/gen_decode_loop_w16.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (16,32) ANS decoder: see cst_ans_w16.hip."]
    ops = ['    : [st] "+v"(st), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued), [rowcur] "+v"(row_cur), [rowprev] "+v"(row_prev),',
           '      [trcur] "+v"(tr_cur), [trprev] "+v"(tr_prev)',
           '    : [lut] "s"(lut_addr), [mask] "s"(mask), [P] "s"(P), [cmask] "s"(ring_mask), [wbase] "s"(words_base), [gbase] "s"(store_base),',
           '      [ntiles] "s"(n_tiles),',
           '      [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr), [dump] "v"(dump_addr), [woff] "v"(words_off)' + (', [tilestep] "s"(tile_step_bytes)' if SYMBOL_MAJOR else ''),
           "    : " + ", ".join(f'"{c}"' for c in clobbers) + ");"]
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions per iteration incl. loop control)")


if __name__ == "__main__":
    main()
