#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_decode_loop_small.inc: the hand-scheduled gfx950 main loop of the (32,64)
ANS decoder with a SMALL LDS footprint (cst_ans_small.hip): for batches of more than one wave per SIMD.

gen_decode_loop.py's statement gives every wave two symbol tiles and reads a 32-KiB (c|p, symbol) table pair: 140 KiB of
LDS per four waves, one wave per SIMD.  A second wave per SIMD doubles the VALU issue rate a SIMD can reach
(scripts/microbench/occupancy.hip: one wave gets at most one instruction per ~4.7 cycles, two waves get two), and its
instructions fill the ~50 cycles per symbol the first one waits for its table lookup.  This statement therefore uses
    * ONE packed table  c | p << 12 | index << 24  (16 KiB; P <= 12, at most 256 symbols): one LDS read per symbol,
    * ONE symbol tile per wave: the tile leaves for HBM at the end of the tile (the sibling wave covers the stall),
    * the 32-slot word ring,
i.e. 17 KiB per wave + 16 KiB per workgroup: eight waves (512 threads) per CU.
Per symbol the serial chain is  entry -> (q - c, p) -> N = (state >> P) * p + (q - c) -> refill? -> state' -> q' -> lookup.

Run:  python scripts/gen_decode_loop_small.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

# GEN_SMALL_N8=1 (round 5): the same loop for INT8 symbol matrices (ans_decode_small_n8_kernel, cst_ans_n8.hip).  A decoded symbol
# leaves its step as a BYTE of the quad's register (the SDWA add that forms  min_symbol + index  writes byte `pos` and preserves
# the others: no instruction more than the int32 form), a quad is one ds_write_b32 into the lane's row of a byte tile -- rows of 128
# symbols + 4 bytes of padding, one line of the matrix -- and the loop body is FOUR tiles: the group of 128 symbols leaves at the
# end of the fourth (eight row blocks: four ds_read_b32 and one 16-byte store each).  16.25 KiB per wave: eight waves per CU.
# (Tried and dropped: TWO tables -- cp[q] = c | p << 16 as in the big decoders and a byte table of symbols at LDS address 0 read with
# ds_read_u8 straight from the quantile: 15.25 VALU per symbol instead of 17.  The D16 byte loads that would have merged two symbols per
# register ZERO the other half on this chip (scripts/microbench/ds_d16.hip: SRAM-ECC), and the kernel ran in the same 0.350 ms with
# the shorter step as with this one: the third LDS read per symbol costs what the two VALU instructions saved.)
N8 = bool(os.environ.get("GEN_SMALL_N8"))
# GEN_SMALL_N8=16: INT16 matrices (cst_decode_loop_small_n16.inc) -- a line is 64 symbols: two tiles per pass, a symbol is a HALF of
# one of the quad's two registers (dst_sel:WORD_n), a quad is two ds_write_b32.
N16 = os.environ.get("GEN_SMALL_N8") == "16"
N8_ROW = 132
SUBTILES = 2 if N16 else 4 if N8 else 1
OUT = Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc" / \
    ("cst_decode_loop_small_n16.inc" if N16 else "cst_decode_loop_small_n8.inc" if N8 else "cst_decode_loop_small.inc")

K_CHUNKS = 3          # window chunks requested per tile (32 symbols * 12 bits = 12 words = 3 chunks)
AHEAD_M1 = 23         # kPtAhead - 1

N0, N1 = "v100", "v101"            # v[100:101] = N
DD = "v102"                        # v[102:103] = [q - c, 0]
PR, T0, T1, LA, CP, WD, RA, R1, Q = (f"v{r}" for r in range(104, 113))
SYM = [f"v{116 + k}" for k in range(8)]
XO = [(f"v[{124 + 4 * k}:{127 + 4 * k}]") for k in range(4)]
PEND = [(f"v[{140 + 4 * k}:{143 + 4 * k}]", [f"v{140 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]
LAND = [f"v{152 + k}" for k in range(K_CHUNKS)]
WANT, TMP, TADDR, TOFF = "v155", "v156", "v157", "v158"
SD, SAVE = "s[84:85]", "s[86:87]"
CLOBBERS = [f"v{r}" for r in range(100, 159)] + [f"s{r}" for r in range(80, 88)] + ["vcc", "scc", "memory"]
SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD"


def wait_if_pending(a, tag, comment=None):
    if tag in a.lds:
        a.wait_lds(tag, comment)


def tail(a, sym_reg, first=False, last=False):
    """everything of a step that is off the chain; issued behind the table read of the NEXT step.
    last: the tile's last step -- the candidate word of the NEXT tile's first refill may be in a chunk that only lands at the end of
    this tile (two tiles in a row that consume their 12 words: data at 12 bits per symbol), so its ring read waits until the landing
    is done (round 5, as in gen_pt_decode_loop.py: it used to be issued here and could read a stale slot --
    tests/test_gpu_max_rate.py::test_small_footprint_kernels_at_the_maximum_rate)."""
    if not first:
        a.i(f"v_subbrev_co_u32 %[rd], {SD}, 0, %[rd], vcc", "rd -= refill")
    a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
    if not last:
        a.ds(f"ds_read_b32 {WD}, {RA}", "w", "candidate word of the next refill")
    if not first:
        a.i(f"v_cndmask_b32 %[hi], {N1}, {N0}, vcc")
    a.i(f"v_min_u32 {R1}, 1, %[rd]")
    a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
    a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")


def step(a, j, sub=0):
    quad, pos = divmod(j, 4)
    sym_reg = SYM[(quad % 2) * 4 + pos]
    if "cp" in a.lds:
        a.wait_lds("cp", f"---- step {j}: entry is back")
    else:                              # (right behind a landing inside the statement: its wait covered the entry)
        a.i(f"; ---- step {j}")
    a.i(f"v_sub_u32 {DD}, {Q}, {CP}")
    a.i(f"v_bfe_u32 {PR}, {CP}, 12, 12", "p")
    a.i(f"v_and_b32 {DD}, 0xfff, {DD}", "q - c")
    a.i(f"v_mad_u64_u32 v[100:101], {SD}, {T0}, {PR}, v[102:103]", "N = (state >> P) * p + (q - c)")
    a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}")
    a.i(f"v_cmp_lt_u32 vcc, {N1}, {R1}", "refill <=> N < 2^32 and words remain")
    if N16:
        a.i(f"v_add_u32_sdwa {SYM[2 * (quad % 2) + (pos >> 1)]}, %[minsym], {CP} dst_sel:WORD_{pos & 1} dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3",
            "decoded symbol -> half of one of the quad's two registers (also: one instruction between vcc's writer and reader)")
    elif N8:
        a.i(f"v_add_u32_sdwa {SYM[quad % 2]}, %[minsym], {CP} dst_sel:BYTE_{pos} dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3",
            "decoded symbol -> byte of the quad's register (also: one instruction between vcc's writer and reader)")
    else:
        a.i(f"v_add_u32_sdwa {sym_reg}, %[minsym], {CP} {SDWA} src0_sel:DWORD src1_sel:BYTE_3", "decoded symbol (also: one instruction between vcc's writer and reader)")
    wait_if_pending(a, "w", "candidate word is back")
    a.i(f"v_cndmask_b32 %[lo], {N0}, {WD}, vcc")
    a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
    a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
    a.ds(f"ds_read_b32 {CP}, {LA}", "cp", "next entry  <- end of the serial chain")
    tail(a, sym_reg, last=(j == 31))
    if pos == 3 and N16:
        a.ds(f"ds_write_b32 %[rowcur], {SYM[2 * (quad % 2)]} offset:{64 * sub + 8 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3} of tile {sub}")
        a.ds(f"ds_write_b32 %[rowcur], {SYM[2 * (quad % 2) + 1]} offset:{64 * sub + 8 * quad + 4}", "tile")
    elif pos == 3 and N8:
        a.ds(f"ds_write_b32 %[rowcur], {SYM[quad % 2]} offset:{32 * sub + 4 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3} of tile {sub}")
    elif pos == 3:
        base = (quad % 2) * 4
        a.ds(f"ds_write_b128 %[rowcur], v[{116 + base}:{119 + base}] offset:{16 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3}")


def gen():
    a = Asm()
    a.i("v_mov_b32 v103, 0")
    a.i("s_mov_b64 s[80:81], %[gbase]", "store base of the current tile, bumped by 128 B per iteration")
    a.i("s_mov_b32 s82, %[ntiles]")
    # first bucket read and the off-chain values of step 0
    a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
    a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
    a.ds(f"ds_read_b32 {CP}, {LA}", "cp")
    tail(a, None, first=True)
    a.i("1:")
    first = len(a.events)
    lds_entry, vm_entry = list(a.lds), list(a.vm)

    for sub in range(SUBTILES):
        if N8:
            a.i(f"; ======== tile {sub} of the group")
        # ---- window: request the chunks this tile's successor may need (landed at the end of this tile) ----
        a.i(f"v_add_u32 {WANT}, %[rd], %[shm1]")
        a.i(f"v_sub_u32_e64 {WANT}, {WANT}, {AHEAD_M1} clamp", "want_lo = max(rd + shift - kPtAhead, 0)")
        for k in range(K_CHUNKS):
            a.i(f"v_cmp_gt_u32 vcc, %[lo_issued], {WANT}", f"chunk slot {k}: needed?")
            a.i(f"v_cndmask_b32_e64 {TMP}, 0, 4, vcc")
            a.i(f"v_sub_u32 %[lo_issued], %[lo_issued], {TMP}")
            a.i(f"v_lshlrev_b32 {TADDR}, 8, %[lo_issued]")
            a.i(f"v_and_or_b32 {TADDR}, {TADDR}, %[cmask], %[lanebase]")
            a.i(f"v_cndmask_b32 {LAND[k]}, %[dump], {TADDR}, vcc", "landing address: ring slot or the dump rows")
            a.i(f"v_lshl_add_u32 {TOFF}, %[lo_issued], 2, %[woff]")
            a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
            a.vmem(f"global_load_dwordx4 {PEND[k][0]}, {TOFF}, %[wbase]", f"chunk{k}")
            a.i(f"s_mov_b64 exec, {SAVE}")

        for j in range(32):
            step(a, j, sub)

        # ---- end of tile: last quad -> tile row, (whole group:) tile -> HBM, chunks -> ring ----
        if sub == SUBTILES - 1:
            a.wait_lds("tile")
            for half in range(2):
                for k in range(4):
                    if N8:
                        base = 124 + 4 * k
                        for c in range(4):
                            a.ds(f"ds_read_b32 v{base + c}, %[trcur] offset:{8 * N8_ROW * (4 * half + k) + 4 * c}", "xo",
                                 f"rows (lane>>3)+{8 * (4 * half + k)}, bytes 16 (lane&7) .. +15" if c == 0 else None)
                    else:
                        a.ds(f"ds_read_b128 {XO[k]}, %[trcur] offset:{1152 * (4 * half + k)}", "xo", f"rows (lane>>3)+{8 * (4 * half + k)}")
                a.wait_lds("xo")
                for k in range(4):
                    a.vmem(f"global_store_dwordx4 %[goff{4 * half + k}], {XO[k]}, s[80:81] \" CST_STORE_MOD \"", "store")
        a.wait_vm(f"chunk{K_CHUNKS - 1}", "the chunk loads are older than this tile's stores")
        for k in range(K_CHUNKS):
            r = PEND[k][1]
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[0]}, {r[1]} offset1:1", "land")
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")
        if sub < SUBTILES - 1:
            a.wait_lds("land", "landed chunks visible to the next tile; the table read of its first step is older")
            a.ds(f"ds_read_b32 {WD}, {RA}", "w", "candidate word of the next tile's first refill: only now, behind the landing")
    a.i("s_add_u32 s80, s80, 0x80")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds("land", "landed chunks visible to the next tile; the table read of its first step is older")
    a.ds(f"ds_read_b32 {WD}, {RA}", "w", "candidate word of the next tile's first refill: only now, behind the landing")
    a.i("s_cbranch_scc1 1b")
    # the back edge must leave the queues as the loop entry found them (modulo completed operations)
    lds_end, vm_end, notes = a.verify_loop(first, list(a.lds), list(a.vm), passes=1)
    assert lds_end == a.lds and vm_end == a.vm, (lds_end, a.lds, vm_end, a.vm)
    assert [t for t in lds_entry if t not in ("cp", "w")] == [] and a.lds == ["w"], (lds_entry, a.lds)
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all("(the candidate word requested for a tile that does not come)")
    return a, notes


def main():
    a, notes = gen()
    header = ["// GENERATED by scripts/gen_decode_loop_small.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (32,64) ANS decoder, small LDS footprint" +
              (", int16 symbol matrices: see decode_groups_loop_small_n8 in cst_ans_n8.hip." if N16 else
               ", int8 symbol matrices: see decode_groups_loop_small_n8 in cst_ans_n8.hip." if N8 else ": see decode_tiles_loop_small in cst_ans_small.hip.")]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued)',
           '    : [lut] "s"(lut_addr), [mask] "s"(mask), [cmask] "s"(ring_mask), [P] "s"(P), [wbase] "s"(words_base), [gbase] "s"(store_base),',
           '      [ntiles] "s"(n_tiles), [minsym] "v"(min_symbol), [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr),',
           '      [dump] "v"(dump_addr), [woff] "v"(words_off), [rowcur] "v"(tile_row_addr), [trcur] "v"(tile_tr_addr),',
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
    OUT.write_text(a.render(header, ops))
    print(f"wrote {OUT} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


if __name__ == "__main__":
    main()
