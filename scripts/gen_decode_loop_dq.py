#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_decode_loop_dq.inc: the main loop of the (32,64), P <= 12 ANS decoder with
LANE-QUAD word loads (cst_ans_dq.hip; the path BASELINE config C2's decoder takes since round 4).

Same decode step and the same tile flow as gen_decode_loop.py; what differs is how the compressed words arrive and how the
previous tile leaves:
  * words: gen_decode_loop.py lets every lane request up to three 16-byte chunks of ITS stream per tile -- three
    instructions of up to 64 requests to 64 different cache lines.  How long the CU's memory pipeline is busy with them
    depends on the slab stride (0.254 ms at a stride of 128 x 64 bytes, 0.366 ms at 103 x 64), and the tile stores queue
    behind them at ISSUE: the decode chain's wave stalls on its store instructions (without the loads OR without the stores
    the kernel runs at 0.250 ms at every stride; never waiting for the loads changes nothing: it is not their latency).
    Here a stream asks for a whole 64-byte group (16 words) when its window needs one, and lanes 4 j .. 4 j + 3 move the four
    chunks of stream 16 p + j's group in pass p = 0 .. 3: at most 16 whole 64-byte segments per instruction.  Positions
    travel with ds_bpermute_b32; the data lands in the OTHER lane's ring column.  A ring of 64 slots per lane holds the
    window (24 words ahead + a group of 16 + the rounding of the prologue).
  * the previous tile: the LDS that second ring half needs is the second tile buffer's.  The finished tile is read back
    (transposed, eight ds_read_b128) into 32 registers at the end of its own iteration and stored from there during the next.

Run:  python scripts/gen_decode_loop_dq.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

OUT = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc") / "cst_decode_loop_dq.inc"
STORE_MOD = '" CST_STORE_MOD "'
AHEAD_M1 = 23         # kDecAhead - 1  (want_lo = max(rd + shift - kDecAhead, 0) = sat_sub(rd + (shift-1), kDecAhead-1))
NO_STORE = bool(os.environ.get("GEN_NO_STORE"))
NO_LOAD = bool(os.environ.get("GEN_NO_LOAD"))
# steps between two passes of group requests.  One per quad (4), each between two stores, measured WORSE than back to back
# (0.293 / 0.345 / 0.271 ms against 0.281 / 0.330 / 0.272 at strides of 97 / 103 / 128 x 64 bytes, gpurun_out/r04_dq_spread.txt)
SPREAD = int(os.environ.get("GEN_DQ_SPREAD", "1"))


def tup(r, n=4):
    return f"v[{r}:{r + n - 1}]"


def gen():
    a = Asm()
    # ---- fixed registers ------------------------------------------------------------------------------
    N0, N1 = "v120", "v121"          # v[120:121] = N
    D = "v122"                       # v[122:123] = [q - c, 0]
    PR, T0, T1, LA, CP, WD, RA, R1, Q = "v124", "v125", "v126", "v127", "v128", "v129", "v131", "v132", "v133"
    SYM = [f"v{134 + k}" for k in range(8)] + ["v142"]   # two quads + spare
    PEND = [(tup(144 + 4 * p), [f"v{144 + 4 * p + j}" for j in range(4)]) for p in range(4)]
    LAND = [f"v{160 + p}" for p in range(4)]
    WANT, TMP, TADDR, TOFF = "v164", "v165", "v166", "v167"
    XT = [tup(168 + 4 * k) for k in range(8)]            # the previous tile, transposed: rows (lane >> 3) + 8 k, symbols 4 (lane & 7) .. + 3
    XQ = "v200"
    XS = [f"v{201 + p}" for p in range(4)]
    WOFFQ = [f"v{205 + p}" for p in range(4)]
    COLQ = [f"v{209 + p}" for p in range(4)]
    C4I, BPA, IDX = "v213", "v214", "v215"
    clobbers = [f"v{r}" for r in range(120, 216)] + ["s80", "s81", "s82", "s84", "s85", "s86", "s87", "vcc", "scc", "memory"]
    SD = "s[84:85]"                  # (s96..s101 hold flat_scratch / xnack_mask on gfx9: never touch them)

    # ---- what depends on the lane only ----
    a.i("v_mov_b32 v123, 0")
    a.i(f"v_mbcnt_lo_u32_b32 {C4I}, -1, 0")
    a.i(f"v_mbcnt_hi_u32_b32 {C4I}, -1, {C4I}", "lane")
    a.i(f"v_and_b32 {BPA}, 0xfc, {C4I}", "4 (lane >> 2): ds_bpermute address of stream (lane >> 2)")
    a.i(f"v_lshlrev_b32 {TMP}, 2, {C4I}")
    a.i(f"v_sub_u32 {TMP}, %[lanebase], {TMP}", "the wave's ring")
    a.i(f"v_add_u32 {COLQ[0]}, {TMP}, {BPA}", "ring column of stream (lane >> 2)")
    for p in range(1, 4):
        a.i(f"v_add_u32 {COLQ[p]}, {64 * p}, {COLQ[0]}", f"... of stream {16 * p} + (lane >> 2)")
    for p in range(4):
        a.ds(f"ds_bpermute_b32 {WOFFQ[p]}, {BPA}, %[woff] offset:{64 * p}", "bp0", f"byte offset of the words of stream {16 * p} + (lane >> 2)")
    a.i(f"v_and_b32 {C4I}, 3, {C4I}")
    a.i(f"v_lshlrev_b32 {C4I}, 2, {C4I}", "4 (lane & 3): first of this lane's four words in a group")
    a.i("s_mov_b64 s[80:81], %[gbase]", "store base of the PREVIOUS tile, bumped by 128 B per iteration")
    a.i("s_mov_b32 s82, %[ntiles]")
    for k in range(8):
        a.ds(f"ds_read_b128 {XT[k]}, %[tr] offset:{1152 * k}", "x", f"tile 0, rows (lane>>3)+{8 * k}")
    a.wait_lds_all()
    a.i("1:", None)

    # ---- window: does this stream's window need its next 64-byte group?  (one per tile keeps it full: a tile takes <= 12 words) ----
    a.i(f"v_add_u32 {WANT}, %[rd], %[shm1]")
    a.i(f"v_sub_u32_e64 {WANT}, {WANT}, {AHEAD_M1} clamp", "want_lo = max(rd + shift - kDecAhead, 0)")
    a.i(f"v_cmp_gt_u32 vcc, %[lo_issued], {WANT}")
    a.i(f"v_cndmask_b32_e64 {TMP}, 0, 16, vcc")
    a.i(f"v_sub_u32 %[lo_issued], %[lo_issued], {TMP}")
    a.i(f"v_lshl_or_b32 {XQ}, {TMP}, 27, %[lo_issued]", "first word of the group | (requested) << 31")
    for p in range(4):
        a.ds(f"ds_bpermute_b32 {XS[p]}, {BPA}, {XQ} offset:{64 * p}", "bp", f"... of stream {16 * p} + (lane >> 2)")

    # ---- first lookup of the tile ----
    a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
    a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
    a.ds(f"ds_read_b32 {CP}, {LA}", "cp")
    a.ds(f"ds_read_b32 {SYM[0]}, {LA} offset:16384", "sym0")
    a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
    a.ds(f"ds_read_b32 {WD}, {RA}", "w")
    a.i(f"v_min_u32 {R1}, 1, %[rd]")
    a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
    a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")

    for j in range(32):
        quad, pos = divmod(j, 4)
        # symbol j+1 goes to: quad registers alternate between SYM[0:4] and SYM[4:8]; symbol 32 to the spare
        nxt = j + 1
        sym_reg = SYM[8] if nxt == 32 else SYM[(nxt // 4 % 2) * 4 + nxt % 4]
        a.wait_lds("cp", f"---- step {j}: entry is back")
        a.i(f"v_sub_u32_sdwa {D}, {Q}, {CP} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "q - c")
        a.i(f"v_lshrrev_b32 {PR}, 16, {CP}", "p")
        a.i(f"v_mad_u64_u32 v[120:121], {SD}, {T0}, {PR}, v[122:123]", "N = (state >> P) * p + (q - c)")
        a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}")
        a.i(f"v_cmp_lt_u32 vcc, {N1}, {R1}", "refill <=> N < 2^32 and words remain")
        a.wait_lds_all("candidate word (and everything older) is back")
        a.i(f"v_cndmask_b32 %[lo], {N0}, {WD}, vcc")
        a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
        a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
        a.ds(f"ds_read_b32 {CP}, {LA}", "cp", "next entry  <- end of the serial chain")
        a.i(f"v_subbrev_co_u32 %[rd], {SD}, 0, %[rd], vcc")
        a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
        a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
        a.ds(f"ds_read_b32 {WD}, {RA}", "w")
        a.ds(f"ds_read_b32 {sym_reg}, {LA} offset:16384", f"sym{nxt}")
        a.i(f"v_cndmask_b32 %[hi], {N1}, {N0}, vcc")
        a.i(f"v_min_u32 {R1}, 1, %[rd]")
        a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
        a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
        if j % SPREAD == 0 and j // SPREAD < 4:
            # pass p of the group requests (the positions came back with step 0's lgkmcnt(0)), one per quad: a load instruction
            # keeps the CU's memory pipeline busy for a while, and the next store should not be the one that waits for it
            p = j // SPREAD
            a.i(f"v_and_or_b32 {IDX}, {XS[p]}, %[c7f], {C4I}", "this lane's first word (bit 31 leaves)")
            a.i(f"v_cmp_gt_i32 vcc, 0, {XS[p]}", "requested?")
            a.i(f"v_lshlrev_b32 {TADDR}, 8, {IDX}")
            a.i(f"v_and_or_b32 {TADDR}, {TADDR}, %[cmask], {COLQ[p]}")
            a.i(f"v_cndmask_b32 {LAND[p]}, %[dump], {TADDR}, vcc", "landing address: the STREAM's ring column or this lane's dump rows")
            a.i(f"v_lshl_add_u32 {TOFF}, {IDX}, 2, {WOFFQ[p]}")
            a.i("s_and_saveexec_b64 s[86:87], vcc")
            if NO_LOAD:
                a.vm.append(f"chunk{p}")
            else:
                a.vmem(f"global_load_dwordx4 {PEND[p][0]}, {TOFF}, %[wbase]", f"chunk{p}")
            a.i("s_mov_b64 exec, s[86:87]")
        if (pos == 2 and (quad >= 1 or SPREAD > 1)) or (j == 4 and SPREAD == 1):
            # (back-to-back requests fill quad 0's slots: its part of the previous tile leaves at the top of quad 1)
            for k in ((0,) if j == 4 and SPREAD == 1 else (quad,)):
                if NO_STORE:
                    a.vm.append(f"store{k}")
                else:
                    a.vmem(f"global_store_dwordx4 %[goff{k}], {XT[k]}, s[80:81] {STORE_MOD}".rstrip(), f"store{k}")
        if pos == 3:
            base = (quad % 2) * 4
            a.ds(f"ds_write_b128 %[rowcur], v[{134 + base}:{137 + base}] offset:{16 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3}")
    a.wait_lds_all("---- end of tile: the tile is complete in LDS")
    for k in range(8):
        if os.environ.get("GEN_DQ_NOXT"):       # timing experiment (results wrong): what the read-back burst costs
            break
        a.ds(f"ds_read_b128 {XT[k]}, %[tr] offset:{1152 * k}", "x", f"this tile, rows (lane>>3)+{8 * k} (leaves during the next one)")
    a.wait_vm("chunk3", "the group loads are older than this tile's stores")
    for p in range(4):
        r = PEND[p][1]
        a.ds(f"ds_write2st64_b32 {LAND[p]}, {r[0]}, {r[1]} offset1:1", "land")
        a.ds(f"ds_write2st64_b32 {LAND[p]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")
    a.i("s_add_u32 s80, s80, 0x80")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds_all("landed groups visible to the next tile; the tile is in registers")
    a.i("s_cbranch_scc1 1b")
    return a, clobbers


def main():
    a, clobbers = gen()
    header = ["// GENERATED by scripts/gen_decode_loop_dq.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the (32,64), P <= 12 ANS decoder with lane-quad word loads: see ans_decode_dq_loop in cst_ans_dq.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued)',
           '    : [rowcur] "v"(row_addr), [tr] "v"(tr_addr), [lut] "s"(lut_addr), [mask] "s"(mask), [P] "s"(P), [cmask] "s"(ring_mask), [c7f] "s"(0x7fffffffu),',
           '      [wbase] "s"(words_base), [gbase] "s"(store_base), [ntiles] "s"(n_tiles), [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr),',
           '      [dump] "v"(dump_addr), [woff] "v"(words_off),',
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in clobbers) + ");"]
    OUT.write_text(a.render(header, ops))
    print(f"wrote {OUT} ({a.n_instr()} instructions per iteration incl. loop control)")


if __name__ == "__main__":
    main()
