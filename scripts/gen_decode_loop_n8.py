#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_decode_loop_n8.inc: the main loop of the (32,64), P <= 12 ANS decoder for INT8 symbol
matrices (cst_ans_n8.hip, round 5) -- one asm statement that decodes ALL groups of 128 symbols of a wave's 64 streams.

The step, the word window (16-byte chunks requested at the top of a 32-symbol tile, landed in the lane's LDS ring at its end)
and the wait-count book are those of gen_decode_loop.py.  What differs is how the symbols leave:

  * a decoded quad is PACKED into one dword (two v_perm_b32 over the low bytes of the four int32 table values and a v_or: +0.75
    VALU per symbol) and written with ONE ds_write_b32 into the lane's row of a byte tile -- rows of 128 symbols + 4 bytes of
    padding (33 words: the b32 writes of a wave and the b32 reads below are conflict-free);
  * a row is a whole 128-byte line of the int8 matrix, so the loop body is FOUR tiles (a "group"), and the previous group is
    streamed to HBM two row blocks per tile: row block k = rows (lane >> 3) + 8 k, the 16 bytes (lane & 7) of each -- four
    ds_read_b32 and one 16-byte store, eight whole lines per instruction.  Two stores per tile instead of eight, a quarter of
    the int32 decoder's output bytes.
  * the statement starts at the stream's FIRST tile: in front of group 0 there is nothing to store, so the first pass stores the
    (stale) other buffer onto group 0's own lines, which the second pass rewrites (same lanes, same addresses, program order).

Run:  python scripts/gen_decode_loop_n8.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

# GEN_N16=1: the same statement for INT16 matrices (cst_decode_loop_n16.inc).  A 128-byte line is 64 symbols, so a pass is TWO tiles;
# a quad is two dwords (one v_perm_b32 per pair of symbols, two ds_write_b32), and the previous group leaves four row blocks per tile.
N16 = bool(os.environ.get("GEN_N16"))
OUT = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc") / \
    ("cst_decode_loop_n16.inc" if N16 else "cst_decode_loop_n8.inc")

K_CHUNKS = 3          # window chunks requested per tile (32 symbols * 12 bits = 12 words = 3 chunks)
AHEAD_M1 = 23         # kDecAhead - 1
ROW_BYTES = 132       # kN8RowBytes
SUBTILES = 2 if N16 else 4          # tiles per group: one 128-byte line of every row
BLOCK_QUADS = {q: i for i, q in enumerate((1, 3, 5, 7) if N16 else (1, 5))}      # the quads in which a row block of the previous group leaves
NO_STORE = bool(os.environ.get("GEN_NO_STORE"))      # timing experiment only


def gen():
    a = Asm()
    N0, N1 = "v120", "v121"          # v[120:121] = N
    D = "v122"                       # v[122:123] = [q - c, 0]
    PR, T0, T1, LA, CP, WD, RA, R1, Q = "v124", "v125", "v126", "v127", "v128", "v129", "v131", "v132", "v133"
    SYM = [f"v{134 + k}" for k in range(8)] + ["v142"]   # two quads + spare
    X = "v[144:147]"
    PEND = [(f"v[{148 + 4 * k}:{151 + 4 * k}]", [f"v{148 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]
    LAND = [f"v{160 + k}" for k in range(K_CHUNKS)]
    WANT, TMP, TADDR, TOFF = "v163", "v164", "v165", "v166"
    PK0, PK1 = "v167", "v168"
    clobbers = [f"v{r}" for r in range(120, 169)] + ["s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "vcc", "scc", "memory"]
    SD = "s[84:85]"

    a.i("v_mov_b32 v123, 0")
    a.i("s_mov_b64 s[80:81], %[gbase]", "where the PREVIOUS group goes (first pass: group 0's own lines, rewritten by the second)")
    a.i("s_mov_b32 s82, %[ngroups]")
    a.i("s_mov_b32 s83, 0", "the store base stays for one pass, then moves by 128 B per group")
    a.i("1:", None)

    def request_chunk(k):
        a.i(f"v_cmp_gt_u32 vcc, %[lo_issued], {WANT}", f"chunk slot {k}: needed?")
        a.i(f"v_cndmask_b32_e64 {TMP}, 0, 4, vcc")
        a.i(f"v_sub_u32 %[lo_issued], %[lo_issued], {TMP}")
        a.i(f"v_lshlrev_b32 {TADDR}, 8, %[lo_issued]")
        a.i(f"v_and_or_b32 {TADDR}, {TADDR}, %[cmask], %[lanebase]")
        a.i(f"v_cndmask_b32 {LAND[k]}, %[dump], {TADDR}, vcc", "landing address: ring slot or the dump rows")
        a.i(f"v_lshl_add_u32 {TOFF}, %[lo_issued], 2, %[woff]")
        a.i("s_and_saveexec_b64 s[86:87], vcc")
        a.vmem(f"global_load_dwordx4 {PEND[k][0]}, {TOFF}, %[wbase]", f"chunk{k}")
        a.i("s_mov_b64 exec, s[86:87]")

    for sub in range(SUBTILES):
        a.i(f"; ======== tile {sub} of the group")
        # ---- window: request the chunks this tile's successor may need (landed at the end of this tile) ----
        a.i(f"v_add_u32 {WANT}, %[rd], %[shm1]")
        a.i(f"v_sub_u32_e64 {WANT}, {WANT}, {AHEAD_M1} clamp", "want_lo = max(rd + shift - kDecAhead, 0)")
        for k in range(K_CHUNKS):
            request_chunk(k)

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
            # the previous group leaves: 8 / SUBTILES row blocks per tile, in the quads of BLOCK_QUADS
            if pos == 1 and quad in BLOCK_QUADS:
                k = len(BLOCK_QUADS) * sub + BLOCK_QUADS[quad]
                for c in range(4):
                    a.ds(f"ds_read_b32 v{144 + c}, %[trprev] offset:{8 * ROW_BYTES * k + 4 * c}", "x",
                         f"previous group, rows (lane>>3)+{8 * k}, bytes 16 (lane&7) .. +15" if c == 0 else None)
            a.i(f"v_cndmask_b32 %[hi], {N1}, {N0}, vcc")
            a.i(f"v_min_u32 {R1}, 1, %[rd]")
            a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
            a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
            if pos == 2 and quad in BLOCK_QUADS:
                k = len(BLOCK_QUADS) * sub + BLOCK_QUADS[quad]
                # x was issued in the step before and is covered by this step's lgkmcnt(0)
                if NO_STORE:
                    a.vm.append(f"store{k}")
                else:
                    a.vmem(f"global_store_dwordx4 %[goff{k}], {X}, s[80:81] nt", f"store{k}")
            if pos == 3 and N16:
                base = 134 + (quad % 2) * 4
                a.i(f"v_perm_b32 {PK0}, v{base + 1}, v{base}, %[sel01]", f"symbols {4 * quad}..{4 * quad + 3} as int16")
                a.i(f"v_perm_b32 {PK1}, v{base + 3}, v{base + 2}, %[sel01]")
                a.ds(f"ds_write_b32 %[rowcur], {PK0} offset:{64 * sub + 8 * quad}", "tile")
                a.ds(f"ds_write_b32 %[rowcur], {PK1} offset:{64 * sub + 8 * quad + 4}", "tile")
            elif pos == 3:
                base = 134 + (quad % 2) * 4
                a.i(f"v_perm_b32 {PK0}, v{base + 1}, v{base}, %[sel01]", f"symbols {4 * quad}..{4 * quad + 3} as bytes")
                a.i(f"v_perm_b32 {PK1}, v{base + 3}, v{base + 2}, %[sel23]")
                a.i(f"v_or_b32 {PK0}, {PK0}, {PK1}")
                a.ds(f"ds_write_b32 %[rowcur], {PK0} offset:{32 * sub + 4 * quad}", "tile")

        a.wait_lds_all("---- end of tile")
        a.wait_vm(f"chunk{K_CHUNKS - 1}", "the chunk loads are older than this tile's stores")
        for k in range(K_CHUNKS):
            r = PEND[k][1]
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[0]}, {r[1]} offset1:1", "land")
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")
        if sub < SUBTILES - 1:
            a.wait_lds_all("landed chunks visible to the next tile")

    a.i("v_swap_b32 %[rowcur], %[rowprev]")
    a.i("v_swap_b32 %[trcur], %[trprev]")
    a.i("s_add_u32 s80, s80, s83")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_movk_i32 s83, 0x80")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds_all("landed chunks visible to the next tile")
    a.i("s_cbranch_scc1 1b")
    return a, clobbers


def main():
    a, clobbers = gen()
    header = ["// GENERATED by scripts/gen_decode_loop_n8.py -- do not edit by hand (edit the generator and re-run it).",
              f"// Main loop of the hand-scheduled (32,64) ANS decoder for {'int16' if N16 else 'int8'} symbol matrices: see ans_decode_n8_loop in cst_ans_n8.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued), [rowcur] "+v"(row_cur), [rowprev] "+v"(row_prev),',
           '      [trcur] "+v"(tr_cur), [trprev] "+v"(tr_prev)',
           '    : [lut] "s"(lut_addr), [mask] "s"(mask), [P] "s"(P), [cmask] "s"(ring_mask), [wbase] "s"(words_base), [gbase] "s"(store_base),',
           '      [ngroups] "s"(n_groups), [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr), [dump] "v"(dump_addr), [woff] "v"(words_off),',
           ('      [sel01] "s"(0x05040100u),' if N16 else '      [sel01] "s"(0x0c0c0400u), [sel23] "s"(0x04000c0cu),'),
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in clobbers) + ");"]
    OUT.write_text(a.render(header, ops))
    print(f"wrote {OUT} ({a.n_instr()} instructions per iteration incl. loop control)")


if __name__ == "__main__":
    main()
