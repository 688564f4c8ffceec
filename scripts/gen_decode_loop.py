#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_decode_loop.inc: the hand-scheduled gfx950 main loop of the (32,64) ANS
decoder (one asm statement that decodes tiles 1..n-1 of a wave's 64 streams).

Why a generator: the statement is ~800 instructions of straight-line code per loop iteration whose s_waitcnt
operands depend on how many LDS / vector-memory operations were issued after the one being waited for.  This
script keeps that book (one wave's LDS operations complete in order, and so do its vector-memory operations) and
emits the counts, so that moving an instruction cannot silently break a wait.

Run:  python scripts/gen_decode_loop.py   (rewrites the .inc; the .inc is checked in)
"""
import os
from pathlib import Path

OUT = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc") / "cst_decode_loop.inc"
OUT_SM = OUT.with_name("cst_decode_loop_sm.inc")
# SYMBOL_MAJOR (cst_decode_loop_sm.inc): symbols[t][stream].  Only the way the previous tile leaves differs: quad k reads
# tile[32 (k & 1) + 4 (lane & 7) + c][(lane >> 3) + 8 (k >> 1)], c = 0..3 (four ds_read_b32, two lanes per bank) and stores the
# 16 bytes at symbol row (lane >> 3) + 8 (k >> 1), streams 32 (k & 1) + 4 (lane & 7) .. + 3: every store instruction writes eight
# WHOLE 128-byte lines (round 2's mapping wrote sixteen half lines: the halves of a line left in different instructions, which
# cost the P = 24 decoder a third of its time); the store base moves by 32 * n_streams * 4 bytes per tile (an operand).
SYMBOL_MAJOR = False
# The tile stores carry the modifier CST_STORE_MOD, a string macro the including function defines: "nt" for rows that start
# on cache-line boundaries, "" otherwise.  With `nt` a 128-byte row segment that straddles two cache lines (rows whose length
# is not a multiple of 32 symbols) goes to HBM as two partial lines per tile: 0.86 ms instead of 0.29 at 65 536 x 4100; plain
# stores let L2 merge the halves (0.54 ms) but cost the aligned case its streaming behaviour (0.37 instead of 0.29 ms).
STORE_MOD = '" CST_STORE_MOD "'

K_CHUNKS = 3          # window chunks requested per tile (32 symbols * 12 bits = 12 words = 3 chunks)
AHEAD_M1 = 23         # kDecAhead - 1  (want_lo = max(rd + shift - kDecAhead, 0) = sat_sub(rd + (shift-1), kDecAhead-1))


import os
import sys
sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402


import os
NO_STORE = bool(os.environ.get("GEN_NO_STORE"))
NO_LOAD = bool(os.environ.get("GEN_NO_LOAD"))
# experiment: the previous tile's eight stores in steps 1 .. 8, right behind the chunk requests, instead of one per quad: the
# end-of-tile wait for the chunks also waits for every OLDER store (one in-order counter), and the last store of a tile is
# then 1.75 tiles old instead of 1.06
EARLY_STORES = bool(os.environ.get("GEN_EARLY_STORES"))
# experiment: the three chunk requests at the top of quads 0, 1, 2 instead of back to back at the top of the tile
SPREAD_LOADS = bool(os.environ.get("GEN_SPREAD_LOADS"))
# K_PRED (experiment, GEN_KPRED=1; the kernel must stage the third table: -DCST_DEC_KPRED): the refill decision without the
# 64-bit product on the way to the next lookup.
#   N = (state >> P) * p + (q - c) < 2^32   <=>   state >> P < 2^32  and  low32(state >> P) <= K[q],  K[q] = floor((2^32 - 1 - (q - c)) / p)
# K[q] sits next to cp[q] in LDS, and the LOW word of N -- all the next lookup needs -- is two 24-bit multiplies: from one table
# entry to the next lookup the DEPENDENT chain is shift, mad24, shift-add, select, and, address (6 full-rate instructions)
# instead of shift, 64-bit mad (quarter rate), mad24, compare, select, and, address; the exact 64-bit product still runs, for
# the state's high word, in the shadow of the next lookup.  Bit-exact (271 batch tests) and SLOWER: 0.294 - 0.300 ms against
# 0.256 (gpurun_out/r04_kpred_ab.txt).  A lone wave issues one instruction per ~4.3 cycles whether it depends on the previous
# one or not, so what counts between the entry's arrival and the next lookup's issue is the NUMBER of instructions there
# (9 before, 12 + a second table read with this form), not their dependent depth; the step is one LDS latency + that section.
K_PRED = bool(os.environ.get("GEN_KPRED"))
# EARLY_ELIG (experiment): N = (state >> P) p + (q - c) < 2^32  <=>  state >> P < 2^32 and the
# 64-bit mad's own high word is 0 -- so the test "state >> P < 2^32 and words remain" moves in front of the lookup's return (it only
# needs the state), the 24-bit mad for the high word of N behind the next lookup's issue, and the section between a table entry's
# arrival and the next lookup is 8 instructions instead of 9.
# ... bit-exact, and 4.7 % SLOWER (0.265 against 0.253 ms, gpurun_out/r04_elig.txt): the two instructions it adds behind the lookup's issue
# cost more than the one it removes in front of it -- the shadow of the lookup is full.  Kept as an experiment (GEN_ELIG=1).
EARLY_ELIG = not K_PRED and bool(os.environ.get("GEN_ELIG"))
# LAZY_SYM (experiment, GEN_LAZY_SYM=1): the wait in front of the refill select covers the candidate WORD only -- lgkmcnt(1) or more instead of
# lgkmcnt(0): the symbol read issued behind it (a second random read: ~13 LDS cycles per wave, scripts/microbench/lds_tput.hip) and the
# transposing tile reads are off the chain; their consumers (the quad's tile write, the tile stores) wait for them by name
# ... bit-exact and NO faster (0.2537 / 0.2571 against 0.2536 / 0.2531 ms, alternating runs on one box): the five instructions between the entry's
# arrival and that wait already cover the symbol read -- the LDS pipe's 70 % load does not lengthen the chain.  Kept as an experiment.
LAZY_SYM = bool(os.environ.get("GEN_LAZY_SYM"))
ABL_R1 = bool(os.environ.get("GEN_ABL_R1"))      # timing experiment: no min(rd, 1) per step (right only while no stream runs out of words)


def gen():
    a = Asm()
    # ---- fixed registers ------------------------------------------------------------------------------
    N0, N1 = "v120", "v121"          # v[120:121] = N
    D = "v122"                       # v[122:123] = [q - c, 0]
    PR, T0, T1, LA, CP, WD, RA, R1, Q = "v124", "v125", "v126", "v127", "v128", "v129", "v131", "v132", "v133"
    SYM = [f"v{134 + k}" for k in range(8)] + ["v142"]   # two quads + spare
    X = "v[144:147]"                 # (register tuples must start at an even register on gfx950)
    PEND = [(f"v[{148 + 4 * k}:{151 + 4 * k}]", [f"v{148 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]
    LAND = [f"v{160 + k}" for k in range(K_CHUNKS)]
    WANT, TMP, TADDR, TOFF = "v163", "v164", "v165", "v166"
    X2 = "v[172:175]"
    KQ, BH, M1, M2 = "v130", "v143", "v167", "v168"
    ELIG = "s[88:89]"
    clobbers = [f"v{r}" for r in range(120, 176 if EARLY_STORES else (169 if K_PRED else 167))] + ["s80", "s81", "s82", "s84", "s85", "s86", "s87"] + \
               (["s88", "s89"] if K_PRED or EARLY_ELIG else []) + ["vcc", "scc", "memory"]
    SD = "s[84:85]"                  # (s96..s101 hold flat_scratch / xnack_mask on gfx9: never touch them)

    a.i("v_mov_b32 v123, 0")
    if ABL_R1:
        a.i(f"v_mov_b32 {R1}, 1")
    a.i("s_mov_b64 s[80:81], %[gbase]", "store base of the PREVIOUS tile, bumped by 128 B per iteration")
    a.i("s_mov_b32 s82, %[ntiles]")
    a.i("1:", None)

    # ---- window: request the chunks this tile's successor may need (landed at the end of this iteration) ----
    a.i(f"v_add_u32 {WANT}, %[rd], %[shm1]")
    a.i(f"v_sub_u32_e64 {WANT}, {WANT}, {AHEAD_M1} clamp", "want_lo = max(rd + shift - kDecAhead, 0)")
    def request_chunk(k):
        a.i(f"v_cmp_gt_u32 vcc, %[lo_issued], {WANT}", f"chunk slot {k}: needed?")
        a.i(f"v_cndmask_b32_e64 {TMP}, 0, 4, vcc")
        a.i(f"v_sub_u32 %[lo_issued], %[lo_issued], {TMP}")
        a.i(f"v_lshlrev_b32 {TADDR}, 8, %[lo_issued]")
        a.i(f"v_and_or_b32 {TADDR}, {TADDR}, %[cmask], %[lanebase]")
        a.i(f"v_cndmask_b32 {LAND[k]}, %[dump], {TADDR}, vcc", "landing address: ring slot or the dump rows")
        a.i(f"v_lshl_add_u32 {TOFF}, %[lo_issued], 2, %[woff]")
        a.i("s_and_saveexec_b64 s[86:87], vcc")
        if NO_LOAD:
            a.vm.append(f"chunk{k}")
        else:
            a.vmem(f"global_load_dwordx4 {PEND[k][0]}, {TOFF}, %[wbase] {os.environ.get('GEN_LOAD_MOD', '').replace('+', ' ')}".rstrip(), f"chunk{k}")
        a.i("s_mov_b64 exec, s[86:87]")

    if not SPREAD_LOADS:
        for k in range(K_CHUNKS):
            request_chunk(k)

    # ---- first lookup of the tile ----
    a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
    a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
    a.ds(f"ds_read_b32 {CP}, {LA}", "cp")
    if K_PRED:
        a.ds(f"ds_read_b32 {KQ}, {LA} offset:32768", "k")
    a.ds(f"ds_read_b32 {SYM[0]}, {LA} offset:16384", "sym0")
    a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
    a.ds(f"ds_read_b32 {WD}, {RA}", "w")
    if EARLY_ELIG:
        a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
        a.i(f"v_cmp_eq_u32_e64 {ELIG}, 0, {T1}")
    if not ABL_R1:
        a.i(f"v_min_u32 {R1}, 1, %[rd]")
    a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
    if not EARLY_ELIG:
        a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
    if K_PRED:
        a.i(f"v_lshrrev_b32 {BH}, 24, {T0}")
        a.i(f"v_cmp_lt_u32 {ELIG}, {T1}, {R1}", "a refill is possible <=> state >> P < 2^32 and words remain")
    if EARLY_ELIG:
        a.i(f"v_cndmask_b32_e64 {R1}, 0, {R1}, {ELIG}", "1 if a refill is possible (state >> P < 2^32 and words remain), else 0")

    for j in range(32):
        quad, pos = divmod(j, 4)
        # symbol j+1 goes to: quad registers alternate between SYM[0:4] and SYM[4:8]; symbol 32 to the spare
        nxt = j + 1
        sym_reg = SYM[8] if nxt == 32 else SYM[(nxt // 4 % 2) * 4 + nxt % 4]
        if K_PRED:
            a.wait_lds("k", f"---- step {j}: entry and K are back")
            a.i(f"v_lshrrev_b32 {PR}, 16, {CP}", "p")
            a.i(f"v_sub_u32_sdwa {D}, {Q}, {CP} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "q - c")
            a.i(f"v_cmp_le_u32 vcc, {T0}, {KQ}", "low32(state >> P) * p + (q - c) < 2^32")
            a.i(f"v_mad_u32_u24 {M1}, {T0}, {PR}, {D}", "low word of N = (state >> P) * p + (q - c): the factor's low 24 bits ...")
            a.i(f"v_mul_u32_u24 {M2}, {BH}, {PR}", "... and its bits 24..31")
            a.i(f"v_lshl_add_u32 {M1}, {M2}, 24, {M1}")
            a.wait_lds_all("candidate word (and everything older) is back")
            if os.environ.get("GEN_K_SALU"):      # (measured slower: the VALU -> SALU -> VALU round trip of the mask sits on the chain)
                a.i(f"s_and_b64 vcc, vcc, {ELIG}", "refill <=> N < 2^32 and words remain")
                a.i(f"v_cndmask_b32 %[lo], {M1}, {WD}, vcc")
            else:
                a.i(f"v_cndmask_b32_e64 {M2}, {M1}, {WD}, {ELIG}", "the word, if a refill is possible at all")
                a.i(f"v_cndmask_b32 %[lo], {M1}, {M2}, vcc")
            a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
            a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
            a.ds(f"ds_read_b32 {CP}, {LA}", "cp", "next entry  <- end of the serial chain")
            a.ds(f"ds_read_b32 {KQ}, {LA} offset:32768", "k")
            # in the shadow of that lookup: the state's high word from the exact product, the next step's operands
            if not os.environ.get("GEN_K_SALU"):
                a.i(f"s_and_b64 vcc, vcc, {ELIG}", "refill <=> N < 2^32 and words remain")
            a.i(f"v_mad_u64_u32 v[120:121], {SD}, {T0}, {PR}, v[122:123]", "N = (state >> P) * p + (q - c)")
            a.i(f"v_subbrev_co_u32 %[rd], {SD}, 0, %[rd], vcc")
            a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
            a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
            a.ds(f"ds_read_b32 {WD}, {RA}", "w")
            a.ds(f"ds_read_b32 {sym_reg}, {LA} offset:16384", f"sym{nxt}")
        else:
            a.wait_lds("cp", f"---- step {j}: entry is back")
            a.i(f"v_sub_u32_sdwa {D}, {Q}, {CP} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "q - c")
            a.i(f"v_lshrrev_b32 {PR}, 16, {CP}", "p")
            a.i(f"v_mad_u64_u32 v[120:121], {SD}, {T0}, {PR}, v[122:123]", "N = (state >> P) * p + (q - c)" + ("   (without the high word of state >> P)" if EARLY_ELIG else ""))
            if not EARLY_ELIG:
                a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}")
            a.i(f"v_cmp_lt_u32 vcc, {N1}, {R1}", "refill <=> N < 2^32 and words remain")
            if LAZY_SYM:
                a.wait_lds("w", "candidate word is back (the symbol read behind it may still fly)")
            else:
                a.wait_lds_all("candidate word (and everything older) is back")
            a.i(f"v_cndmask_b32 %[lo], {N0}, {WD}, vcc")
            a.i(f"v_and_b32 {Q}, %[mask], %[lo]")
            a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
            a.ds(f"ds_read_b32 {CP}, {LA}", "cp", "next entry  <- end of the serial chain")
            if EARLY_ELIG:
                a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}", "the high word of N, in the shadow of the lookup")
            a.i(f"v_subbrev_co_u32 %[rd], {SD}, 0, %[rd], vcc")
            a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
            a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
            a.ds(f"ds_read_b32 {WD}, {RA}", "w")
            a.ds(f"ds_read_b32 {sym_reg}, {LA} offset:16384", f"sym{nxt}")
        if EARLY_STORES:
            assert not SYMBOL_MAJOR
            if j < 8:
                a.ds(f"ds_read_b128 {X2 if j % 2 else X}, %[trprev] offset:{1152 * j}", "x", f"previous tile, rows (lane>>3)+{8 * j}")
        elif pos == 1 and SYMBOL_MAJOR:
            for c in range(4):
                a.ds(f"ds_read_b32 v{144 + c}, %[trprev] offset:{(32 * (quad & 1) + c) * 144 + 32 * (quad >> 1)}", "x",
                     f"previous tile, stream 32*{quad & 1}+4*(lane&7)+{c}, symbol (lane>>3)+{8 * (quad >> 1)}")
        elif pos == 1:
            a.ds(f"ds_read_b128 {X}, %[trprev] offset:{1152 * quad}", "x", f"previous tile, rows (lane>>3)+{8 * quad}")
        if K_PRED:
            a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}")
            a.i(f"v_min_u32 {R1}, 1, %[rd]")
            a.i(f"v_cndmask_b32 %[hi], {N1}, {M1}, vcc")
            a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
            a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
            a.i(f"v_lshrrev_b32 {BH}, 24, {T0}")
            a.i(f"v_cmp_lt_u32 {ELIG}, {T1}, {R1}")
        else:
            a.i(f"v_cndmask_b32 %[hi], {N1}, {N0}, vcc")
            if EARLY_ELIG:
                a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
                a.i(f"v_cmp_eq_u32_e64 {ELIG}, 0, {T1}")
                a.i(f"v_min_u32 {R1}, 1, %[rd]")
                a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
                a.i(f"v_cndmask_b32_e64 {R1}, 0, {R1}, {ELIG}", "1 if a refill is possible, else 0")
            else:
                if not ABL_R1:
                    a.i(f"v_min_u32 {R1}, 1, %[rd]")
                a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
                a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
        if SPREAD_LOADS and pos == 0 and quad < K_CHUNKS:
            request_chunk(quad)      # (after the step's last reader of vcc)
        if EARLY_STORES:
            if 1 <= j <= 8:     # x was issued in the step before and is covered by this step's lgkmcnt(0)
                k = j - 1
                a.vmem(f"global_store_dwordx4 %[goff{k}], {X2 if k % 2 else X}, s[80:81] {STORE_MOD}".rstrip(), f"store{k}")
        elif pos == 2:
            # x was issued in step pos 1 and is covered by this step's lgkmcnt(0)
            if LAZY_SYM and "x" in a.lds:
                a.wait_lds("x", "the previous tile's piece is back")
            if NO_STORE:
                a.vm.append(f"store{quad}")
            else:
                a.vmem(f"global_store_dwordx4 %[goff{quad}], {X}, s[80:81] {os.environ['GEN_STORE_MOD'].replace('+', ' ') if 'GEN_STORE_MOD' in os.environ else STORE_MOD}".rstrip(), f"store{quad}")
        if pos == 3:
            base = (quad % 2) * 4
            if LAZY_SYM and f"sym{j}" in a.lds:
                a.wait_lds(f"sym{j}", "the quad's last symbol is back")
            a.ds(f"ds_write_b128 %[rowcur], v[{134 + base}:{137 + base}] offset:{16 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3}")

    a.wait_lds_all("---- end of tile")
    a.wait_vm(f"chunk{K_CHUNKS - 1}", "the chunk loads are older than this tile's stores")
    for k in range(K_CHUNKS):
        r = PEND[k][1]
        a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[0]}, {r[1]} offset1:1", "land")
        a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")
    a.i("v_swap_b32 %[rowcur], %[rowprev]")
    a.i("v_swap_b32 %[trcur], %[trprev]")
    a.i("s_add_u32 s80, s80, %[tilestep]" if SYMBOL_MAJOR else "s_add_u32 s80, s80, 0x80")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds_all("landed chunks visible to the next tile")
    a.i("s_cbranch_scc1 1b")
    return a, clobbers


def main():
    global SYMBOL_MAJOR
    for SYMBOL_MAJOR, out in ((False, OUT), (True, OUT_SM)):
        if SYMBOL_MAJOR and EARLY_STORES:
            continue
        emit(out)


def emit(out):
    a, clobbers = gen()
    header = ["// GENERATED by scripts/gen_decode_loop.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (32,64) ANS decoder: see ans_decode_tiles_loop in cst_ans_kernels.hpp."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued), [rowcur] "+v"(row_cur), [rowprev] "+v"(row_prev),',
           '      [trcur] "+v"(tr_cur), [trprev] "+v"(tr_prev)',
           '    : [lut] "s"(lut_addr), [mask] "s"(mask), [P] "s"(P), [cmask] "s"(ring_mask), [wbase] "s"(words_base), [gbase] "s"(store_base),',
           '      [ntiles] "s"(n_tiles), [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr), [dump] "v"(dump_addr), [woff] "v"(words_off),'
           + (' [tilestep] "s"(tile_step_bytes),' if SYMBOL_MAJOR else ''),
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in clobbers) + ");"]
    if SYMBOL_MAJOR:
        header[1] = "// Main loop of the hand-scheduled (32,64) ANS decoder, symbols[t][stream]: see ans_decode_tiles_loop_sm in cst_ans_asm.hpp."
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions per iteration incl. loop control)")


if __name__ == "__main__":
    main()
