#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_range_decode_loop{,_ends}.inc: the hand-scheduled gfx950 main loop of the (32,64)
range decoder, P <= 12 (quantile table in LDS) -- one asm statement that decodes full 32-symbol tiles of a wave's 64
streams.  Skeleton as gen_decode_loop.py (ANS): compressed words are requested in 16-byte chunks at the top of a tile
and land in the lane's LDS ring at its end, the decoded symbols go to the lane's row of an LDS tile, and the PREVIOUS
tile is streamed to HBM (transposed reads, 8 rows x 128 B per store) in the shadow of the steps' table lookups.

The step (RangeDecoder::decode_symbol, queue.rs:968-1033) on the state  x = point - lower  (the decoder never needs
`lower` or `point` on their own: x' = x - scale * c, and a renormalisation is x' = x' << 32 | word):
    scale = range >> P;  quantile = x / scale;  (sym, c, p) = table[quantile];  rem = x - scale * c;  nr = scale * p
    renorm <=> nr < 2^32:  range = nr << 32, x = rem << 32 | next word;  else  range = nr, x = rem
The division is an ESTIMATE: x * (1 / scale) in f64 from v_rcp_f64 (measured: only 2^-24.4 accurate,
scripts/microbench/rcp_f64_error.hip) + one Newton step, PLUS 2^-30, truncated: never below the true quotient and
above it only if the true quotient is within 2^-29 below an integer.  (The bias must go this way: an encoder's sealed
point sits at the very bottom of the final interval, so x / scale of a stream's last symbols is routinely an exact
integer -- the first quantile of its bin.)  Instead of correcting the quotient the step checks the SYMBOL it led to:
rem = x - scale * c < nr  <=>  the quantile's bin is the right one (a bin too far gives a negative, i.e. huge, rem; a
quantile of 2^P or more -- invalid data, queue.rs:989-993 -- fails the check too, through the clamped lookup).  A
failed check raises a sticky flag and the caller repeats the wave's streams with the exact C++ step; on valid data
that takes a 2^-29 * n_symbols / 2^P coincidence per symbol.

Two variants: the main one assumes every lane has its next 13 words (it leaves the loop at the top of a tile as soon
as one has not); `_ends` also handles the end of the compressed data (reads past it deliver zeros, queue.rs:1020-1024)
and decodes what the main one left.

Run:  python scripts/gen_range_decode_loop.py   (rewrites the .inc files; they are checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

# GEN_RANGE_SUB=1 (round 5): the loops of the SUB-LANE decoder (range_decode_sub_kernel: k jump points per stream, eight waves
# per workgroup, two per SIMD).  As for the ANS sub-lane decoder (gen_pt_decode_loop.py, GEN_PT_SUB) the second wave is paid for
# by symbol tiles of BYTES: a step leaves its symbol INDEX (< 256) with ds_write_b8 in the lane's 36-byte row; the previous
# tile's pieces are read back four bytes at a time and widened by four SDWA adds of min_symbol in front of their 16-byte store.
# Written to cst_range_decode_loop{,_b16}_sub{,_ends}.inc (stream-major only).
SUB = bool(os.environ.get("GEN_RANGE_SUB"))
SUB_ROW = 36
# GEN_RANGE_N8=1 (round 6, with GEN_RANGE_SUB): the sub-lane decoder that writes an INT8 symbol matrix itself (the reference's Symbol
# is generic: queue.rs:968, quantize.rs:229-255).  The byte tile then holds the symbols themselves (index + min_symbol, low byte:
# P <= 12 through a biased index table, the bucket-entry form through the step's add), the previous tile's four bytes leave as they
# are -- one global_store_dword per quad instead of four SDWA adds and a 16-byte store -- and a tile is 32 bytes of a row.
# Written to cst_range_decode_loop{,_b16}_sub_n8{,_ends}.inc.
N8 = bool(os.environ.get("GEN_RANGE_N8"))
assert not N8 or SUB
CSRC = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc")
OUT = {(False, False): CSRC / "cst_range_decode_loop.inc", (False, True): CSRC / "cst_range_decode_loop_ends.inc",
       (True, False): CSRC / "cst_range_decode_loop_b16.inc", (True, True): CSRC / "cst_range_decode_loop_b16_ends.inc"}
# SYMBOL_MAJOR (the same four files with _sm): symbols[t][stream], the staging of gen_decode_loop.py's SYMBOL_MAJOR (whole lines): quad k of the
# previous tile leaves as streams 32 (k & 1) + 4 (lane & 7) .. + 3 of symbol row (lane >> 3) + 8 (k >> 1); full waves only
# (the store base moves by %[tilestep] per tile).
SYMBOL_MAJOR = False

K_CHUNKS = 3          # window chunks requested per tile (32 symbols * 12 bits = 12 words = 3 chunks)
AHEAD = 24            # kDecAhead


def tup(base, n=2):
    return f"v[{base}:{base + n - 1}]"


X0, X1, RG0, RG1, SC0, SC1 = (f"v{r}" for r in range(120, 126))
PC0, PC1, NR0, NR1, REM0, REM1 = (f"v{r}" for r in range(126, 132))
PC_T, NR_T, REM_T = tup(126), tup(128), tup(130)
FS, FS0, FX, FX0, RC, EST, C2P32, DELTA, NE = (tup(132 + 2 * i) for i in range(9))      # f64 pairs v132 .. v149
Q, LA, C, PR, WD, RA, CP, POS, HI = (f"v{r}" for r in range(150, 159))
SYM = [f"v{160 + k}" for k in range(8)]                                                  # two quads
XT = tup(172, 4)
XB = "v171"           # SUB: four index bytes of the previous tile
PEND = [(tup(176 + 4 * k, 4), [f"v{176 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]
LAND = [f"v{188 + k}" for k in range(K_CHUNKS)]
WANT, TMP, TADDR, TOFF = "v191", "v192", "v193", "v194"
GOFF = [f"v{196 + k}" for k in range(8)]
NXT, IDX, TMPA = "v208", "v209", "v210"                                                # bucket-entry variant:
ESET = [[f"v{204 + k}" for k in range(4)], [f"v{216 + k}" for k in range(4)]]           # entry registers of even / odd steps
ESET_T = [tup(204, 4), tup(216, 4)]
PAIR_T, PAIR0, PAIR1 = tup(212), "v212", "v213"
NEW, NEW_T = [f"v{220 + k}" for k in range(4)], tup(220, 4)                             # a second-level entry on its way in
CLOBBERS = [f"v{r}" for r in range(120, 224)] + [f"s{r}" for r in range(70, 96)] + ["vcc", "scc", "memory"]
SD, SAVE, BAD, CHK, HV, REN = "s[84:85]", "s[86:87]", "s[88:89]", "s[90:91]", "s[92:93]", "s[94:95]"
RET, XSAVE, FLAGGED, V1, V2 = "s[70:71]", "s[72:73]", "s[74:75]", "s[76:77]", "s[78:79]"    # (s96..s101 are flat_scratch / xnack_mask)
B16 = False           # True: 12 < P <= 24, the lookup is one 16-byte bucket entry (DecLut::b16) instead of the table of 2^P quantiles


def quotient_lookup(a, nxt_sym, step=0):
    """from (x, range) to the requests for the next step's table entry and symbol"""
    a.i(f"v_alignbit_b32 {SC0}, {RG1}, {RG0}, %[P]", "scale = range >> P")
    a.i(f"v_lshrrev_b32 {SC1}, %[P], {RG1}")
    a.i(f"v_cvt_f64_u32 {FS}, {SC1}")
    a.i(f"v_cvt_f64_u32 {FS0}, {SC0}")
    a.i(f"v_cvt_f64_u32 {FX}, {X1}")
    a.i(f"v_cvt_f64_u32 {FX0}, {X0}")
    a.i(f"v_fma_f64 {FS}, {FS}, {C2P32}, {FS0}", "scale as f64 (exact below 2^53)")
    a.i(f"v_rcp_f64 {RC}, {FS}", "2^-24 accurate ...")
    a.i(f"v_fma_f64 {FX}, {FX}, {C2P32}, {FX0}", "x as f64  (also the wait state gfx950 needs between a transcendental and its reader)")
    a.i(f"v_fma_f64 {NE}, -{FS}, {RC}, 1.0")
    a.i(f"v_fma_f64 {RC}, {RC}, {NE}, {RC}", "... one Newton step: 2^-48")
    a.i(f"v_fma_f64 {EST}, {FX}, {RC}, {DELTA}", "x / scale + 2^-30: never below the true quotient")
    a.i(f"v_cvt_u32_f64 {Q}, {EST}")
    a.i(f"v_min_u32 {Q}, %[qmax], {Q}", "(a quantile >= 2^P is invalid data: the clamped lookup fails the check)")
    if B16:
        a.i(f"v_lshrrev_b32 {LA}, %[bsh], {Q}")
        a.i(f"v_lshl_add_u32 {LA}, {LA}, 4, %[lut]")
        a.ds(f"ds_read_b128 {ESET_T[step % 2]}, {LA}", "cp", "next bucket entry  <- end of the serial chain")
        return
    a.i(f"v_lshl_add_u32 {LA}, {Q}, 2, %[lut]")
    a.ds(f"ds_read_b32 {CP}, {LA}", "cp", "next c | p << 16  <- end of the serial chain (a random 64-bit read costs ~35 cycles more)")
    a.ds(f"ds_read_b32 {nxt_sym}, {LA} offset:16384", "sym")


def word_request(a):
    a.i(f"v_lshlrev_b32 {RA}, 8, {POS}")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
    a.ds(f"ds_read_b32 {WD}, {RA}", "w", "the word a renormalisation would take")


def gen(ends):
    a = Asm()
    halves = (0, 16) if B16 else (0,)       # where a window of K_CHUNKS chunks is requested (it lands before the next one)

    def window_requests():
        a.i(f"v_add_u32 {WANT}, {AHEAD}, {POS}")
        a.i(f"v_min_u32 {WANT}, {WANT}, %[endr]", "want_hi = min(pos + kDecAhead, end rounded up to a chunk)")
        for k in range(K_CHUNKS):
            a.i(f"v_cmp_lt_u32 vcc, {HI}, {WANT}", f"chunk slot {k}: needed?")
            a.i(f"v_lshlrev_b32 {TADDR}, 8, {HI}")
            a.i(f"v_and_or_b32 {TADDR}, {TADDR}, %[cmask], %[lanebase]")
            a.i(f"v_cndmask_b32 {LAND[k]}, %[dump], {TADDR}, vcc", "landing address: ring slot or the dump rows")
            a.i(f"v_lshl_add_u32 {TOFF}, {HI}, 2, %[woff]")
            a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
            a.vmem(f"global_load_dwordx4 {PEND[k][0]}, {TOFF}, %[wbase]", f"chunk{k}")
            a.i(f"s_mov_b64 exec, {SAVE}")
            a.i(f"v_cndmask_b32_e64 {TMP}, 0, 4, vcc")
            a.i(f"v_add_u32 {HI}, {HI}, {TMP}")

    def window_landing(comment):
        a.wait_lds_all(comment)
        a.wait_vm(f"chunk{K_CHUNKS - 1}", "the chunk loads are older than the stores since")
        for k in range(K_CHUNKS):
            r = PEND[k][1]
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[0]}, {r[1]} offset1:1", "land")
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")

    a.i(f"v_mov_b32 {X0}, %[x0]"); a.i(f"v_mov_b32 {X1}, %[x1]"); a.i(f"v_mov_b32 {RG0}, %[rg0]"); a.i(f"v_mov_b32 {RG1}, %[rg1]")
    a.i(f"v_mov_b32 {POS}, %[pos]"); a.i(f"v_mov_b32 {HI}, %[hi_issued]")
    a.i("v_mov_b32 v144, 0"); a.i("v_mov_b32 v145, 0x41f00000", "2^32")
    a.i("v_mov_b32 v146, 0"); a.i("v_mov_b32 v147, %[dhi]", "+2^-30 (P <= 16) or +2^-22: above the estimate's error of 2^(P - 48.5)")
    # the eight store offsets wait in the lane's row of the CURRENT tile buffer (gen_decode_loop_b16.py: rows of any length,
    # partial waves and the symbol-major mapping are the kernel's business; it writes them in front of EACH of the two
    # statements, whose current buffer differs)
    if SUB:
        for k in range(8):
            a.ds(f"ds_read_b32 {GOFF[k]}, %[rowcur] offset:{4 * k}", "goff")
    else:
        a.ds(f"ds_read_b128 {tup(196, 4)}, %[rowcur]", "goff")
        a.ds(f"ds_read_b128 {tup(200, 4)}, %[rowcur] offset:16", "goff")
    a.wait_lds_all("the store offsets")
    a.i(f"s_mov_b64 {BAD}, 0")
    a.i("s_mov_b64 s[80:81], %[gbase]", "where the PREVIOUS tile goes (first tile of all: onto itself, rewritten one tile later)")
    a.i("v_readfirstlane_b32 s82, %[tiles]", "tiles left")
    a.i("v_readfirstlane_b32 s83, %[ginc]", "0 in front of the very first tile, then 128 B")
    # the first lookup: every tile's last step issues the next tile's
    quotient_lookup(a, SYM[0], 0)
    a.i("1:", None)
    if not ends:
        a.i(f"v_add_u32 {TMP}, {25 if B16 else 13}, {POS}")
        a.i(f"v_cmp_gt_u32 vcc, {TMP}, %[lens]", "a lane that may run out of words inside this tile?")
        a.i("s_cbranch_vccnz 2f", "-> the rest goes to the statement that knows about the end of the data")

    for j in range(32):
        quad, pos = divmod(j, 4)
        nxt = j + 1
        sym_reg = SYM[(nxt // 4 % 2) * 4 + nxt % 4]        # (step 31: symbol 0 of the next tile)
        if j in halves:
            word_request(a)       # (after the previous window's chunks have landed)
            window_requests()
        if "cp" in a.lds:                                  # (right after a landing: its wait covered the entry)
            a.wait_lds("cp", f"---- step {j}: the table entry is back")
        else:
            a.i(f"; ---- step {j}")
        if B16:
            E0, E1, E2, E3 = ESET[j % 2]
            a.i(f"3{j:02d}:", None)
            a.i(f"v_cmp_ge_u32 {V1}, {Q}, {E1}")
            a.i(f"v_cmp_ge_u32 {V2}, {Q}, {E2}")
            a.i(f"v_cmp_ge_u32 vcc, {Q}, {E3}", "beyond the third symbol of the bucket?")
            a.i(f"v_and_b32 {C}, %[cfield], {E0}", "(the cumulative shares its word with the index: 24 + 8 or 22 + 10 bits)")
            a.i(f"v_cndmask_b32_e64 {NXT}, {E1}, {E2}, {V1}")
            a.i(f"v_cndmask_b32_e64 {C}, {C}, {E1}, {V1}")
            a.i(f"v_cndmask_b32_e64 {NXT}, {NXT}, {E3}, {V2}")
            a.i(f"v_cndmask_b32_e64 {C}, {C}, {E2}, {V2}")
            a.i(f"s_cbranch_vccnz 1{j:02d}f", "-> walk the cdf table for those lanes (rare; it also leaves index - 2 in the entry)")
            a.i(f"2{j:02d}:", None)
            a.i(f"v_sub_u32 {PR}, {NXT}, {C}", "p")
        else:
            a.i(f"v_and_b32 {C}, 0xffff, {CP}")
            a.i(f"v_lshrrev_b32 {PR}, 16, {CP}")
        a.i(f"v_mad_u64_u32 {PC_T}, {SD}, {SC0}, {C}, 0", "scale * c")
        a.i(f"v_mad_u64_u32 {NR_T}, {SD}, {SC0}, {PR}, 0", "nr = scale * p")
        a.i(f"v_mad_u32_u24 {PC1}, {SC1}, {C}, {PC1}")
        a.i(f"v_mad_u32_u24 {NR1}, {SC1}, {PR}, {NR1}")
        a.i(f"v_sub_co_u32 {REM0}, vcc, {X0}, {PC0}", "rem = x - scale * c")
        a.i(f"v_subb_co_u32 {REM1}, vcc, {X1}, {PC1}, vcc")
        a.i(f"v_cmp_eq_u32 vcc, 0, {NR1}", "renorm <=> nr < 2^32")
        if ends:
            a.i(f"v_cmp_lt_u32 {HV}, {POS}, %[lens]", "is there a word left?")
        a.i(f"v_cmp_ge_u64 {CHK}, {REM_T}, {NR_T}", "rem >= nr: not this quantile's bin")
        a.wait_lds_all("candidate word (and everything older) is back")
        if ends:
            a.i(f"v_cndmask_b32_e64 {WD}, 0, {WD}, {HV}", "past the end: zeros (queue.rs:1020-1024)")
            a.i(f"s_and_b64 {REN}, vcc, {HV}")
        a.i(f"v_cndmask_b32_e32 {RG1}, {NR1}, {NR0}, vcc", "range = renorm ? nr << 32 : nr")
        a.i(f"v_cndmask_b32_e64 {RG0}, {NR0}, 0, vcc")
        a.i(f"v_cndmask_b32_e32 {X1}, {REM1}, {REM0}, vcc", "x = renorm ? rem << 32 | word : rem")
        a.i(f"v_cndmask_b32_e32 {X0}, {REM0}, {WD}, vcc")
        quotient_lookup(a, sym_reg, j + 1)
        a.i(f"s_or_b64 {BAD}, {BAD}, {CHK}", "(sticky: the caller repeats the streams with the exact step)")
        a.i(f"v_addc_co_u32_e64 {POS}, {SD}, 0, {POS}, {REN if ends else 'vcc'}", "a renormalisation took the word")
        if (j + 1) % 32 not in halves:
            word_request(a)
        if B16:
            a.i(f"v_lshrrev_b32 {IDX}, %[ishift], {E0}", "symbol index = i0 + (q >= e1) + (q >= e2)   (off the chain)")
            a.i(f"v_addc_co_u32_e64 {IDX}, {SD}, 0, {IDX}, {V1}")
            if SUB and not N8:
                a.i(f"v_addc_co_u32_e64 {SYM[(quad % 2) * 4 + pos]}, {SD}, 0, {IDX}, {V2}", "the decoded symbol's index")
            else:
                a.i(f"v_addc_co_u32_e64 {SYM[(quad % 2) * 4 + pos]}, {SD}, %[minsym], {IDX}, {V2}", "the decoded symbol (min_symbol in a VGPR: one scalar operand per instruction)")
        if pos == 1 and SYMBOL_MAJOR:
            for c in range(4):
                a.ds(f"ds_read_b32 v{172 + c}, %[trprev] offset:{(32 * (quad & 1) + c) * 144 + 32 * (quad >> 1)}", "x")
        elif pos == 1 and SUB:
            a.ds(f"ds_read_b32 {XB}, %[trprev] offset:{8 * SUB_ROW * quad}", "x", f"previous tile, rows (lane>>3)+{8 * quad}")
        elif pos == 1:
            a.ds(f"ds_read_b128 {XT}, %[trprev] offset:{1152 * quad}", "x", f"previous tile, rows (lane>>3)+{8 * quad}")
        if pos == 2:
            # XT / XB was read in step pos 1 and is covered by this step's lgkmcnt(0)
            if N8:
                a.vmem(f"global_store_dword {GOFF[quad]}, {XB}, s[80:81] \" CST_STORE_MOD \"", f"store{quad}", "four int8 symbols of a row")
            else:
                if SUB:
                    for b in range(4):
                        a.i(f"v_add_u32_sdwa v{172 + b}, %[minsym], {XB} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{b}",
                            "index -> int32 symbol" if b == 0 else None)
                a.vmem(f"global_store_dwordx4 {GOFF[quad]}, {XT}, s[80:81] \" CST_STORE_MOD \"", f"store{quad}")
        if pos == 3 and SUB:
            base = 160 + (quad % 2) * 4
            for b in range(4):
                a.ds(f"ds_write_b8 %[rowcur], v{base + b} offset:{4 * quad + b}", "tile", f"symbols {4 * quad}..{4 * quad + 3}" if b == 0 else None)
        elif pos == 3:
            base = 160 + (quad % 2) * 4
            a.ds(f"ds_write_b128 %[rowcur], v[{base}:{base + 3}] offset:{16 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3}")
        if (j + 1) % 32 in halves and j != 31:
            window_landing("---- middle of the tile: the first half's chunks land")
            a.wait_lds_all("landed chunks visible")

    window_landing("---- end of tile")
    a.i("v_swap_b32 %[rowcur], %[rowprev]")
    a.i("v_swap_b32 %[trcur], %[trprev]")
    a.i("s_add_u32 s80, s80, s83")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_mov_b32 s83, %[tilestep]" if SYMBOL_MAJOR else "s_movk_i32 s83, 0x20" if N8 else "s_movk_i32 s83, 0x80")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds_all("landed chunks visible to the next tile")
    a.i("s_cbranch_scc1 1b")
    a.i("2:", None)
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    a.i(f"v_mov_b32 %[x0], {X0}"); a.i(f"v_mov_b32 %[x1], {X1}"); a.i(f"v_mov_b32 %[rg0], {RG0}"); a.i(f"v_mov_b32 %[rg1], {RG1}")
    a.i(f"v_mov_b32 %[pos], {POS}"); a.i(f"v_mov_b32 %[hi_issued], {HI}")
    a.i("v_mov_b32 %[tiles], s82"); a.i("v_mov_b32 %[ginc], s83")
    a.i("s_or_b32 s88, s88, s89"); a.i("v_mov_b32 %[bad], s88")
    if B16:
        a.i("s_branch 3f")
        # ---- out of line, per step: second-level entry for the lanes beyond their bucket's third symbol (vcc), the selects again,
        # ---- and the walk over the cdf table for whoever is still beyond (gen_decode_loop_b16.py; one copy of the two routines per
        # ---- entry register set)
        for j in range(32):
            E0, E1, E2, E3 = ESET[j % 2]
            a.i(f"1{j:02d}:", None)
            a.i(f"s_call_b64 {RET}, {4 + j % 2}f", "second-level entries (DecLut, cst_common.hpp) for the lanes in vcc")
            a.i(f"v_cmp_ge_u32 {V1}, {Q}, {E1}")
            a.i(f"v_cmp_ge_u32 {V2}, {Q}, {E2}")
            a.i(f"v_cmp_ge_u32 vcc, {Q}, {E3}")
            a.i(f"v_and_b32 {C}, %[cfield], {E0}")
            a.i(f"v_cndmask_b32_e64 {NXT}, {E1}, {E2}, {V1}")
            a.i(f"v_cndmask_b32_e64 {C}, {C}, {E1}, {V1}")
            a.i(f"v_cndmask_b32_e64 {NXT}, {NXT}, {E3}, {V2}")
            a.i(f"v_cndmask_b32_e64 {C}, {C}, {E2}, {V2}")
            a.i(f"s_cbranch_vccz 2{j:02d}b")
            a.i(f"s_call_b64 {RET}, {8 + j % 2}f", "still beyond the third symbol: walk")
            a.i(f"s_branch 2{j:02d}b")
        for st in range(2):
            E0, E1 = ESET[st][0], ESET[st][1]
            a.i(f"{4 + st}:", None)
            a.i(f"s_mov_b64 {XSAVE}, exec")
            a.i("s_mov_b64 exec, vcc")
            a.i("s_min_u32 s86, %[bsh], 4", "2^min(4, P - 11) parts per bucket (kSubBitsMax) ...")
            a.i("s_sub_u32 s87, %[bsh], s86")
            a.i("s_add_u32 s86, s86, 5", "... in the slot (bucket mod 32: kSubTables) the bucket may own")
            a.i("s_bfm_b32 s86, s86, 0", "(slot | part as a mask)")
            a.i(f"v_lshrrev_b32 {TMPA}, s87, {Q}")
            a.i(f"v_and_b32 {TMPA}, s86, {TMPA}")
            a.i(f"v_lshl_add_u32 {TMPA}, {TMPA}, 4, %[lut]")
            a.i(f"ds_read_b128 {NEW_T}, {TMPA} offset:32768", "(behind the 2048 bucket entries)")
            a.i("s_waitcnt lgkmcnt(0)")
            a.i(f"v_and_b32 {TMPA}, %[cfield], {NEW[0]}")
            a.i(f"v_cmp_le_u32 vcc, {TMPA}, {Q}", "its first cumulative does not lie above q ...")
            a.i(f"v_cmp_gt_u32 {FLAGGED}, {NEW[1]}, {TMPA}", "... and it is an entry (free slots hold zeros)")
            a.i(f"s_and_b64 vcc, vcc, {FLAGGED}")
            a.i("s_and_b64 exec, exec, vcc")
            for k in range(4):
                a.i(f"v_mov_b32 {ESET[st][k]}, {NEW[k]}")
            a.i(f"s_mov_b64 exec, {XSAVE}")
            a.i(f"s_setpc_b64 {RET}")
            a.i(f"{8 + st}:", None)
            a.i(f"s_mov_b64 {XSAVE}, exec")
            a.i(f"s_mov_b64 {FLAGGED}, vcc")
            a.i("s_mov_b64 exec, vcc")
            a.i(f"v_lshrrev_b32 {IDX}, %[ishift], {E0}")
            a.i(f"v_add_u32 {IDX}, 3, {IDX}", "the entry's first three symbols lie below q")
            a.i(f"{6 + st}:", None)
            a.i(f"v_lshl_add_u32 {TMPA}, {IDX}, 2, %[cdf]")
            a.i(f"ds_read_b32 {NXT}, {TMPA} offset:4", "cdf[idx + 1]   (cdf[n] = 2^P lies above every quantile)")
            a.i("s_waitcnt lgkmcnt(0)")
            a.i(f"v_cmp_le_u32 vcc, {NXT}, {Q}")
            a.i(f"v_addc_co_u32_e64 {IDX}, {SD}, 0, {IDX}, vcc")
            a.i("s_and_b64 exec, exec, vcc")
            a.i(f"s_cbranch_execnz {6 + st}b")
            a.i(f"s_mov_b64 exec, {FLAGGED}")
            a.i(f"v_lshl_add_u32 {TMPA}, {IDX}, 2, %[cdf]")
            a.i(f"ds_read2_b32 {PAIR_T}, {TMPA} offset1:1")
            a.i("s_waitcnt lgkmcnt(0)")
            a.i(f"v_mov_b32 {C}, {PAIR0}")
            a.i(f"v_mov_b32 {NXT}, {PAIR1}")
            a.i(f"v_sub_u32 {IDX}, {IDX}, 2", "q >= e1 and q >= e2 hold for these lanes: the step adds 2 again")
            a.i(f"v_lshlrev_b32 {IDX}, %[ishift], {IDX}")
            a.i(f"v_and_b32 {E0}, %[cfield], {E0}")
            a.i(f"v_or_b32 {E0}, {E0}, {IDX}")
            a.i(f"s_mov_b64 exec, {XSAVE}")
            a.i(f"s_setpc_b64 {RET}")
        a.i("3:", None)
    return a


def main():
    global B16, SYMBOL_MAJOR
    for sm, b16 in (((False, False), (False, True)) if SUB else ((False, False), (False, True), (True, False), (True, True))):
        B16, SYMBOL_MAJOR = b16, sm
        for ends in (False, True):
            a = gen(ends)
            header = ["// GENERATED by scripts/gen_range_decode_loop.py -- do not edit by hand (edit the generator and re-run it).",
                      "// Main loop of the hand-scheduled (32,64) range decoder" + (", 12 < P <= 24 (bucket entries)" if b16 else "") +
                      (", end-of-data aware" if ends else "") + ": see cst_range_fast.hip."]
            ops = ['    : [x0] "+v"(x0), [x1] "+v"(x1), [rg0] "+v"(rg0), [rg1] "+v"(rg1), [pos] "+v"(pos), [hi_issued] "+v"(hi_issued),',
                   '      [rowcur] "+v"(row_cur), [rowprev] "+v"(row_prev), [trcur] "+v"(tr_cur), [trprev] "+v"(tr_prev), [tiles] "+v"(tiles), [ginc] "+v"(ginc),',
                   '      [bad] "=v"(bad)',
                   '    : [lut] "s"(lut_addr), [qmax] "s"(qmax), [P] "s"(P), [cmask] "s"(ring_mask), [wbase] "s"(words_base), [dhi] "s"(delta_hi),',
                   '      [gbase] "s"(store_base), [lens] "v"(lens), [endr] "v"(endr), [lanebase] "v"(ring_lane_addr),',
                   '      [dump] "v"(dump_addr), [woff] "v"(words_off)' +
                   (', [bsh] "s"(bucket_shift), [cdf] "s"(cdf_addr), [minsym] "v"(min_symbol), [cfield] "s"(c_field_mask), [ishift] "s"(index_shift)' if b16 else
                    (', [minsym] "v"(min_symbol)' if SUB else '')) +
                   (', [tilestep] "s"(tile_step_bytes)' if sm else ''),
                   "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
            out = OUT[(b16, ends)]
            if SUB:
                tag = "_sub_n8" if N8 else "_sub"
                out = out.with_name(out.name.replace("_ends.inc", tag + "_ends.inc") if ends else out.name.replace(".inc", tag + ".inc"))
                if N8:
                    header[1] = header[1].replace(": see", ", int8 symbol matrix: see")
            if sm:
                out = out.with_name(out.name.replace(".inc", "_sm.inc"))
                header[1] = header[1].replace(": see", ", symbols[t][stream]: see")
            out.write_text(a.render(header, ops))
            print(f"wrote {out} ({a.n_instr()} instructions incl. loop control)")


if __name__ == "__main__":
    main()
