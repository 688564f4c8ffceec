#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_decode_loop_b16.inc: the hand-scheduled gfx950 main loop of the (32,64) ANS
decoder for 12 < P <= 24 (DefaultAnsCoder's PRECISION = 24), alphabets of at most 256 symbols.

There is no table of 2^P quantiles at this precision.  The lookup (lookup_contiguous.rs:564-605 / contiguous.rs:628-665)
is ONE 16-byte LDS read of a bucket entry (cst_common.hpp, DecLut::b16), bucket = quantile >> (P - 11):
    { cdf[i0] | i0 << 24, cdf[i0+1], cdf[i0+2], cdf[i0+3] }      i0 = the symbol that holds the bucket's first quantile
    k = (q >= e1) + (q >= e2);  c = (c0, e1, e2)[k];  next = (e1, e2, e3)[k];  p = next - c;  symbol index = i0 + k
and a quantile >= e3 (more than three symbols begin in its bucket below it: the far tails of a distribution, 2^-11 of
the probability mass per such bucket) takes a wave-uniform walk over the cdf table (an out-of-line block at the end of
the statement, entered through one s_cbranch per step).

Everything else is gen_decode_loop.py's skeleton (the step of stack.rs:1084-1097 on 32-bit halves, the word ring, the
symbol tile streamed out one tile later), except that a tile of up to 24 words is handled as two half tiles of at most
12: the window requests / landings of gen_decode_loop.py run at steps 0 and 16, so the ring stays at 32 slots and the
workgroup's LDS image at 142 KiB.

Run:  python scripts/gen_decode_loop_b16.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc" / "cst_decode_loop_b16.inc"
# GEN_B16_NARROW=1 / 2 (round 5: cst_decode_loop_b16_n8.inc / _n16.inc, ans_decode_b16_narrow_kernel): int8 / int16 symbol matrices.  A
# decoded quad is packed (two v_perm_b32 + v_or for int8, two v_perm_b32 for int16) into the lane's row of ONE byte tile -- rows of 128
# bytes + 4: a line of the matrix -- and the group of 128 / 64 symbols leaves when its last tile is done, in a block of eight row
# blocks (four ds_read_b32 + one 16-byte store each) that a scalar branch skips on the other tiles.  Nobody covers that block's ~300
# cycles (one wave per SIMD), but they come once per 128 / 64 symbols of ~214 cycles each; in exchange the statement's body stays ONE
# tile.  The block's memory operations are not in the wait-count book: operations the book does not know can only make a wait
# LONGER (they are younger than what it waits for, or older and retired with it), never shorter.
NARROW = int(os.environ.get("GEN_B16_NARROW", "0"))
# GEN_B16_SMALL=1 with GEN_B16_NARROW=1 / 2 / 4 (cst_decode_loop_b16_s8.inc / _s16.inc / _s32.inc, ans_decode_b16_narrow_kernel<BYTES, true>): the
# SMALL-FOOTPRINT form for two waves per SIMD (more than 256 streams per CU: a C5-sized shard, or the virtual streams of a batch decoded through
# jump points).  Rings of 16 slots -- a window every QUARTER tile: 8 symbols take at most 6 words, two chunks -- and the one tile per wave of the
# narrow form, for int32 too (GEN_B16_NARROW=4: rows of 36 words, the tile leaves in the block behind its landing): eight waves in 140 KiB.
SMALL = int(os.environ.get("GEN_B16_SMALL", "0"))
assert not SMALL or NARROW in (1, 2, 4), "the small-footprint form: int8, int16 or int32 through the one-tile block"
assert NARROW != 4 or SMALL
N8_ROW = 144 if NARROW == 4 else 132
TILES_PER_GROUP = {0: 1, 1: 4, 2: 2, 4: 1}[NARROW]
if SMALL:
    OUT = OUT.with_name({1: "cst_decode_loop_b16_s8.inc", 2: "cst_decode_loop_b16_s16.inc", 4: "cst_decode_loop_b16_s32.inc"}[NARROW])
elif NARROW:
    OUT = OUT.with_name("cst_decode_loop_b16_n8.inc" if NARROW == 1 else "cst_decode_loop_b16_n16.inc")

# bucket entries: 2^11 as the model tabulates them, 2^12 in the one-wave narrow kernels (LDS to spare: B16NarrowGeo::kTableBits)
BUCKET_BITS = 12 if (NARROW in (1, 2) and not SMALL) else 11
WINDOWS = (0, 8, 16, 24) if SMALL else (0, 16)     # steps in front of which the next part of the tile's words is requested
K_CHUNKS = 2 if SMALL else 3     # window chunks requested per part (16 symbols * 24 bits = 12 words = 3 chunks; 8 symbols: 6 words = 2 chunks)
AHEAD_M1 = 11 if SMALL else 23   # kDecAhead - 1


def tup(base, n=4):
    return f"v[{base}:{base + n - 1}]"


# SYMBOL_MAJOR (cst_decode_loop_b16_sm.inc): symbols[t][stream], the staging of gen_decode_loop.py's SYMBOL_MAJOR: quad k of the
# previous tile leaves as streams 32 (k & 1) + 4 (lane & 7) .. + 3 of symbol row (lane >> 3) + 8 (k >> 1); full waves only
# (the base moves by %[tilestep] per tile).
SYMBOL_MAJOR = False
OUT_SM = OUT.with_name("cst_decode_loop_b16_sm.inc")


def gen():
    a = Asm()
    N0, N1, D = "v120", "v121", "v122"          # v[120:121] = N, v[122:123] = [q - c, 0]
    PR, T0, T1, LA, WD, RA, R1, Q = "v124", "v125", "v126", "v127", "v129", "v131", "v132", "v133"
    SYM = [f"v{134 + k}" for k in range(8)]      # two quads
    X = tup(144)
    ESET = [[f"v{148 + k}" for k in range(4)], [f"v{188 + k}" for k in range(4)]]     # entry registers of even / odd steps
    ESET_T = [tup(148), tup(188)]
    C, NXT, IDX, TMPA = "v152", "v153", "v154", "v155"
    PAIR0, PAIR1, PAIR_T = "v156", "v157", tup(156, 2)
    PEND = [(tup(160 + 4 * k), [f"v{160 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]
    LAND = [f"v{172 + k}" for k in range(K_CHUNKS)]
    WANT, TMP, TADDR, TOFF = "v175", "v176", "v177", "v178"
    GOFF = [f"v{180 + k}" for k in range(8)]
    NEW = [f"v{192 + k}" for k in range(4)]                                            # a second-level entry on its way in
    NEW_T = tup(192)
    clobbers = [f"v{r}" for r in range(120, 196)] + [f"s{r}" for r in range(70, 92)] + ["vcc", "scc", "memory"]
    SD, SAVE, V1, V2 = "s[84:85]", "s[86:87]", "s[88:89]", "s[90:91]"
    RET, XSAVE, FLAGGED = "s[70:71]", "s[72:73]", "s[74:75]"
    SBITS, SSH = "s76", "s77"

    def window_requests():
        a.i(f"v_add_u32 {WANT}, %[rd], %[shm1]")
        a.i(f"v_sub_u32_e64 {WANT}, {WANT}, {AHEAD_M1} clamp", "want_lo = max(rd + shift - kDecAhead, 0)")
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

    def window_landing(comment):
        a.wait_lds_all(comment)
        a.wait_vm(f"chunk{K_CHUNKS - 1}", "(this half tile's stores are younger than the chunk loads)")
        for k in range(K_CHUNKS):
            r = PEND[k][1]
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[0]}, {r[1]} offset1:1", "land")
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")

    def lookup(step):
        a.i(f"v_bfe_u32 {LA}, %[lo], %[bsh], {BUCKET_BITS}", f"bucket = bits [P - {BUCKET_BITS}, P) of the state")
        a.i(f"v_lshl_add_u32 {LA}, {LA}, 4, %[lut]")
        a.ds(f"ds_read_b128 {ESET_T[step % 2]}, {LA}", "e", "bucket entry  <- end of the serial chain")
        a.i(f"v_and_b32 {Q}, %[mask], %[lo]", "quantile")

    def word_request():
        a.i(f"v_add_lshl_u32 {RA}, %[rd], %[shm1], 8")
        a.i(f"v_and_or_b32 {RA}, {RA}, %[cmask], %[lanebase]")
        a.ds(f"ds_read_b32 {WD}, {RA}", "w")

    a.i("v_mov_b32 v123, 0")
    a.i(f"s_min_u32 {SBITS}, %[bsh], 4", "second-level tables: 2^min(4, P - 11) parts per bucket (kSubBitsMax) ...")
    a.i(f"s_sub_u32 {SSH}, %[bsh], {SBITS}")
    a.i(f"s_add_u32 {SBITS}, {SBITS}, 5", "... in the slot (bucket mod 32: kSubTables) the bucket may own")
    a.i(f"s_bfm_b32 {SBITS}, {SBITS}, 0", "(slot | part as a mask)")
    # the eight store offsets of a tile's pieces wait in the lane's row of the CURRENT tile buffer (written by the kernel, read
    # here before the first step writes symbols over them): rows of any length take per-row offsets (row_skew,
    # cst_ans_kernels.hpp), partial waves repeat their last row, symbol-major batches have their own mapping -- all the
    # kernel's business, and an asm statement has room for 30 operands
    if NARROW in (1, 2):
        for k in range(8):
            a.ds(f"ds_read_b32 {GOFF[k]}, %[rowcur] offset:{4 * k}", "goff")      # (rows of 33 words: not 16-byte aligned)
        a.i("s_mov_b32 s83, 0", "tiles of the current group done")
    else:
        a.ds(f"ds_read_b128 {tup(180)}, %[rowcur]", "goff")
        a.ds(f"ds_read_b128 {tup(184)}, %[rowcur] offset:16", "goff")
    a.wait_lds_all("the store offsets")
    a.i("s_mov_b64 s[80:81], %[gbase]", "store base of the PREVIOUS tile, bumped by 128 B per iteration")
    a.i("s_mov_b32 s82, %[ntiles]")
    a.i("1:", None)

    for j in range(32):
        quad, pos = divmod(j, 4)
        if j in WINDOWS:
            # ---- window: request the chunks the next half tile may need, then this half tile's first lookup ----
            window_requests()
            lookup(j)
            word_request()
            a.i(f"v_min_u32 {R1}, 1, %[rd]")
            a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
            a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
        E0, E1, E2, E3 = ESET[j % 2]
        a.wait_lds("e", f"---- step {j}: the bucket entry is back")
        a.i(f"3{j:02d}:", None)
        a.i(f"v_cmp_ge_u32 {V1}, {Q}, {E1}")
        a.i(f"v_cmp_ge_u32 {V2}, {Q}, {E2}")
        a.i(f"v_cmp_ge_u32 vcc, {Q}, {E3}", "beyond the third symbol of the bucket?")
        a.i(f"v_and_b32 {C}, %[cfield], {E0}", "(the cumulative shares its word with the index: 24 + 8 or 22 + 10 bits)")
        a.i(f"v_cndmask_b32_e64 {NXT}, {E1}, {E2}, {V1}")
        a.i(f"v_cndmask_b32_e64 {C}, {C}, {E1}, {V1}")
        a.i(f"v_cndmask_b32_e64 {NXT}, {NXT}, {E3}, {V2}")
        a.i(f"v_cndmask_b32_e64 {C}, {C}, {E2}, {V2}")
        if os.environ.get("GEN_NO_WALK") == "2":    # timing experiment only (results wrong): a branch that never waits for the VALU's vcc
            a.i(f"s_cbranch_execz 1{j:02d}f")
        elif not os.environ.get("GEN_NO_WALK"):     # (GEN_NO_WALK=1: timing experiment only, results wrong)
            a.i(f"s_cbranch_vccnz 1{j:02d}f", "-> walk the cdf table for those lanes (rare; it also leaves index - 2 in the entry)")
        a.i(f"2{j:02d}:", None)
        a.i(f"v_sub_u32 {PR}, {NXT}, {C}", "p")
        a.i(f"v_sub_u32 {D}, {Q}, {C}", "q - c")
        a.i(f"v_mad_u64_u32 v[120:121], {SD}, {T0}, {PR}, v[122:123]", "N = (state >> P) * p + (q - c)")
        a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}")
        a.i(f"v_cmp_lt_u32 vcc, {N1}, {R1}", "refill <=> N < 2^32 and words remain")
        a.wait_lds_all("candidate word (and everything older) is back")
        a.i(f"v_cndmask_b32 %[lo], {N0}, {WD}, vcc")
        last_of_half = (j + 1) in WINDOWS or j == 31
        if not last_of_half:
            lookup(j + 1)
        a.i(f"v_subbrev_co_u32 %[rd], {SD}, 0, %[rd], vcc")
        if not last_of_half:
            word_request()
        a.i(f"v_lshrrev_b32 {IDX}, %[ishift], {E0}", "symbol index = i0 + (q >= e1) + (q >= e2)   (off the chain)")
        a.i(f"v_addc_co_u32_e64 {IDX}, {SD}, 0, {IDX}, {V1}")
        a.i(f"v_addc_co_u32_e64 {SYM[(quad % 2) * 4 + pos]}, {SD}, %[minsym], {IDX}, {V2}", "the decoded symbol (min_symbol in a VGPR: one scalar operand per instruction)")
        if NARROW:
            pass                          # (the group leaves in a block of its own behind its last tile)
        elif pos == 1 and SYMBOL_MAJOR:
            for c in range(4):
                a.ds(f"ds_read_b32 v{144 + c}, %[trprev] offset:{(32 * (quad & 1) + c) * 144 + 32 * (quad >> 1)}", "x")
        elif pos == 1:
            a.ds(f"ds_read_b128 {X}, %[trprev] offset:{1152 * quad}", "x", f"previous tile, rows (lane>>3)+{8 * quad}")
        a.i(f"v_cndmask_b32 %[hi], {N1}, {N0}, vcc")
        if not last_of_half:
            a.i(f"v_min_u32 {R1}, 1, %[rd]")
            a.i(f"v_alignbit_b32 {T0}, %[hi], %[lo], %[P]")
            a.i(f"v_lshrrev_b32 {T1}, %[P], %[hi]")
        if pos == 2 and not NARROW:
            a.vmem(f"global_store_dwordx4 {GOFF[quad]}, {X}, s[80:81] \" CST_STORE_MOD \"", f"store{quad}")
        if pos == 3 and NARROW == 1:
            base = 134 + (quad % 2) * 4
            a.i(f"v_perm_b32 v144, v{base + 1}, v{base}, %[sel01]", f"symbols {4 * quad}..{4 * quad + 3} as bytes")
            a.i(f"v_perm_b32 v145, v{base + 3}, v{base + 2}, %[sel23]")
            a.i("v_or_b32 v144, v144, v145")
            a.ds(f"ds_write_b32 %[rowcur], v144 offset:{4 * quad}", "tile")
        elif pos == 3 and NARROW == 2:
            base = 134 + (quad % 2) * 4
            a.i(f"v_perm_b32 v144, v{base + 1}, v{base}, %[sel01]", f"symbols {4 * quad}..{4 * quad + 3} as int16")
            a.i(f"v_perm_b32 v145, v{base + 3}, v{base + 2}, %[sel01]")
            a.ds(f"ds_write_b32 %[rowcur], v144 offset:{8 * quad}", "tile")
            a.ds(f"ds_write_b32 %[rowcur], v145 offset:{8 * quad + 4}", "tile")
        elif pos == 3:
            base = 134 + (quad % 2) * 4
            a.ds(f"ds_write_b128 %[rowcur], v[{base}:{base + 3}] offset:{16 * quad}", "tile", f"symbols {4 * quad}..{4 * quad + 3}")
        if (j + 1) in WINDOWS:
            window_landing("---- inside the tile: the chunks of the part just decoded land")
            a.wait_lds_all("landed chunks visible")

    window_landing("---- end of tile")
    if NARROW == 4:
        a.wait_lds_all("landed chunks (and the tile's last quad) are in LDS")
        # ---- the tile leaves: eight row blocks of whole 128-byte lines (not in the book: see the top of the file) ----
        for k in range(8):
            a.i(f"ds_read_b128 v[144:147], %[trcur] offset:{8 * N8_ROW * k}")
            a.i("s_waitcnt lgkmcnt(0)")
            a.i(f"global_store_dwordx4 {GOFF[k]}, v[144:147], s[80:81] nt")
        a.i("s_add_u32 s80, s80, 0x80")
        a.i("s_addc_u32 s81, s81, 0")
    elif NARROW:
        a.wait_lds_all("landed chunks (and the tile's last quad) are in LDS")
        a.i(f"v_add_u32 %[rowcur], {128 // TILES_PER_GROUP}, %[rowcur]", "the next tile's bytes of the row")
        a.i("s_add_u32 s83, s83, 1")
        a.i(f"s_cmp_lg_u32 s83, {TILES_PER_GROUP}")
        a.i("s_cbranch_scc1 50f", "the group is not complete: nothing leaves")
        # ---- the group leaves: eight row blocks of whole 128-byte lines (not in the book: see the top of the file) ----
        for k in range(8):
            for c in range(4):
                a.i(f"ds_read_b32 v{144 + c}, %[trcur] offset:{8 * N8_ROW * k + 4 * c}")
            a.i("s_waitcnt lgkmcnt(0)")
            a.i(f"global_store_dwordx4 {GOFF[k]}, v[144:147], s[80:81] nt")
        a.i("v_subrev_u32 %[rowcur], 128, %[rowcur]")
        a.i("s_mov_b32 s83, 0")
        a.i("s_add_u32 s80, s80, 0x80")
        a.i("s_addc_u32 s81, s81, 0")
        a.i("50:", None)
    else:
        a.i("v_swap_b32 %[rowcur], %[rowprev]")
        a.i("v_swap_b32 %[trcur], %[trprev]")
        a.i("s_add_u32 s80, s80, %[tilestep]" if SYMBOL_MAJOR else "s_add_u32 s80, s80, 0x80")
        a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds_all("landed chunks visible to the next tile")
    a.i("s_cbranch_scc1 1b")
    a.i("s_branch 3f")

    # ---- out of line, per step: second-level entry for the lanes beyond their bucket's third symbol (vcc), the selects again,
    # ---- and the walk over the cdf table for whoever is still beyond (one copy of the two routines per entry register set)
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
        E0, E1, E2, E3 = ESET[st]
        # the bucket's slot is (bucket mod 32); the entry found there is the lane's if its first cumulative does not lie
        # above the quantile and it is a real entry (a free slot holds zeros; a slot owned by a LOWER bucket passes and leads to a
        # longer walk, which is correct; one owned by a higher bucket does not pass)
        a.i(f"{4 + st}:", None)
        a.i(f"s_mov_b64 {XSAVE}, exec")
        a.i("s_mov_b64 exec, vcc")
        a.i(f"v_lshrrev_b32 {TMPA}, {SSH}, {Q}", "slot | part of the bucket")
        a.i(f"v_and_b32 {TMPA}, {SBITS}, {TMPA}")
        a.i(f"v_lshl_add_u32 {TMPA}, {TMPA}, 4, %[lut]")
        if BUCKET_BITS == 11:
            a.i(f"ds_read_b128 {NEW_T}, {TMPA} offset:32768", "(behind the 2048 bucket entries)")
        else:
            a.i(f"v_add_u32 {TMPA}, {16 << BUCKET_BITS}, {TMPA}", "(behind the 4096 bucket entries: beyond the 16-bit offset field)")
            a.i(f"ds_read_b128 {NEW_T}, {TMPA}")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_and_b32 {TMPA}, %[cfield], {NEW[0]}")
        a.i(f"v_cmp_le_u32 vcc, {TMPA}, {Q}")
        a.i(f"v_cmp_gt_u32 {FLAGGED}, {NEW[1]}, {TMPA}")
        a.i(f"s_and_b64 vcc, vcc, {FLAGGED}")
        a.i("s_and_b64 exec, exec, vcc")
        for k in range(4):
            a.i(f"v_mov_b32 {ESET[st][k]}, {NEW[k]}")
        a.i(f"s_mov_b64 exec, {XSAVE}")
        a.i(f"s_setpc_b64 {RET}")
        # the walk: lanes in vcc; Q and the entry as in the step; leaves (C, NXT) and index - 2 in the entry
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
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    return a, clobbers


def main():
    global SYMBOL_MAJOR
    for SYMBOL_MAJOR, out in (((False, OUT),) if NARROW else ((False, OUT), (True, OUT_SM))):
        emit(out)
    SYMBOL_MAJOR = False


def emit(out):
    a, clobbers = gen()
    header = ["// GENERATED by scripts/gen_decode_loop_b16.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (32,64) ANS decoder for 12 < P <= 24 (bucket entries): see cst_ans_b16.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued), [rowcur] "+v"(row_cur), [rowprev] "+v"(row_prev),',
           '      [trcur] "+v"(tr_cur), [trprev] "+v"(tr_prev)',
           '    : [lut] "s"(lut_addr), [cdf] "s"(cdf_addr), [mask] "s"(mask), [P] "s"(P), [bsh] "s"(bucket_shift), [minsym] "v"(min_symbol), [cfield] "s"(c_field_mask), [ishift] "s"(index_shift),',
           '      [cmask] "s"(ring_mask), [wbase] "s"(words_base), [gbase] "s"(store_base), [ntiles] "s"(n_tiles),',
           '      [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr), [dump] "v"(dump_addr), [woff] "v"(words_off)' + (', [tilestep] "s"(tile_step_bytes)' if SYMBOL_MAJOR else '') +
           (', [sel01] "s"(0x0c0c0400u), [sel23] "s"(0x04000c0cu)' if NARROW == 1 else ', [sel01] "s"(0x05040100u)' if NARROW == 2 else ''),
           "    : " + ", ".join(f'"{c}"' for c in clobbers) + ");"]
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions per iteration incl. loop control)")


if __name__ == "__main__":
    main()
