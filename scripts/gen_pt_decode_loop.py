#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_pt_decode_loop.inc: the hand-scheduled gfx950 main loop of the ANS decoder
with ONE TABLE PER STREAM in its compact form (cst_ans_pt.hip, BASELINE config C3), (W,S,P) = (32,64,12).

One asm statement decodes all full 32-symbol tiles of a wave's 64 streams.  Per symbol the serial chain is
    q -> bucket index (ds_read_u8) -> EIGHT packed entries (two ds_read_b128 from the 16-byte aligned quad that holds the first
         candidate: a misaligned LDS read costs 5x, and a read with random lane addresses ~13 LDS cycles whatever its width --
         scripts/microbench/lds_tput.hip; until round 6: six entries by ds_read2_b64 + ds_read_b64, 37 cycles against 23)
      -> the entry of the smallest wrapping distance (q << 20 | 0xffffe) - entry among the first seven   [the eighth decides whether
         the lane must look further: a wave-uniform loop that is branched around]
      -> (c, p, index) -> N = (state >> P) * p + (q - c) -> refill? -> state' -> q'
i.e. two dependent LDS round trips and 24 issue slots; everything else of a step (ring read of the next candidate
word, shifted state halves, read-position update, symbol index -> symbol) is issued in the shadow of the bucket read.
Per tile: up to three 16-byte chunks of compressed words are requested at the top and landed in the lane's LDS ring
at the bottom (exactly as in gen_decode_loop.py); the tile of decoded symbols leaves for HBM at the end of the tile
(eight transposing 16-byte LDS reads + eight 128-byte-row-segment stores per lane; a second tile buffer that would
hide this costs 36 KiB of LDS the rows need).

Run:  python scripts/gen_pt_decode_loop.py   (rewrites the .inc; the .inc is checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

# GEN_PT_SUB=1 (round 5): the loop of the SUB-LANE decoder (ans_decode_pt_sub_kernel: k jump points per stream, 512-thread
# workgroups, two waves per SIMD).  What a second resident wave costs is LDS, so the symbol tile holds BYTES: the decoded
# index (< 256 on this path) goes into byte `pos` of the quad's register with one SDWA add (dst_sel:BYTE_n,
# UNUSED_PRESERVE: the same instruction count as the v_add3 of the int32 tile; the v_and that isolated the index is gone,
# the select for runs takes its slot between the vcc write and its reader), a quad is ONE ds_write_b32, rows are 36 bytes
# apart (2304 bytes per wave instead of 9216), and the tile leaves as int32 symbols: each 4-byte piece read back is widened
# by four SDWA adds of min_symbol (src1_sel:BYTE_n) in front of its 16-byte store.
SUB = bool(os.environ.get("GEN_PT_SUB"))
# GEN_PT_SUB=2: the same for FOUR waves per SIMD (sixteen per workgroup).  The ring shrinks to 16 slots (4 KiB per wave), which
# cannot hold two tiles' worth of words, so the window moves every HALF tile: 16 symbols consume at most 6 words, the ring keeps the 12
# below the read position (two halves), two chunk slots are requested at steps 0 and 16 and landed after steps 15 and 31 -- a chunk
# with words 4c .. 4c + 3 lands on the slots of words 4c + 16 ..., which lie at or above rd_top + 1 and are dead by then.  The first
# refill candidate of the NEXT half is read behind the landing (it may be in the chunk that has just landed).  And the statement's
# registers are renumbered from v100 - v177 down to v48 - v125: sixteen waves per workgroup leave every wave 128.
SUB16 = os.environ.get("GEN_PT_SUB") == "2"
REG_SHIFT = 52 if SUB16 else 0
SUB_ROW = 36          # bytes between the rows of the byte tile (9 dwords: the lanes' quad writes hit 64 different banks)
# GEN_PT_N8=1 (round 6, with GEN_PT_SUB): the sub-lane decoders writing an INT8 symbol matrix themselves.  A four-byte piece of the byte
# tile (four symbol INDICES) becomes four int8 symbols by ONE packed add of min_symbol to every byte (carries between the bytes
# suppressed: five VALU instructions instead of four SDWA adds) and leaves with one global_store_dword; a tile is 32 bytes of a row.
# (Adding min_symbol to the index bytes of the staged rows instead would be free -- and wrong: the search compares whole entries, and a
# run mark 0xfff with index byte 0xff exceeds the key's 0xffffe.)
N8 = bool(os.environ.get("GEN_PT_N8"))
assert not N8 or SUB
OUT = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc") / \
    (("cst_pt_decode_loop_sub16_n8.inc" if SUB16 else "cst_pt_decode_loop_sub_n8.inc") if N8 else
     "cst_pt_decode_loop_sub16.inc" if SUB16 else "cst_pt_decode_loop_sub.inc" if SUB else "cst_pt_decode_loop.inc")

K_CHUNKS = 2 if SUB16 else 3      # window chunks requested per tile (32 symbols * 12 bits = 12 words = 3 chunks) / per half tile
AHEAD_M1 = 11 if SUB16 else 23    # kPtAhead - 1
HALVES = (0, 16) if SUB16 else (0,)

N0, N1 = "v100", "v101"            # v[100:101] = N
DD = "v102"                        # v[102:103] = [q - c (0 for a run), 0]
PR, T0, T1, TT, R0, WD, RA, R1, QK, RA2 = (f"v{r}" for r in range(104, 114))
# WINDOW = 8 (the default since round 6): EIGHT consecutive entries per look -- seven candidates + the one that says "further on" --
# with the answer taken as the entry of the SMALLEST wrapping distance key - entry (entries above the key wrap to huge distances,
# the 0xffffffff sentinels behind a row to key + 1, more than any real distance): 7 subtractions + 3 v_min3_u32 instead of compare /
# select pairs, and the distance itself is (q - c) << 20 | ..., so the step's v_sub for q - c is gone.  Round 4 measured it with
# 2 x ds_read2_b64 from the aligned PAIR and did not adopt it (0.716 against 0.710 ms on the plain kernel); with rows that start on
# 16 bytes and the bucket index counting QUADS the eight entries are two ds_read_b128 -- 23 LDS cycles per wave instead of the 37
# of ds_read2_b64 + ds_read_b64 -- and the LDS-bound sub-lane decoder gains 15 % (C3: 0.410 -> 0.349 ms; plain kernel 0.713 -> 0.637).
# GEN_PT_WINDOW=6: the six-entry form of rounds 2 - 5 (needs the pair-aligned rows of those rounds: kept for the record).
WINDOW = int(os.environ.get("GEN_PT_WINDOW", "8"))
X = ["v116", "v117", "v118", "v119", "v126", "v127"] if WINDOW == 6 else ["v116", "v117", "v118", "v119", "v172", "v173", "v174", "v175"]
X_T, X45_T = "v[116:119]", ("v[126:127]" if WINDOW == 6 else "v[172:175]")
DM, DM2 = "v176", "v177"
E, PM1, D, IDXA, TS, IDX = (f"v{r}" for r in range(120, 126))
SYM = [f"v{128 + k}" for k in range(8)]
PK = ["v128", "v129"]                     # SUB: the packed quads (even / odd quad)
XB = [f"v{130 + k}" for k in range(4)]    # SUB: 4-byte pieces of the byte tile, read back transposed
XO = [(f"v[{136 + 4 * k}:{139 + 4 * k}]") for k in range(4)]
PEND = [(f"v[{152 + 4 * k}:{155 + 4 * k}]", [f"v{152 + 4 * k + j}" for j in range(4)]) for k in range(K_CHUNKS)]
LAND = [f"v{164 + k}" for k in range(K_CHUNKS)]
WANT, TMP, TADDR, TOFF = "v167", "v168", "v169", "v170"
SD, SAVE, M2, MORE, RUN, M3, M4 = "s[84:85]", "s[86:87]", "s[88:89]", "s[90:91]", "s[92:93]", "s[76:77]", "s[78:79]"
CLOBBERS = [f"v{r}" for r in range(100, 178)] + [f"s{r}" for r in range(74 if N8 else 76, 94)] + ["vcc", "scc", "memory"]


def wait_if_pending(a, tag, comment=None):
    if tag in a.lds:
        a.wait_lds(tag, comment)


def tail(a, sym_reg, first=False, last=False):
    """everything of a step that is off the chain; issued behind the bucket read of the NEXT step.
    last: the tile's last step -- the candidate word of the NEXT tile's first refill may be in a chunk that only lands at the end of
    this tile (two tiles in a row that consume their 12 words: data above ~11.5 bits per symbol), so its ring read waits until the
    landing is done (round 5: it used to be issued here, and read a stale slot -- tests/test_gpu_max_rate.py)."""
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
    a.i(f"v_lshl_or_b32 {QK}, %[lo], 20, %[fffe]", "search key: every entry of a bin <= q compares <=")
    if not first and SUB:
        pk, pos = sym_reg
        a.i(f"v_add_u32_sdwa {pk}, {E}, {IDXA} dst_sel:BYTE_{pos} dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD",
            "decoded index -> its byte of the quad")
    elif not first:
        a.i(f"v_cndmask_b32_e64 {IDXA}, 0, {D}, {RUN}", "inside a run: index + (q - c)")
        a.i(f"v_add3_u32 {sym_reg}, {IDX}, {IDXA}, %[minsym]", "decoded symbol")


def step(a, j):
    quad, pos = divmod(j, 4)
    sym_reg = (PK[quad % 2], pos) if SUB else SYM[(quad % 2) * 4 + pos]
    if "l1" in a.lds:
        a.wait_lds("l1", f"---- step {j}: first candidate is back")
    else:
        a.i(f"; ---- step {j}: (the landing's wait covered the bucket read)")
    a.i(f"v_lshl_add_u32 {RA2}, {R0}, {4 if WINDOW == 8 else 3}, %[rowaddr]")
    if WINDOW == 8:
        a.ds(f"ds_read_b128 {X_T}, {RA2}", "x", "eight consecutive entries from the 16-byte aligned quad that holds the first candidate")
        a.ds(f"ds_read_b128 {X45_T}, {RA2} offset:16", "x")
        if pos == 0 and j > 0:
            base = ((quad - 1) % 2) * 4
            if SUB:
                a.ds(f"ds_write_b32 %[rowcur], {PK[(quad - 1) % 2]} offset:{4 * (quad - 1)}", "tile", f"symbols {4 * quad - 4}..{4 * quad - 1}")
            else:
                a.ds(f"ds_write_b128 %[rowcur], v[{128 + base}:{131 + base}] offset:{16 * (quad - 1)}", "tile", f"symbols {4 * quad - 4}..{4 * quad - 1}")
        a.wait_lds("x")
        a.i(f"v_cmp_le_u32_e64 {MORE}, {X[7]}, {QK}", "eighth entry <= q: the bin lies further on")
        for i in range(7):
            a.i(f"v_sub_u32 {X[i]}, {QK}, {X[i]}", "wrapping distance: small and >= 0 for the entries at or below the key" if i == 0 else None)
        a.i(f"v_min3_u32 {DM}, {X[0]}, {X[1]}, {X[2]}")
        a.i(f"s_cmp_lg_u64 {MORE}, 0")
        a.i(f"v_min3_u32 {DM}, {DM}, {X[3]}, {X[4]}")
        a.i(f"v_min3_u32 {DM}, {DM}, {X[5]}, {X[6]}")
        a.i("s_branch 4f" if os.environ.get("GEN_NO_MORE") else "s_cbranch_scc0 4f")      # (GEN_NO_MORE: timing experiment only)
        # wave-uniform continuation for the lanes in MORE (the others re-read their eight entries and keep their distance)
        a.i("3:")
        a.i(f"v_cndmask_b32_e64 {TS}, 0, 16, {MORE}")
        a.i(f"v_add_u32 {RA2}, {RA2}, {TS}", "continue from the fifth entry (the next quad)")
        a.i(f"ds_read_b128 {X_T}, {RA2}")
        a.i(f"ds_read_b128 {X45_T}, {RA2} offset:16")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_cmp_le_u32_e64 {M2}, {X[7]}, {QK}")
        for i in range(7):
            a.i(f"v_sub_u32 {X[i]}, {QK}, {X[i]}")
        a.i(f"v_min3_u32 {DM2}, {X[0]}, {X[1]}, {X[2]}")
        a.i(f"v_min3_u32 {DM2}, {DM2}, {X[3]}, {X[4]}")
        a.i(f"v_min3_u32 {DM2}, {DM2}, {X[5]}, {X[6]}")
        a.i(f"v_min_u32 {DM}, {DM}, {DM2}", "(a lane that did not move read the same entries again)")
        a.i(f"s_and_b64 {MORE}, {MORE}, {M2}")
        a.i(f"s_cmp_lg_u64 {MORE}, 0")
        a.i("s_cbranch_scc1 3b")
        a.i("4:")
        a.i(f"v_sub_u32 {E}, {QK}, {DM}", "the entry itself")
        a.i(f"v_bfe_u32 {PM1}, {E}, 8, 12", "p - 1, or the run mark")
        a.i(f"v_lshrrev_b32 {D}, 20, {DM}", "q - c")
    else:
        a.ds(f"ds_read2_b64 {X_T}, {RA2} offset1:1", "x", "six consecutive entries from the aligned pair that holds the first candidate")
        a.ds(f"ds_read_b64 {X45_T}, {RA2} offset:16", "x")
        if pos == 0 and j > 0:
            base = ((quad - 1) % 2) * 4
            if SUB:
                a.ds(f"ds_write_b32 %[rowcur], {PK[(quad - 1) % 2]} offset:{4 * (quad - 1)}", "tile", f"symbols {4 * quad - 4}..{4 * quad - 1}")
            else:
                a.ds(f"ds_write_b128 %[rowcur], v[{128 + base}:{131 + base}] offset:{16 * (quad - 1)}", "tile", f"symbols {4 * quad - 4}..{4 * quad - 1}")
        a.wait_lds("x")
        a.i(f"v_cmp_le_u32 vcc, {X[1]}, {QK}")
        a.i(f"v_cmp_le_u32_e64 {M2}, {X[2]}, {QK}")
        a.i(f"v_cmp_le_u32_e64 {M3}, {X[3]}, {QK}")
        a.i(f"v_cndmask_b32 {E}, {X[0]}, {X[1]}, vcc")
        a.i(f"v_cmp_le_u32_e64 {M4}, {X[4]}, {QK}")
        a.i(f"v_cndmask_b32_e64 {E}, {E}, {X[2]}, {M2}")
        a.i(f"v_cmp_le_u32_e64 {MORE}, {X[5]}, {QK}", "sixth entry <= q: the bin lies further on")
        a.i(f"v_cndmask_b32_e64 {E}, {E}, {X[3]}, {M3}")
        a.i(f"s_cmp_lg_u64 {MORE}, 0")
        a.i(f"v_cndmask_b32_e64 {E}, {E}, {X[4]}, {M4}")
        a.i("s_branch 4f" if os.environ.get("GEN_NO_MORE") else "s_cbranch_scc0 4f")      # (GEN_NO_MORE: timing experiment only)
        # wave-uniform continuation for the lanes in MORE (the others re-read their six entries and keep E)
        a.i("3:")
        a.i(f"v_cndmask_b32_e64 {TS}, 0, 16, {MORE}")
        a.i(f"v_add_u32 {RA2}, {RA2}, {TS}", "continue from the fifth entry (8-byte aligned)")
        a.i(f"ds_read2_b64 {X_T}, {RA2} offset1:1")
        a.i(f"ds_read_b64 {X45_T}, {RA2} offset:16")
        a.i("s_waitcnt lgkmcnt(0)")
        a.i(f"v_cmp_le_u32 vcc, {X[2]}, {QK}", "(the first two are known to be <= q)")
        a.i(f"v_cmp_le_u32_e64 {M3}, {X[3]}, {QK}")
        a.i(f"v_cmp_le_u32_e64 {M4}, {X[4]}, {QK}")
        a.i(f"v_cndmask_b32 {TS}, {X[1]}, {X[2]}, vcc")
        a.i(f"v_cmp_le_u32_e64 {M2}, {X[5]}, {QK}")
        a.i(f"v_cndmask_b32_e64 {TS}, {TS}, {X[3]}, {M3}")
        a.i(f"v_cndmask_b32_e64 {TS}, {TS}, {X[4]}, {M4}")
        a.i(f"v_cndmask_b32_e64 {E}, {E}, {TS}, {MORE}")
        a.i(f"s_and_b64 {MORE}, {MORE}, {M2}")
        a.i(f"s_cmp_lg_u64 {MORE}, 0")
        a.i("s_cbranch_scc1 3b")
        a.i("4:")
        a.i(f"v_sub_u32 {D}, {QK}, {E}")
        a.i(f"v_bfe_u32 {PM1}, {E}, 8, 12", "p - 1, or the run mark")
        a.i(f"v_lshrrev_b32 {D}, 20, {D}", "q - c")
    a.i(f"v_cmp_eq_u32_e64 {RUN}, {PM1}, %[fff]", "run of unit probabilities: (c, p) = (q, 1)")
    a.i(f"v_add_u32 {PR}, 1, {PM1}")
    a.i(f"v_cndmask_b32_e64 {DD}, {D}, 0, {RUN}")
    a.i(f"v_cndmask_b32_e64 {PR}, {PR}, 1, {RUN}")
    a.i(f"v_mad_u64_u32 v[100:101], {SD}, {T0}, {PR}, v[102:103]", "N = (state >> P) * p + (q - c)")
    a.i(f"v_mad_u32_u24 {N1}, {T1}, {PR}, {N1}")
    a.i(f"v_cmp_lt_u32 vcc, {N1}, {R1}", "refill <=> N < 2^32 and words remain")
    wait_if_pending(a, "w", "candidate word is back")      # (older than the entries: normally retired with them)
    if SUB:
        a.i(f"v_cndmask_b32_e64 {IDXA}, 0, {D}, {RUN}", "inside a run: index + (q - c)   (one instruction between a VALU write of vcc and its VALU reader)")
    else:
        a.i(f"v_and_b32 {IDX}, 0xff, {E}", "(one instruction between a VALU write of vcc and its VALU reader)")
    a.i(f"v_cndmask_b32 %[lo], {N0}, {WD}, vcc")
    a.i(f"v_and_or_b32 {TT}, %[lo], %[bmask], %[l1base]", "bucket of the next quantile (index interleaved by lane)")
    a.ds(f"ds_read_u8 {R0}, {TT}", "l1", "<- end of the serial chain")
    tail(a, sym_reg, last=(j == 31 or (SUB16 and j == 15)))


def gen():
    a = Asm()
    a.i("v_mov_b32 v103, 0")
    if N8:
        a.i("s_and_b32 s74, %[minsym], 0xff")
        a.i("s_mul_i32 s74, s74, 0x01010101", "min_symbol in every byte")
        a.i("s_and_b32 s75, s74, 0x7f7f7f7f")
    a.i("s_mov_b64 s[80:81], %[gbase]", "store base of the current tile, bumped by 128 B per iteration")
    a.i("s_mov_b32 s82, %[ntiles]")
    # first bucket read and the off-chain values of step 0
    a.i(f"v_and_or_b32 {TT}, %[lo], %[bmask], %[l1base]")
    a.ds(f"ds_read_u8 {R0}, {TT}", "l1")
    tail(a, None, first=True)
    a.i("1:")
    first = len(a.events)
    lds_entry, vm_entry = list(a.lds), list(a.vm)

    def window_requests():
        """request the chunks the next tile (half tile) may need; they land at the end of this one"""
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

    def window_landing(comment):
        a.wait_vm(f"chunk{K_CHUNKS - 1}", comment)
        for k in range(K_CHUNKS):
            r = PEND[k][1]
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[0]}, {r[1]} offset1:1", "land")
            a.ds(f"ds_write2st64_b32 {LAND[k]}, {r[2]}, {r[3]} offset0:2 offset1:3", "land")

    for j in range(32):
        if j in HALVES:
            window_requests()
        step(a, j)
        if SUB16 and j == 15:
            window_landing("---- middle of the tile: the first half's chunks land")
            a.wait_lds("land", "visible before the second half reads its first candidate word")
            a.ds(f"ds_read_b32 {WD}, {RA}", "w", "candidate word of step 16's refill: only now, behind the landing")

    # ---- end of tile: last quad -> tile row, tile -> HBM, chunks -> ring ----
    if SUB:
        a.ds(f"ds_write_b32 %[rowcur], {PK[1]} offset:28", "tile", "symbols 28..31")
    else:
        a.ds(f"ds_write_b128 %[rowcur], v[132:135] offset:112", "tile", "symbols 28..31")
    a.wait_lds("tile")
    for half in range(2):
        for k in range(4):
            if SUB:
                a.ds(f"ds_read_b32 {XB[k]}, %[trcur] offset:{8 * SUB_ROW * (4 * half + k)}", "xo", f"rows (lane>>3)+{8 * (4 * half + k)}")
            else:
                a.ds(f"ds_read_b128 {XO[k]}, %[trcur] offset:{1152 * (4 * half + k)}", "xo", f"rows (lane>>3)+{8 * (4 * half + k)}")
        for k in range(4):
            if SUB:
                if f"xo" in a.lds:
                    n_after = 3 - k                      # pieces requested after piece k
                    a.events.append(("wait_lds", None, n_after))
                    a.lds = a.lds[len(a.lds) - n_after:] if n_after else []
                    a.i(f"s_waitcnt lgkmcnt({n_after})", f"piece {k} is back")
                if N8:
                    t, u = f"v{136 + 4 * k}", f"v{137 + 4 * k}"
                    a.i(f"v_and_b32 {t}, 0x7f7f7f7f, {XB[k]}", "four indices + min_symbol, byte by byte (mod 256): the low seven bits add ...")
                    a.i(f"v_xor_b32 {u}, s74, {XB[k]}")
                    a.i(f"v_add_u32 {t}, s75, {t}")
                    a.i(f"v_and_b32 {u}, 0x80808080, {u}", "... the top bits by xor: no carry crosses a byte")
                    a.i(f"v_xor_b32 {t}, {t}, {u}")
                    a.vmem(f"global_store_dword %[goff{4 * half + k}], {t}, s[80:81] \" CST_STORE_MOD \"", "store", "four int8 symbols of a row")
                    continue
                for b in range(4):
                    a.i(f"v_add_u32_sdwa v{136 + 4 * k + b}, %[minsym], {XB[k]} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{b}",
                        "index -> int32 symbol" if b == 0 else None)
            elif k == 0:
                a.wait_lds("xo")
            a.vmem(f"global_store_dwordx4 %[goff{4 * half + k}], {XO[k]}, s[80:81] \" CST_STORE_MOD \"", "store")
    window_landing("the chunk loads are older than this tile's stores")
    a.i("s_add_u32 s80, s80, 0x20" if N8 else "s_add_u32 s80, s80, 0x80")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.wait_lds("land", "landed chunks visible to the next tile; the bucket read of its first step is older")
    a.ds(f"ds_read_b32 {WD}, {RA}", "w", "candidate word of the next tile's first refill: only now, behind the landing")
    a.i("s_cbranch_scc1 1b")
    # the back edge must leave the queues as the loop entry found them (modulo completed operations)
    lds_end, vm_end, notes = a.verify_loop(first, list(a.lds), list(a.vm), passes=1)
    assert lds_end == a.lds and vm_end == a.vm, (lds_end, a.lds, vm_end, a.vm)
    assert [t for t in lds_entry if t not in ("l1", "w")] == [] and a.lds == ["w"], (lds_entry, a.lds)
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all("(the candidate word requested for a tile that does not come)")
    return a, notes


def main():
    a, notes = gen()
    header = ["// GENERATED by scripts/gen_pt_decode_loop.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled per-stream-table ANS decoder" + (", int8 symbol matrix" if N8 else "") + ": see pt_decode_tiles_loop in cst_ans_pt.hip."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [lo_issued] "+v"(lo_issued)',
           '    : [bmask] "s"(bucket_mask), [cmask] "s"(ring_mask), [P] "s"(P), [fffe] "s"(0xffffeu), [fff] "s"(0xfffu), [minsym] "s"(min_symbol),',
           '      [wbase] "s"(words_base), [gbase] "s"(store_base), [ntiles] "s"(n_tiles),',
           '      [l1base] "v"(l1_lane_addr), [rowaddr] "v"(row_addr), [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr),',
           '      [dump] "v"(dump_addr), [woff] "v"(words_off), [rowcur] "v"(tile_row_addr), [trcur] "v"(tile_tr_addr),',
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
    text = a.render(header, ops)
    if REG_SHIFT:
        import re
        def shift(m):
            n = int(m.group(2))
            return f"{m.group(1)}{n - REG_SHIFT}" if n >= 100 else m.group(0)
        text = re.sub(r"(\bv\[?)(\d+)", shift, text)                     # v123, v[123
        text = re.sub(r"(v\[\d+:)(\d+)", shift, text)                    # the upper end of v[123:126]
    OUT.write_text(text)
    print(f"wrote {OUT} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


if __name__ == "__main__":
    main()
