#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_range_encode_loop{,_2f}.inc: the hand-scheduled gfx950 main loop of the
(32,64) range encoder -- ONE asm statement that encodes all full 32-symbol tiles of a wave's 64 streams, first tile
first (a queue).  Same skeleton as gen_encode_loop.py (software pipeline of quads that runs on across tile boundaries,
two LDS tile buffers, symbols requested two tiles ahead into two register sets, 64-byte word groups leaving the LDS
ring at fixed places of a tile); what differs is the coder step and the direction.

The step (queue.rs:612-705) in the "held word" form of the lazy carry.  The reference holds back the words of an
Inverted situation (the interval straddles a word boundary) until it knows whether a carry reaches them: first + 1, then
zeros, or first, then 0xffffffff.  That is addition with carry on the number the emitted words spell, and a carry
can only ever happen while words are held back (in the Normal situation lower + range does not wrap).  So the
statement keeps just the LAST word in a register (LW): a carry out of `lower + scale * c` is added to it, it goes to
the ring when the next word is produced, and no situation is tracked at all.  The one thing this cannot do in
registers is carry into a word that already left (LW == 0xffffffff when the carry arrives: an Inverted run of two or
more words -- needs ~2^-20 luck on model-distributed data); then a sticky flag is raised and the caller repeats the
wave's streams with the general C++ step (RangeEncHeld::carry_back).
    scale = range >> P;  nr = scale * p;  nl = lower + scale * c  (carry -> LW);  renorm <=> nr < 2^32 <=> hi(nr) == 0
    ring[wr] = LW (always);  wr += renorm;  LW = renorm ? hi(nl) : LW;  lower, range = renorm ? (lo << 32) : (nl, nr)
17 VALU + 1 SALU + the ring write; the dependent chain is only alignbit -> mad -> mad24 -> cmp -> select.

Variants: one word group (16 words) leaves per tile (P <= 16: a tile emits at most 16 words), or two (P <= 24).

Run:  python scripts/gen_range_encode_loop.py   (rewrites the .inc files; they are checked in)
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

# GEN_RANGE_CK=1 (round 5): the stream-major loops with JUMP POINTS -- RangeEncoder::pos() (queue.rs:182-196: words emitted
# including held-back ones, and the coder state (lower, range)) noted in front of every chunk of `cktiles` tiles: a scalar
# countdown and a branch at the top of every tile, four instructions + three stores k times per stream.  The held word counts:
# pos = wr + 1 (wr is -1 until the first word exists).  Written to cst_range_encode_loop{,_2f}_ck.inc; the words are the plain loop's.
CKPT = bool(os.environ.get("GEN_RANGE_CK"))
# GEN_RANGE_N8=1 (round 6, with GEN_RANGE_CK): the same loops over an INT8 symbol matrix (the reference's Symbol is generic:
# queue.rs:612, quantize.rs:229-255).  A tile is 32 BYTES of a row: a lane requests four symbols with one global_load_dword into the last
# register of its quad, and sign-extends them into the quad (three v_bfe_i32 and a shift) in front of the transposed ds_write_b128 --
# one more VALU instruction per symbol, a quarter of the symbol bytes.  Written to cst_range_encode_loop{,_2f}_ck_n8.inc; the kernel
# that runs them (range_encode_ckpt_kernel<FLUSHES, 1>) also serves the plain call (one chunk: the jump point is the start).
N8 = bool(os.environ.get("GEN_RANGE_N8"))
assert not N8 or CKPT
CSRC = Path(os.environ.get("GEN_CSRC") or Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc")
OUT = {(1, False): CSRC / "cst_range_encode_loop.inc", (2, False): CSRC / "cst_range_encode_loop_2f.inc",
       (1, True): CSRC / "cst_range_encode_loop_sm.inc", (2, True): CSRC / "cst_range_encode_loop_2f_sm.inc"}
FLUSHES = 1
# SYMBOL_MAJOR (cst_range_encode_loop*_sm.inc): symbols[t][stream].  Only the staging differs (gen_encode_loop.py): a load
# instruction reads 16 symbol rows of 64 bytes (lane l: row l >> 2, streams 4 (l & 3) .. + 3), the four symbols of a
# register go to four ROWS of the LDS tile (tile[stream][t], stride 36 words: 64 different banks per ds_write_b32), and the
# base moves by 32 * n_streams * 4 bytes per tile (an operand) instead of 128.  Full waves only.
SYMBOL_MAJOR = False


def regs(base, n=4):
    return [f"v{base + i}" for i in range(n)]


def tup(base, n=4):
    return f"v[{base}:{base + n - 1}]"


R = {"A": [tup(100 + 4 * k) for k in range(8)], "B": [tup(132 + 4 * k) for k in range(8)]}
S = [regs(164 + 4 * i) for i in range(4)]                    # four symbol quads (three live + one being filled)
S_T = [tup(164 + 4 * i) for i in range(4)]
E = [[regs(180 + 8 * e + 2 * i, 2) for i in range(4)] for e in range(2)]     # (c, p) of a quad's symbols
E_T = [[tup(180 + 8 * e + 2 * i, 2) for i in range(4)] for e in range(2)]
LO0, LO1, RG0, RG1, LW = (f"v{r}" for r in (212, 213, 214, 215, 216))
LO_T = tup(212, 2)
SC0, SC1 = "v218", "v219"
NR0, NR1, NL0, NL1, T0, T1, X0, X1 = (f"v{r}" for r in range(220, 228))
NR_T, NL_T, T_T, X_T = (tup(220 + 2 * i, 2) for i in range(4))
RA, EA = "v228", "v229"
FD = [(tup(230 + 4 * k, 2), tup(232 + 4 * k, 2), tup(230 + 4 * k)) for k in range(4)]
NCH, LIM, FADDR, FOFF = "v246", "v247", "v248", "v249"
CARRY, SAVE, OVF, SLOW, JUNK, MASK = "s[84:85]", "s[86:87]", "s[90:91]", "s[92:93]", "s[94:95]", "s[96:97]"
CLOBBERS = [f"v{r}" for r in range(100, 250)] + [f"s{r}" for r in range(80, 98)] + (["s78"] if CKPT else []) + ["vcc", "scc", "memory"]
ROW = ["%[row0]", "%[row1]"]          # the lane's own row in tile buffer 0 / 1
TR = ["%[tr0]", "%[tr1]"]             # transposed write address in tile buffer 0 / 1


def step(a, c, p):
    a.i(f"v_alignbit_b32 {SC0}, {RG1}, {RG0}, %[P]", "scale = range >> P")
    a.i(f"v_lshrrev_b32 {SC1}, %[P], {RG1}")
    a.i(f"v_mad_u64_u32 {NR_T}, {JUNK}, {SC0}, {p}, 0", "nr = scale * p")
    a.i(f"v_mad_u64_u32 {NL_T}, {CARRY}, {SC0}, {c}, {LO_T}", "nl = lower + scale_lo * c, carry out of bit 63")
    a.i(f"v_mad_u32_u24 {NR1}, {SC1}, {p}, {NR1}", "(scale_hi < 2^(32-P) <= 2^24, p < 2^24)")
    a.i(f"v_mov_b32 {T0}, {NL1}", "[hi(nl), 0]")
    a.i(f"v_cmp_eq_u32 vcc, 0, {NR1}", "renorm <=> nr < 2^32")
    a.i(f"v_mad_u64_u32 {X_T}, {JUNK}, {SC1}, {c}, {T_T}", "X0 = hi(nl) + scale_hi * c, X1 = its carry (0 / 1)")
    a.i(f"v_lshlrev_b32 {RA}, 8, %[wr]")
    a.i(f"v_cndmask_b32_e32 {RG1}, {NR1}, {NR0}, vcc", "range = renorm ? nr << 32 : nr")
    a.i(f"v_cndmask_b32_e64 {RG0}, {NR0}, 0, vcc")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[c3f00], %[lanebase]")
    a.i(f"v_addc_co_u32_e64 {LW}, {OVF}, {LW}, {X1}, {CARRY}", "a carry out of lower reaches the held word")
    a.ds(f"ds_write_b32 {RA}, {LW}", "W", "the held word, always written; it counts once wr moves on")
    a.i(f"s_or_b64 {SLOW}, {SLOW}, {OVF}", "... and must not leave it (sticky: the caller repeats the streams)")
    a.i(f"v_addc_co_u32_e64 %[wr], {JUNK}, 0, %[wr], vcc")
    a.i(f"v_cndmask_b32_e32 {LW}, {LW}, {X0}, vcc", "the new held word is hi(nl)")
    a.i(f"v_cndmask_b32_e32 {LO1}, {X0}, {NL0}, vcc", "lower = renorm ? nl << 32 : nl")
    a.i(f"v_cndmask_b32_e64 {LO0}, {NL0}, 0, vcc")


def read_syms(a, g, buf, quad):
    a.ds(f"ds_read_b128 {S_T[g % 4]}, {ROW[buf]} offset:{16 * quad}", f"S{g}")


def fetch_entries(a, g):
    for i, sym in enumerate(S[g % 4]):
        a.i(f"v_lshl_add_u32 {EA}, {sym}, 3, %[tbl]")
        a.ds(f"ds_read_b64 {E_T[g % 2][i]}, {EA}", f"E{g}")


def fold_minmax(a, g):
    x, y, z, w = S[g % 4]
    a.i(f"v_max3_i32 %[smax], %[smax], {x}, {y}")
    a.i(f"v_max3_i32 %[smax], %[smax], {z}, {w}")
    a.i(f"v_min3_i32 %[smin], %[smin], {x}, {y}")
    a.i(f"v_min3_i32 %[smin], %[smin], {z}, {w}")


def advance_base(a):
    """s[80:81] -> symbols of the next tile to request; stays on the last tile once every tile has been requested"""
    a.i("s_cmp_lg_u32 s83, 0")
    a.i("s_cselect_b32 s88, %[tilestep], 0" if SYMBOL_MAJOR else "s_cselect_b32 s88, 0x20, 0" if N8 else "s_cselect_b32 s88, 0x80, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_add_u32 s80, s80, s88")
    a.i("s_addc_u32 s81, s81, 0")
    a.i("s_sub_u32 s83, s83, s89")


def stage_wait(a, name):
    a.wait_vm(f"ld{name}", f"symbols in set {name} have arrived")


def stage_one(a, name, buf, k):
    if SYMBOL_MAJOR:
        base = {"A": 100, "B": 132}[name]
        for c in range(4):
            a.ds(f"ds_write_b32 {TR[buf]}, v{base + 4 * k + c} offset:{(16 * (k >> 1) + c) * 144 + 64 * (k & 1)}", "tl")
        return
    if N8:
        b = {"A": 100, "B": 132}[name] + 4 * k
        for c in range(3):
            a.i(f"v_bfe_i32 v{b + c}, v{b + 3}, {8 * c}, 8", "four int8 symbols -> the quad" if c == 0 else None)
        a.i(f"v_ashrrev_i32 v{b + 3}, 24, v{b + 3}")
    a.ds(f"ds_write_b128 {TR[buf]}, {R[name][k]} offset:{1152 * k}", "tl")


def load_one(a, name, k):
    if N8:
        a.vmem(f"global_load_dword v{ {'A': 100, 'B': 132}[name] + 4 * k + 3}, %[goff{k}], s[80:81] nt", f"ld{name}")
    else:
        a.vmem(f"global_load_dwordx4 {R[name][k]}, %[goff{k}], s[80:81] nt", f"ld{name}")
    if k == 7:
        advance_base(a)


def group_reads(a, ks, decide):
    """ring reads of the 64-byte word group that may be complete (4 chunks; lgkmcnt only counts to 15: two per quad)"""
    if decide:
        # decide NOW whether the group is complete: words written after these reads must not count.  wr is -1 until
        # the first word exists (signed arithmetic).
        a.i(f"v_sub_u32 {NCH}, %[wr], %[flushed]")
        a.i(f"v_ashrrev_i32 {NCH}, 4, {NCH}")
        a.i(f"v_med3_i32 {NCH}, {NCH}, 0, 1", "whole 16-word groups to move now: 0 or 1")
    for k in ks:
        a.i(f"v_add_lshl_u32 {FADDR}, %[flushed], {4 * k}, 8")
        a.i(f"v_and_or_b32 {FADDR}, {FADDR}, %[c3f00], %[lanebase]")
        a.ds(f"ds_read2st64_b32 {FD[k][0]}, {FADDR} offset1:1", "fl")
        a.ds(f"ds_read2st64_b32 {FD[k][1]}, {FADDR} offset0:2 offset1:3", "fl")


def store_one(a, k):
    """one 16-byte chunk of the word group, for the lanes whose group is complete and inside the slab.  The memory
    instructions of a tile are spread over its steps, one per step: a lone wave issues in order, and a burst of them
    stalls it on the depth of the memory pipelines' queues (measured: 8 loads + 8 tile writes + 4 stores back to back
    cost ~40 cycles per symbol)."""
    if k == 0:
        a.i(f"v_add_u32 {LIM}, 16, %[flushed]")
        a.i(f"v_lshl_add_u32 {FOFF}, %[flushed], 2, %[slaboff]")
        a.i(f"v_cmp_le_u32 {MASK}, {LIM}, %[cap]", "group inside the slab (cap % 16 == 0 on this path)")
        a.i(f"v_cmp_ne_u32 {SAVE}, 0, {NCH}")
        a.i(f"s_and_b64 {MASK}, {MASK}, {SAVE}")
        a.wait_lds("fl", cap=True)
    a.i(f"s_mov_b64 exec, {MASK}")
    a.vmem(f"global_store_dwordx4 {FOFF}, {FD[k][2]}, %[wbase] offset:{16 * k}", "st")
    a.i("s_mov_b64 exec, -1", "(the statement runs on full waves only)")
    if k == 3:
        a.i(f"v_lshl_add_u32 %[flushed], {NCH}, 4, %[flushed]")


def jump_point(a, label):
    """top of a tile, before its first step: does a chunk start here?  then note RangeEncoder::pos() (the stores are not in
    the generator's book: they end with vmcnt(0), and waiting for more than the book knows is always safe)"""
    a.i("s_sub_u32 s78, s78, 1")
    a.i("s_cmp_lg_u32 s78, 0")
    a.i(f"s_cbranch_scc1 {label}f")
    a.i(f"v_add_u32 {RA}, 1, %[wr]", "pos = words so far, the held one included")
    a.i(f"v_lshlrev_b32 {EA}, 2, %[ckidx]")
    a.i(f"global_store_dword {EA}, {RA}, %[ckpos]")
    a.i(f"v_lshlrev_b32 {EA}, 3, %[ckidx]")
    a.i(f"global_store_dwordx2 {EA}, {LO_T}, %[cklower]")
    a.i(f"global_store_dwordx2 {EA}, {tup(214, 2)}, %[ckrange]")
    a.i("v_add_u32 %[ckidx], 1, %[ckidx]")
    a.i("s_mov_b32 s78, %[cktiles]")
    a.i("s_waitcnt vmcnt(0)")
    a.i(f"{label}:")


def half(a, h, g0):
    """one tile: register set / tile buffer h (0 = A), global quad indices g0 .. g0+7 are its quads 0 .. 7"""
    own, other = "AB"[h], "AB"[1 - h]
    a.i(f"; ---- tile in buffer {h} (symbols came from set {own})")
    if CKPT:
        jump_point(a, 7 + h)
    # what follows step s of the tile (s = 0 .. 31)
    after = {}
    for k in range(4):
        after.setdefault(8 + k, []).append(lambda k=k: store_one(a, k))
        if FLUSHES == 2:
            after.setdefault(24 + k, []).append(lambda k=k: store_one(a, k))
    for k in range(8):
        # next tile's symbols -> the other tile buffer, then request tile + 3 into the freed register
        after.setdefault(12 + k, []).append(lambda k=k: (stage_wait(a, other) if k == 0 else None, stage_one(a, other, 1 - h, k)))
        after.setdefault(20 + k, []).append(lambda k=k: load_one(a, other, k))
    for j in range(8):
        g = g0 + j
        # the pipeline runs on into the next tile: quads 8 and 9 are quads 0 and 1 of the other buffer
        if f"S{g + 1}" in a.lds:
            a.wait_lds(f"S{g + 1}", f"quad {j}: symbols of the next quad are back", cap=True)
        far = j + 2
        read_syms(a, g + 2, h if far < 8 else 1 - h, far % 8)
        fetch_entries(a, g + 1)
        if f"E{g}" in a.lds:
            a.wait_lds(f"E{g}", f"entries of quad {j} are back", cap=True)
        if j in (0, 1):
            group_reads(a, (0, 1) if j == 0 else (2, 3), j == 0)
        if FLUSHES == 2 and j in (4, 5):
            group_reads(a, (0, 1) if j == 4 else (2, 3), j == 4)
        fold_minmax(a, g)
        for i, (c, p) in enumerate(E[g % 2]):
            step(a, c, p)
            for f in after.get(4 * j + i, []):
                f()


def load_set(a, name):
    for k in range(8):
        load_one(a, name, k)


def stage_set(a, name, buf):
    stage_wait(a, name)
    for k in range(8):
        stage_one(a, name, buf, k)


def gen():
    a = Asm()
    for dst, src in ((LO0, "%[lo0]"), (LO1, "%[lo1]"), (RG0, "%[rg0]"), (RG1, "%[rg1]"), (LW, "%[lw]")):
        a.i(f"v_mov_b32 {dst}, {src}")
    a.i(f"v_mov_b32 {T1}, 0")
    a.i(f"s_mov_b64 {SLOW}, 0")
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the FIRST tile of stream s0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    if CKPT:
        a.i("s_mov_b32 s78, 1", "the first tile starts chunk 0")
    load_set(a, "A")                  # first tile
    load_set(a, "B")                  # the one after
    stage_set(a, "A", 0)
    load_set(a, "A")                  # two after
    read_syms(a, 0, 0, 0)
    read_syms(a, 1, 0, 1)
    a.wait_lds("S0")
    fetch_entries(a, 0)
    a.i("1:")
    first = len(a.events)
    half(a, 0, 0)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    half(a, 1, 8)
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    ren = {"S16": "S0", "S17": "S1", "E16": "E0"}
    lds_back = [ren.get(t, t) for t in a.lds]
    lds_end, vm_end, notes = a.verify_loop(first, lds_back, a.vm, passes=1)
    lds_end = [ren.get(t, t) for t in lds_end]
    assert (lds_end == lds_back and vm_end == a.vm), (lds_end, lds_back, vm_end, a.vm)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    a.wait_lds_all()
    for dst, src in ((LO0, "%[lo0]"), (LO1, "%[lo1]"), (RG0, "%[rg0]"), (RG1, "%[rg1]"), (LW, "%[lw]")):
        a.i(f"v_mov_b32 {src}, {dst}")
    a.i("s_or_b32 s92, s92, s93")
    a.i("v_mov_b32 %[slow], s92")
    return a, notes


def emit(flushes, symbol_major=False):
    global FLUSHES, SYMBOL_MAJOR
    FLUSHES, SYMBOL_MAJOR = flushes, symbol_major
    a, notes = gen()
    header = ["// GENERATED by scripts/gen_range_encode_loop.py -- do not edit by hand (edit the generator and re-run it).",
              f"// Main loop of the hand-scheduled (32,64) range encoder, {flushes} word group(s) per tile: see cst_range_fast.hip."]
    ops = ['    : [lo0] "+v"(lo0), [lo1] "+v"(lo1), [rg0] "+v"(rg0), [rg1] "+v"(rg1), [lw] "+v"(lw), [wr] "+v"(wr), [flushed] "+v"(flushed),',
           '      [smin] "+v"(smin), [smax] "+v"(smax), [slow] "=v"(slow)' + (', [ckidx] "+v"(ck_index)' if CKPT else ''),
           '    : [row0] "v"(tile_row_addr[0]), [row1] "v"(tile_row_addr[1]), [tr0] "v"(tile_tr_addr[0]), [tr1] "v"(tile_tr_addr[1]),',
           '      [lanebase] "v"(ring_lane_addr), [cap] "v"(cap), [slaboff] "v"(slab_off),',
           '      [tbl] "s"(table_addr_biased), [P] "s"(P), [c3f00] "s"(0x3f00u), [wbase] "s"(words_base),',
           '      [sbase] "s"(symbols_base), [ntiles] "s"(n_tiles),' + (' [tilestep] "s"(tile_step_bytes),' if symbol_major else '') +
           (' [ckpos] "s"(ck_pos_base), [cklower] "s"(ck_lower_base), [ckrange] "s"(ck_range_base), [cktiles] "s"(ck_tiles),' if CKPT else ''),
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
    if symbol_major:
        header[1] = header[1].replace(": see", ", symbols[t][stream]: see")
    out = OUT[(flushes, symbol_major)]
    if CKPT:
        out = out.with_name(out.name.replace(".inc", "_ck_n8.inc" if N8 else "_ck.inc"))
        if N8:
            header[1] = header[1].replace(": see", ", int8 symbol matrix: see")
    out.write_text(a.render(header, ops))
    print(f"wrote {out} ({a.n_instr()} instructions incl. prologue)")
    for n in notes:
        print("  note:", n)


def main():
    for sm in ((False,) if CKPT else (False, True)):
        emit(1, sm)
        emit(2, sm)


if __name__ == "__main__":
    main()
