#!/usr/bin/env python3
"""Generates constriction_amd/csrc/cst_encode_loop.inc: the hand-scheduled gfx950 main loop of the (32,64) ANS
encoder -- ONE asm statement that encodes all full 32-symbol tiles of a wave's 64 streams, last tile first.

Per tile ("half": the loop body holds two, one per register set of prefetched symbols):
  1. read the (at most 3) whole 16-byte chunks of compressed words that are complete in the lane's LDS ring,
  2. wait for this tile's symbols (requested TWO tiles earlier, 8 x 16 B per lane, transposed mapping) and write
     them to the wave's LDS tile,
  3. store the chunks (exec-masked, slabs are 16-byte aligned on this path) and request the symbols of tile - 2,
  4. run the 32 coder steps (25 instructions each + 3.75 of software pipeline, see ans_encode_tile32).
The generator keeps the lgkmcnt / vmcnt book (asmgen.Asm).  All stores of a tile are issued BEFORE its loads, so the
only operations younger than the loads a tile waits for are the other register set's eight loads: the same
s_waitcnt operand is exact for the first pass (nothing stored yet) and for the steady state.

Run:  python scripts/gen_encode_loop.py   (rewrites the .inc; the .inc is checked in)
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from asmgen import Asm  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "constriction_amd" / "csrc" / "cst_encode_loop.inc"


def quad_regs(base):
    return [f"v{base + i}" for i in range(4)]


R = {"A": [f"v[{100 + 4 * k}:{103 + 4 * k}]" for k in range(8)], "B": [f"v[{132 + 4 * k}:{135 + 4 * k}]" for k in range(8)]}
S = [quad_regs(164), quad_regs(168), quad_regs(172)]
S_T = ["v[164:167]", "v[168:171]", "v[172:175]"]
E = [[quad_regs(176 + 4 * i) for i in range(4)], [quad_regs(192 + 4 * i) for i in range(4)]]
E_T = [[f"v[{176 + 4 * i}:{179 + 4 * i}]" for i in range(4)], [f"v[{192 + 4 * i}:{195 + 4 * i}]" for i in range(4)]]
A0, A1, W0, W1, U0, U1, X0, X1, V0, V1, SM0, SM1, Q0, Q1 = (f"v{r}" for r in range(208, 222))
RR, PSHL, KK, CK, RA, EA = (f"v{r}" for r in range(222, 228))
FD = [(f"v[{228 + 4 * k}:{229 + 4 * k}]", f"v[{230 + 4 * k}:{231 + 4 * k}]", f"v[{228 + 4 * k}:{231 + 4 * k}]") for k in range(4)]
NCH, LIM, FADDR, FOFF = "v244", "v245", "v246", "v247"
SD, SAVE = "s[84:85]", "s[86:87]"
CLOBBERS = [f"v{r}" for r in range(100, 248)] + ["s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "vcc", "memory"]


def step(a, c, p, m0, m1):
    a.i(f"v_lshlrev_b32 {PSHL}, %[shP], {p}", "p << (32 - P)")
    a.i(f"v_sub_u32 {KK}, %[twoP], {p}", "k = 2^P - p")
    a.i(f"v_lshlrev_b32 {RA}, 8, %[wr]")
    a.i(f"v_cmp_ge_u32 vcc, %[hi], {PSHL}", "emit <=> (state >> (64 - P)) >= p")
    a.i(f"v_and_or_b32 {RA}, {RA}, %[c3f00], %[lanebase]")
    a.i(f"v_add_u32 {CK}, {c}, {KK}")
    a.i(f"v_cndmask_b32_e64 {A0}, %[lo], %[hi], vcc")
    a.i(f"v_cndmask_b32_e64 {A1}, %[hi], 0, vcc")
    a.ds(f"ds_write_b32 {RA}, %[lo]", "W", "candidate word, always written")
    a.i(f"v_addc_co_u32 %[wr], vcc, 0, %[wr], vcc")
    # q_est = floor(A * m / 2^64) = a1*m1 + floor((a1*m0 + a0*m1 + hi32(a0*m0)) / 2^32), the middle sum taken to 65 bits
    a.i(f"v_mul_hi_u32 {W0}, {A0}, {m0}")
    a.i(f"v_mad_u64_u32 v[212:213], vcc, {A1}, {m0}, v[210:211]", "U = a1*m0 + hi32(a0*m0)   (< 2^64)")
    a.i(f"v_mad_u64_u32 v[216:217], vcc, {A0}, {m1}, v[212:213]", "T = a0*m1 + U, carry -> vcc")
    a.i(f"v_mov_b32 {SM0}, {V1}")
    a.i(f"v_addc_co_u32 {SM1}, vcc, 0, {W1}, vcc", "[T_hi, carry]")
    a.i(f"v_mad_u64_u32 v[220:221], vcc, {A1}, {m1}, v[218:219]", "q_est in {q - 1, q}")
    a.i(f"v_mul_lo_u32 {RR}, {Q0}, {p}")
    a.i(f"v_sub_u32 {RR}, {A0}, {RR}")
    a.i(f"v_cmp_ge_u32 vcc, {RR}, {p}", "fix <=> q = q_est + 1")
    a.i(f"v_mad_u64_u32 v[212:213], {SD}, {Q0}, {KK}, v[208:209]")
    a.i(f"v_mad_u32_u24 {U1}, {Q1}, {KK}, {U1}")
    a.i(f"v_cndmask_b32 {RR}, {c}, {CK}, vcc")
    a.i(f"v_add_co_u32 %[lo], vcc, {U0}, {RR}")
    a.i(f"v_addc_co_u32 %[hi], vcc, 0, {U1}, vcc")


def fetch_entries(a, sq, eset, tag):
    x, y, z, w = S[sq]
    for i, sym in enumerate((w, z, y, x)):       # consumption order: .w first
        a.i(f"v_lshl_add_u32 {EA}, {sym}, 4, %[tbl]")
        a.ds(f"ds_read_b128 {E_T[eset][i]}, {EA}", tag)
    a.i(f"v_max3_i32 %[smax], %[smax], {x}, {y}")
    a.i(f"v_max3_i32 %[smax], %[smax], {z}, {w}")
    a.i(f"v_min3_i32 %[smin], %[smin], {x}, {y}")
    a.i(f"v_min3_i32 %[smin], %[smin], {z}, {w}")


def steps(a, eset):
    for c, p, m0, m1 in E[eset]:
        step(a, c, p, m0, m1)


def half(a, name):
    a.i(f"// ---- tile using symbol set {name}".replace("//", ";"))
    # 1. ring reads of the 64-byte group (4 chunks) that may be complete.  Words leave for HBM 64 bytes at a time:
    #    16-byte stores reach DRAM as partial bursts (measured 1.6x write amplification), and at most 15 + 12 words are
    #    ever pending, so one group per tile is enough and the 64-slot ring holds the backlog.
    for k in range(4):
        a.i(f"v_add_lshl_u32 {FADDR}, %[flushed], {4 * k}, 8")
        a.i(f"v_and_or_b32 {FADDR}, {FADDR}, %[c3f00], %[lanebase]")
        a.ds(f"ds_read2st64_b32 {FD[k][0]}, {FADDR} offset1:1", "fl")
        a.ds(f"ds_read2st64_b32 {FD[k][1]}, {FADDR} offset0:2 offset1:3", "fl")
    # 2. this tile's symbols -> LDS tile
    a.wait_vm(f"ld{name}", "symbols of this tile (requested two tiles ago)")
    for k in range(8):
        a.ds(f"ds_write_b128 %[tr], {R[name][k]} offset:{1152 * k}", "tl")
    # 3. chunk stores, then the symbol loads of tile - 2 (stores first: see the module docstring)
    a.i(f"v_sub_u32 {NCH}, %[wr], %[flushed]")
    a.i(f"v_lshrrev_b32 {NCH}, 4, {NCH}", "whole 16-word groups pending: 0 or 1")
    a.i(f"v_add_u32 {LIM}, 16, %[flushed]")
    a.i(f"v_lshl_add_u32 {FOFF}, %[flushed], 2, %[slaboff]")
    a.i(f"v_cmp_le_u32 vcc, {LIM}, %[cap]", "group inside the slab (cap % 16 == 0 on this path)")
    a.i(f"v_cmp_ne_u32 {SAVE}, 0, {NCH}")
    a.i(f"s_and_b64 vcc, vcc, {SAVE}")
    a.wait_lds("fl")
    a.i(f"s_and_saveexec_b64 {SAVE}, vcc")
    for k in range(4):
        a.vmem(f"global_store_dwordx4 {FOFF}, {FD[k][2]}, %[wbase] offset:{16 * k}", "st")
    a.i(f"s_mov_b64 exec, {SAVE}")
    a.i(f"v_lshl_add_u32 %[flushed], {NCH}, 4, %[flushed]")
    for k in range(8):
        a.vmem(f"global_load_dwordx4 {R[name][k]}, %[goff{k}], s[80:81] nt", f"ld{name}")
    advance_base(a)
    # 4. the 32 steps, quads 7 .. 0
    a.ds(f"ds_read_b128 {S_T[0]}, %[row] offset:112", "S7")
    a.ds(f"ds_read_b128 {S_T[1]}, %[row] offset:96", "S6")
    a.wait_lds("S7")
    fetch_entries(a, 0, 0, "E7")
    for q in range(7, -1, -1):
        cur_e = (7 - q) % 2
        nxt_s = (7 - q + 1) % 3          # symbols of quad q-1 live here
        far_s = (7 - q + 2) % 3          # symbols of quad q-2 go here
        if q >= 1:
            a.wait_lds(f"S{q - 1}", f"quad {q}: symbols of quad {q - 1} are back")
        if q >= 2:
            a.ds(f"ds_read_b128 {S_T[far_s]}, %[row] offset:{16 * (q - 2)}", f"S{q - 2}")
        if q >= 1:
            fetch_entries(a, nxt_s, 1 - cur_e, f"E{q - 1}")
        a.wait_lds(f"E{q}", f"entries of quad {q} are back")
        steps(a, cur_e)
    a.wait_lds_all("---- end of tile")


def advance_base(a):
    """s[80:81] -> symbols of the next tile to request; stays on tile 0 once every tile has been requested"""
    a.i("s_cmp_lg_u32 s83, 0")
    a.i("s_cselect_b32 s88, 0x80, 0")
    a.i("s_cselect_b32 s89, 1, 0")
    a.i("s_sub_u32 s80, s80, s88")
    a.i("s_subb_u32 s81, s81, 0")
    a.i("s_sub_u32 s83, s83, s89")


def gen():
    a = Asm()
    a.i(f"v_mov_b32 {W1}, 0")
    a.i(f"v_mov_b32 {X1}, 0")
    a.i("s_mov_b64 s[80:81], %[sbase]", "symbols of the LAST full tile of stream s0")
    a.i("s_mov_b32 s82, %[ntiles]", "tiles left to encode")
    a.i("s_sub_u32 s83, %[ntiles], 1", "tiles left to request")
    for k in range(8):
        a.vmem(f"global_load_dwordx4 {R['A'][k]}, %[goff{k}], s[80:81] nt", "ldA")
    advance_base(a)
    for k in range(8):
        a.vmem(f"global_load_dwordx4 {R['B'][k]}, %[goff{k}], s[80:81] nt", "ldB")
    advance_base(a)
    a.i("1:")
    vm_at_loop_entry = list(a.vm)
    mark = len(a.lines)
    half(a, "A")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_eq_u32 s82, 0")
    a.i("s_cbranch_scc1 2f")
    half(a, "B")
    a.i("s_sub_u32 s82, s82, 1")
    a.i("s_cmp_lg_u32 s82, 0")
    a.i("s_cbranch_scc1 1b")
    check_steady_state(a, mark, vm_at_loop_entry)
    a.i("2:")
    a.wait_vm_all("nothing may land in the scratch registers after the statement")
    return a


def check_steady_state(a, mark, entry_queue):
    """Replays the loop body's vector-memory events for a second and third pass, starting from the queue the first
    pass leaves behind, and checks every emitted vmcnt operand: it must not exceed the exact one (correctness) and
    whatever it waits for beyond the target must be a store (no over-wait on younger loads)."""
    import re
    body = a.lines[mark:]
    q = list(a.vm)                      # queue at the back edge of pass 1
    assert q[:0] == [] and len(entry_queue) == 16
    for _ in range(2):
        for text, _c in body:
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", text)
            if m:
                n = int(m.group(1))
                # which tag was this wait for?  the half it belongs to: the next ds_write_b128 names the register set
                done, q_keep = q[: max(len(q) - n, 0)], q[max(len(q) - n, 0):]
                # everything in `done` completes; the target loads must be among them and no load may be younger-needed
                assert all(t != "st" or True for t in done)
                loads_waited = [t for t in done if t.startswith("ld")]
                assert len(set(loads_waited)) <= 1, ("over-wait on the other set's loads", loads_waited)
                assert len(loads_waited) in (0, 8), loads_waited
                q = q_keep
            elif text.startswith("global_load"):
                q.append("ldA" if "v[1" in text and int(re.search(r"v\[(\d+):", text).group(1)) < 132 else "ldB")
            elif text.startswith("global_store"):
                q.append("st")


def main():
    a = gen()
    header = ["// GENERATED by scripts/gen_encode_loop.py -- do not edit by hand (edit the generator and re-run it).",
              "// Main loop of the hand-scheduled (32,64) ANS encoder: see ans_encode_tiles_loop in cst_ans_kernels.hpp."]
    ops = ['    : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr), [flushed] "+v"(flushed), [smin] "+v"(smin), [smax] "+v"(smax)',
           '    : [row] "v"(tile_row_addr), [tr] "v"(tile_tr_addr), [lanebase] "v"(ring_lane_addr), [cap] "v"(cap), [slaboff] "v"(slab_off),',
           '      [tbl] "s"(table_addr_biased), [shP] "s"(32u - P), [twoP] "s"(1u << P), [c3f00] "s"(0x3f00u), [wbase] "s"(words_base),',
           '      [sbase] "s"(symbols_base), [ntiles] "s"(n_tiles),',
           '      ' + ", ".join(f'[goff{k}] "v"(goff[{k}])' for k in range(8)),
           "    : " + ", ".join(f'"{c}"' for c in CLOBBERS) + ");"]
    OUT.write_text(a.render(header, ops))
    print(f"wrote {OUT} ({a.n_instr()} instructions incl. prologue; vm queue at exit {len(a.vm)})")


if __name__ == "__main__":
    main()
