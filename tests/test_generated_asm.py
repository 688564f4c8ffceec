"""The generated main-loop statements (constriction_amd/csrc/cst_{encode,decode}_loop.inc) must be exactly what
scripts/gen_{encode,decode}_loop.py emit: the generators keep the s_waitcnt book, a hand edit of the .inc would not."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _load(name):
    sys.path.insert(0, str(ROOT / "scripts"))
    spec = importlib.util.spec_from_file_location(name, ROOT / "scripts" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _regenerate(mod, tmp_path, name):
    mod.OUT = tmp_path / name
    if hasattr(mod, "OUT_SINGLE"):          # (never rewrite a checked-in file: its mtime triggers a rebuild of the library)
        mod.OUT_SINGLE = tmp_path / ("single_" + name)
    for attr, prefix in (("OUT_SM", "sm_"), ("OUT_PLAIN", "plain_"), ("OUT_SM_PLAIN", "sm_plain_")):
        if hasattr(mod, attr):
            setattr(mod, attr, tmp_path / (prefix + name))
    mod.main()
    return (tmp_path / name).read_text()


def test_encode_loop_is_in_sync(tmp_path, monkeypatch):
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT", "GEN_NO_STORE", "GEN_NO_LOAD"):
        monkeypatch.delenv(var, raising=False)
    text = _regenerate(_load("gen_encode_loop"), tmp_path, "cst_encode_loop.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop.inc").read_text()


def test_decode_loop_is_in_sync(tmp_path, monkeypatch):
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT", "GEN_NO_STORE", "GEN_NO_LOAD"):
        monkeypatch.delenv(var, raising=False)
    text = _regenerate(_load("gen_decode_loop"), tmp_path, "cst_decode_loop.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop.inc").read_text()
    # ... and its symbol-major variant
    assert (tmp_path / "sm_cst_decode_loop.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_sm.inc").read_text()


def test_small_footprint_loops_are_in_sync(tmp_path, monkeypatch):
    """the loops of cst_ans_small.hip (batches of more than one wave per SIMD)"""
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT", "GEN_NO_STORE", "GEN_NO_LOAD"):
        monkeypatch.delenv(var, raising=False)
    mod = _load("gen_encode_loop")
    mod.OUT, mod.OUT_SINGLE, mod.OUT_SM = tmp_path / "a.inc", tmp_path / "b.inc", tmp_path / "c.inc"
    mod.main()
    assert (tmp_path / "b.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_1buf.inc").read_text()
    # ... and the symbol-major staging of the same loop
    assert (tmp_path / "c.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_sm.inc").read_text()
    text = _regenerate(_load("gen_decode_loop_small"), tmp_path, "cst_decode_loop_small.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_small.inc").read_text()
    monkeypatch.setenv("GEN_SMALL_N8", "1")                   # round 5: the same loop over byte tiles (int8 matrices)
    text = _regenerate(_load("gen_decode_loop_small"), tmp_path, "cst_decode_loop_small_n8.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_small_n8.inc").read_text()
    monkeypatch.setenv("GEN_SMALL_N8", "16")                  # ... and int16
    text = _regenerate(_load("gen_decode_loop_small"), tmp_path, "cst_decode_loop_small_n16.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_small_n16.inc").read_text()


def test_pt_loops_are_in_sync(tmp_path, monkeypatch):
    """the per-stream-table (C3) main loops"""
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT", "GEN_NO_STORE", "GEN_NO_LOAD"):
        monkeypatch.delenv(var, raising=False)
    for name in ("pt_encode", "pt_decode"):
        text = _regenerate(_load(f"gen_{name}_loop"), tmp_path, f"cst_{name}_loop.inc")
        assert text == (ROOT / "constriction_amd" / "csrc" / f"cst_{name}_loop.inc").read_text()


def test_range_loops_are_in_sync(tmp_path, monkeypatch):
    """the range encoder's main loops (one / two word groups per tile, both symbol layouts)"""
    monkeypatch.delenv("GEN_NO_LGKM", raising=False)
    mod = _load("gen_range_encode_loop")
    names = {key: path.name for key, path in mod.OUT.items()}
    mod.OUT = {key: tmp_path / name for key, name in names.items()}
    mod.main()
    assert len(names) == 4
    for name in names.values():
        assert (tmp_path / name).read_text() == (ROOT / "constriction_amd" / "csrc" / name).read_text(), name


def test_range_decode_loops_are_in_sync(tmp_path, monkeypatch):
    """the range decoder's main loops (plain / end-of-data aware, P <= 12 / bucket entries, both symbol layouts)"""
    monkeypatch.delenv("GEN_NO_LGKM", raising=False)
    mod = _load("gen_range_decode_loop")
    names = {key: path.name for key, path in mod.OUT.items()}
    mod.OUT = {key: tmp_path / name for key, name in names.items()}
    mod.main()
    assert len(names) == 4
    for name in names.values():
        assert (tmp_path / name).read_text() == (ROOT / "constriction_amd" / "csrc" / name).read_text(), name
        sm = name.replace(".inc", "_sm.inc")
        assert (tmp_path / sm).read_text() == (ROOT / "constriction_amd" / "csrc" / sm).read_text(), sm


def test_wide_precision_loops_are_in_sync(tmp_path, monkeypatch):
    """the ANS coder's main loops for 12 < P <= 24 (bucket entries / unpacked table entries)"""
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT"):
        monkeypatch.delenv(var, raising=False)
    text = _regenerate(_load("gen_decode_loop_b16"), tmp_path, "cst_decode_loop_b16.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_b16.inc").read_text()
    assert (tmp_path / "sm_cst_decode_loop_b16.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_b16_sm.inc").read_text()
    for mode, name in (("1", "cst_decode_loop_b16_n8.inc"), ("2", "cst_decode_loop_b16_n16.inc")):      # round 5: int8 / int16 matrices
        monkeypatch.setenv("GEN_B16_NARROW", mode)
        assert _regenerate(_load("gen_decode_loop_b16"), tmp_path, name) == (ROOT / "constriction_amd" / "csrc" / name).read_text()
    monkeypatch.setenv("GEN_B16_SMALL", "1")                  # ... and the small-footprint forms (two waves per SIMD), int32 too
    for mode, name in (("1", "cst_decode_loop_b16_s8.inc"), ("2", "cst_decode_loop_b16_s16.inc"), ("4", "cst_decode_loop_b16_s32.inc")):
        monkeypatch.setenv("GEN_B16_NARROW", mode)
        assert _regenerate(_load("gen_decode_loop_b16"), tmp_path, name) == (ROOT / "constriction_amd" / "csrc" / name).read_text()
    monkeypatch.delenv("GEN_B16_NARROW")
    monkeypatch.delenv("GEN_B16_SMALL")
    text = _regenerate(_load("gen_encode_loop_wide"), tmp_path, "cst_encode_loop_wide.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_wide.inc").read_text()
    assert (tmp_path / "sm_cst_encode_loop_wide.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_wide_sm.inc").read_text()


def test_w16_loops_are_in_sync(tmp_path, monkeypatch):
    """the (16,32) ANS coder's main loops"""
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT"):
        monkeypatch.delenv(var, raising=False)
    text = _regenerate(_load("gen_decode_loop_w16"), tmp_path, "cst_decode_loop_w16.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_w16.inc").read_text()
    assert (tmp_path / "sm_cst_decode_loop_w16.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_w16_sm.inc").read_text()
    text = _regenerate(_load("gen_encode_loop_w16"), tmp_path, "cst_encode_loop_w16.inc")
    assert text == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_w16.inc").read_text()
    assert (tmp_path / "sm_cst_encode_loop_w16.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_w16_sm.inc").read_text()


def test_wait_bookkeeping_rejects_unreachable_counts():
    """asmgen refuses a wait whose operand would exceed what the hardware counter can express."""
    asmgen = _load("asmgen")
    a = asmgen.Asm()
    a.ds("ds_read_b32 v0, v1", "first")
    for k in range(16):
        a.ds(f"ds_read_b32 v{2 + k}, v1", "later")
    try:
        a.wait_lds("first")
    except AssertionError:
        return
    raise AssertionError("lgkmcnt operand > 15 must be rejected")


def test_producer_consumer_loops_are_in_sync(tmp_path, monkeypatch):
    """the coder and helper halves of the producer / consumer encoder (cst_ans_pc.hip)"""
    for var in ("GEN_NO_BARRIER", "GEN_PRIO", "GEN_HABL", "GEN_NSETS", "GEN_PAIRS", "GEN_LSETS", "GEN_HLOAD_MOD", "GEN_HSTORE_MOD"):
        monkeypatch.delenv(var, raising=False)
    mod = _load("gen_encode_loop_pc")
    mod.OUT, mod.OUT_HELPER = tmp_path / "coder.inc", tmp_path / "helper.inc"
    mod.OUT_LOADER, mod.OUT_STORER = tmp_path / "loader.inc", tmp_path / "storer.inc"
    mod.OUT_N8, mod.OUT_LOADER_N8 = tmp_path / "n8.inc", tmp_path / "loader_n8.inc"      # round 5: the int8 forms of coder and loader
    mod.OUT_CK, mod.OUT_N8_CK = tmp_path / "ck.inc", tmp_path / "n8_ck.inc"              # ... and the coders that note jump points
    mod.OUT_N16, mod.OUT_N16_CK, mod.OUT_LOADER_N16 = tmp_path / "n16.inc", tmp_path / "n16_ck.inc", tmp_path / "loader_n16.inc"      # int16
    mod.OUT_N8W, mod.OUT_N8W_CK, mod.OUT_N16W, mod.OUT_N16W_CK = (tmp_path / f"{n}.inc" for n in ("n8w", "n8w_ck", "n16w", "n16w_ck"))    # 12 < P <= 24
    mod.OUT_STORER2 = tmp_path / "storer2.inc"
    mod.main_all()
    for name in ("loader", "storer", "storer2", "n8", "loader_n8", "ck", "n8_ck", "n16", "n16_ck", "loader_n16", "n8w", "n8w_ck", "n16w", "n16w_ck"):
        assert (tmp_path / f"{name}.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / f"cst_encode_loop_pc_{name}.inc").read_text()
    assert (tmp_path / "coder.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_pc.inc").read_text()
    assert (tmp_path / "helper.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_encode_loop_pc_helper.inc").read_text()


def test_lane_quad_decoder_loop_is_in_sync(tmp_path, monkeypatch):
    """the main loop of the opt-in lane-quad decoder (cst_ans_dq.hip)"""
    for var in ("GEN_NO_STORE", "GEN_NO_LOAD", "GEN_DQ_SPREAD"):
        monkeypatch.delenv(var, raising=False)
    mod = _load("gen_decode_loop_dq")
    mod.OUT = tmp_path / "dq.inc"
    mod.main()
    assert (tmp_path / "dq.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_dq.inc").read_text()


def test_int8_decoder_loop_is_in_sync(tmp_path, monkeypatch):
    """the main loop of the decoder that writes int8 matrices itself (cst_ans_n8.hip)"""
    monkeypatch.delenv("GEN_NO_STORE", raising=False)
    mod = _load("gen_decode_loop_n8")
    mod.OUT = tmp_path / "n8.inc"
    mod.main()
    assert (tmp_path / "n8.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_n8.inc").read_text()
    monkeypatch.setenv("GEN_N16", "1")                        # ... and int16 matrices (two tiles per pass)
    mod = _load("gen_decode_loop_n8")
    mod.OUT = tmp_path / "n16.inc"
    mod.main()
    assert (tmp_path / "n16.inc").read_text() == (ROOT / "constriction_amd" / "csrc" / "cst_decode_loop_n16.inc").read_text()


def test_jump_point_and_sub_lane_variants_are_in_sync(tmp_path, monkeypatch):
    """round 5: the checkpointing encoders (GEN_PT_CK, GEN_RANGE_CK) and the byte-tile loops of the sub-lane decoders
    (GEN_PT_SUB, GEN_RANGE_SUB) are the same generators run with a knob"""
    csrc = ROOT / "constriction_amd" / "csrc"
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT", "GEN_PT_WINDOW", "GEN_NO_MORE"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("GEN_PT_SUB", "1")
    assert _regenerate(_load("gen_pt_decode_loop"), tmp_path, "cst_pt_decode_loop_sub.inc") == (csrc / "cst_pt_decode_loop_sub.inc").read_text()
    monkeypatch.setenv("GEN_PT_SUB", "2")                     # 16-slot rings, a window every half tile: four waves per SIMD
    assert _regenerate(_load("gen_pt_decode_loop"), tmp_path, "cst_pt_decode_loop_sub16.inc") == (csrc / "cst_pt_decode_loop_sub16.inc").read_text()
    monkeypatch.delenv("GEN_PT_SUB")
    monkeypatch.setenv("GEN_PT_CK", "1")
    assert _regenerate(_load("gen_pt_encode_loop"), tmp_path, "cst_pt_encode_loop_ck.inc") == (csrc / "cst_pt_encode_loop_ck.inc").read_text()
    monkeypatch.delenv("GEN_PT_CK")
    monkeypatch.setenv("GEN_RANGE_CK", "1")
    mod = _load("gen_range_encode_loop")
    mod.OUT = {key: tmp_path / path.name for key, path in mod.OUT.items()}
    mod.main()
    for name in ("cst_range_encode_loop_ck.inc", "cst_range_encode_loop_2f_ck.inc"):
        assert (tmp_path / name).read_text() == (csrc / name).read_text(), name
    monkeypatch.delenv("GEN_RANGE_CK")
    monkeypatch.setenv("GEN_RANGE_SUB", "1")
    mod = _load("gen_range_decode_loop")
    mod.OUT = {key: tmp_path / path.name for key, path in mod.OUT.items()}
    mod.main()
    for name in ("cst_range_decode_loop_sub.inc", "cst_range_decode_loop_sub_ends.inc", "cst_range_decode_loop_b16_sub.inc",
                 "cst_range_decode_loop_b16_sub_ends.inc"):
        assert (tmp_path / name).read_text() == (csrc / name).read_text(), name
    # round 6: the per-stream-table generators over INT8 symbol matrices (GEN_PT_N8 on top of GEN_PT_CK / GEN_PT_SUB)
    monkeypatch.delenv("GEN_RANGE_SUB")
    monkeypatch.setenv("GEN_PT_N8", "1")
    monkeypatch.setenv("GEN_PT_CK", "1")
    assert _regenerate(_load("gen_pt_encode_loop"), tmp_path, "cst_pt_encode_loop_ck_n8.inc") == (csrc / "cst_pt_encode_loop_ck_n8.inc").read_text()
    monkeypatch.delenv("GEN_PT_CK")
    for mode, name in (("1", "cst_pt_decode_loop_sub_n8.inc"), ("2", "cst_pt_decode_loop_sub16_n8.inc")):
        monkeypatch.setenv("GEN_PT_SUB", mode)
        assert _regenerate(_load("gen_pt_decode_loop"), tmp_path, name) == (csrc / name).read_text(), name
    monkeypatch.delenv("GEN_PT_SUB")
    monkeypatch.delenv("GEN_PT_N8")
    monkeypatch.setenv("GEN_RANGE_SUB", "1")
    # round 6: the same two generators over INT8 symbol matrices (GEN_RANGE_N8 on top of GEN_RANGE_SUB / GEN_RANGE_CK)
    monkeypatch.setenv("GEN_RANGE_N8", "1")
    mod = _load("gen_range_decode_loop")
    mod.OUT = {key: tmp_path / path.name for key, path in mod.OUT.items()}
    mod.main()
    for name in ("cst_range_decode_loop_sub_n8.inc", "cst_range_decode_loop_sub_n8_ends.inc", "cst_range_decode_loop_b16_sub_n8.inc",
                 "cst_range_decode_loop_b16_sub_n8_ends.inc"):
        assert (tmp_path / name).read_text() == (csrc / name).read_text(), name
    monkeypatch.delenv("GEN_RANGE_SUB")
    monkeypatch.setenv("GEN_RANGE_CK", "1")
    mod = _load("gen_range_encode_loop")
    mod.OUT = {key: tmp_path / path.name for key, path in mod.OUT.items()}
    mod.main()
    for name in ("cst_range_encode_loop_ck_n8.inc", "cst_range_encode_loop_2f_ck_n8.inc"):
        assert (tmp_path / name).read_text() == (csrc / name).read_text(), name


def test_packed_w16_loops_are_in_sync(tmp_path, monkeypatch):
    """CST_FLAG_PACKED_W16: the (16,32) loops with two words per slot (GEN_W16_PACKED)"""
    csrc = ROOT / "constriction_amd" / "csrc"
    for var in ("GEN_NO_LGKM", "GEN_NO_VMWAIT"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("GEN_W16_PACKED", "1")
    assert _regenerate(_load("gen_decode_loop_w16"), tmp_path, "cst_decode_loop_w16_pk.inc") == (csrc / "cst_decode_loop_w16_pk.inc").read_text()
    assert _regenerate(_load("gen_encode_loop_w16"), tmp_path, "cst_encode_loop_w16_pk.inc") == (csrc / "cst_encode_loop_w16_pk.inc").read_text()
    monkeypatch.setenv("GEN_W16_CK", "1")                 # round 6: the packed encoder noting jump points on its way
    assert _regenerate(_load("gen_encode_loop_w16"), tmp_path, "cst_encode_loop_w16_pk_ck.inc") == (csrc / "cst_encode_loop_w16_pk_ck.inc").read_text()


def test_every_loop_head_is_pinned_to_a_cache_line():
    """code placement is worth +-10 % on these loops (profiles/r04_placement.txt): every generated statement carries
    `.p2align 6` directly in front of its loop head, so that an edit upstream of a loop cannot move it (asmgen.py)"""
    incs = sorted((ROOT / "constriction_amd" / "csrc").glob("*.inc"))
    statements = [p for p in incs if "asm volatile(" in p.read_text()]
    assert len(statements) >= 40
    for p in statements:
        lines = [ln.split("//")[0].strip() for ln in p.read_text().splitlines()]
        heads = [i for i, ln in enumerate(lines) if ln == '"1:\\n"']
        assert len(heads) == 1, p.name
        assert lines[heads[0] - 1] == '".p2align 6\\n\\t"', p.name


def test_every_statement_declares_scc_clobbered():
    """the generated statements count tiles and compare on the scalar side (s_sub_u32 / s_cmp_* / s_add_u32 ...: all write SCC).  Until
    round 5 none declared it, and one harmless edit made the compiler keep a flag in SCC ACROSS a statement"""
    incs = sorted((ROOT / "constriction_amd" / "csrc").glob("*.inc"))
    statements = [p for p in incs if "asm volatile(" in p.read_text()]
    assert len(statements) >= 40
    for p in statements:
        text = p.read_text()
        clobbers = text[text.rindex("    : "):]
        assert '"scc"' in clobbers and '"vcc"' in clobbers and '"memory"' in clobbers, p.name
