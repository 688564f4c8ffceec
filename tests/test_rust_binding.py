"""CPU-only: the Rust side of the boundary (bindings/rust/) against the C header, WITHOUT a Rust toolchain (none exists in
the build image).  bindings/rust/src/ffi.rs is generated from include/constriction_amd.h (scripts/gen_rust_ffi.py); this
test parses BOTH files again with its own, independent parsers and compares every function's name, arity, parameter names
and types (usize <-> size_t, *const u32 <-> const uint32_t *, ...), every struct's fields and every enum constant -- the
check tests/test_abi.py makes for the ctypes binding.  The safe wrappers (src/lib.rs) are checked as far as a text can be:
every `ffi::cst_*` call names a declared function and passes as many arguments as it declares, every entry point that is
not a test hook is reachable from a wrapper, and the files are balanced."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "constriction_amd.h"
RUST = ROOT / "bindings" / "rust"

RUST_TO_C = {"i32": "int32_t", "u32": "uint32_t", "u16": "uint16_t", "i64": "int64_t", "u64": "uint64_t", "usize": "size_t", "f64": "double",
             "c_void": "void", "c_char": "char", "CstModel": "cst_model", "CstRangeState": "cst_range_state",
             "CstChainHeads": "cst_chain_heads", "CstCoderConfig": "cst_coder_config", "CstStatus": "cst_status", "CstLayout": "cst_layout"}


def c_canonical(t):
    """'const uint32_t *' -> ('uint32_t', ['const*']); pointer levels from the innermost outwards"""
    t = re.sub(r"\s+", " ", t.replace("*", " * ")).strip().split(" ")
    const = t[0] == "const"
    base = [x for x in t if x not in ("const", "*")]
    assert len(base) == 1, t
    levels = ["const*" if const else "mut*"] + ["mut*"] * (t.count("*") - 1) if "*" in t else []
    return base[0], levels


def rust_canonical(t):
    t = t.strip()
    levels = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        levels.append(m.group(1) + "*")
        t = m.group(2).strip()
    return RUST_TO_C[t], levels[::-1]


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    out = {}
    for m in re.finditer(r"([\w \*]+?)\b(cst_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args != "void":
            for a in args.split(","):
                mm = re.match(r"(.*?)(\w+)$", re.sub(r"\s+", " ", a).strip())
                params.append((mm.group(2), c_canonical(mm.group(1))))
        out[name] = (c_canonical(ret), params)
    return out


def rust_functions():
    text = re.sub(r"//[^\n]*", "", (RUST / "src" / "ffi.rs").read_text())
    block = re.search(r'extern "C" \{(.*)\}', text, re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        params = []
        for a in [x for x in args.split(",") if x.strip()]:
            pname, ptype = a.split(":", 1)
            params.append((pname.strip().replace("r#", ""), rust_canonical(ptype)))
        out[name] = (rust_canonical(ret) if ret else ("void", []), params)
    return out


def test_generated_file_is_current():
    assert subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_rust_ffi.py"), "--check"]).returncode == 0, \
        "bindings/rust/src/ffi.rs is stale: run scripts/gen_rust_ffi.py"


def test_extern_block_matches_the_header():
    c, r = header_functions(), rust_functions()
    assert len(c) >= 60 and sorted(c) == sorted(r)
    for name in c:
        (cret, cparams), (rret, rparams) = c[name], r[name]
        # enums are ints on both sides: cst_status / cst_layout aliases compare as themselves, plain int32_t as int32_t
        assert cret == rret, (name, cret, rret)
        assert len(cparams) == len(rparams), name
        for (cn, ct), (rn, rt) in zip(cparams, rparams):
            assert cn == rn and ct == rt, (name, cn, ct, rn, rt)


def test_structs_enums_and_version_match():
    h = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    ffi = (RUST / "src" / "ffi.rs").read_text()
    for cname, rname in (("cst_coder_config", "CstCoderConfig"), ("cst_range_state", "CstRangeState"), ("cst_chain_heads", "CstChainHeads")):
        cfields = re.findall(r"(\w+)\s+(\w+);", re.search(r"typedef struct %s \{(.*?)\}" % cname, h, re.S).group(1))
        rbody = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive[^\]]*\]\s*)?pub struct %s \{(.*?)\}" % rname, ffi, re.S).group(1)
        rfields = re.findall(r"pub (\w+): (\w+),", rbody)
        assert [(f, t) for t, f in cfields] == [(f, RUST_TO_C[t]) for f, t in rfields], cname
    for const, value in re.findall(r"\b(CST_[A-Z_]+)\s*=\s*(-?\d+)", h):
        assert re.search(r"pub const %s: \w+ = %s;" % (const, value), ffi), const
    version = re.search(r"#define CST_ABI_VERSION (\d+)", h).group(1)
    assert f"pub const CST_ABI_VERSION: i32 = {version};" in ffi
    for flag, value in re.findall(r"#define (CST_FLAG_\w+) (\d+)u", h):
        assert f"pub const {flag}: u32 = {value};" in ffi


def _split_args(s):
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    return args


def test_wrappers_call_declared_functions_with_the_declared_arity():
    decl = rust_functions()
    text = re.sub(r"//[^\n]*", "", (RUST / "src" / "lib.rs").read_text())
    called = set()
    for m in re.finditer(r"ffi::(cst_\w+)\s*\(", text):
        name, i, depth = m.group(1), m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        args = _split_args(text[m.end(): i - 1])
        assert name in decl, name
        assert len(args) == len(decl[name][1]), (name, len(args), len(decl[name][1]))
        called.add(name)
    # every entry point that is not a test hook or plain introspection has a wrapper
    unwrapped = {n for n in decl if n not in called and not n.startswith("cst_debug_")}
    assert unwrapped <= {"cst_model_copy_cdfs", "cst_family_cdf_rows", "cst_ans_count_until", "cst_ans_encode_ragged_ordered",
                         "cst_ans_decode_ragged_ordered", "cst_ans_count_until_ordered"}, unwrapped
    for must in ("cst_ans_encode_gaussian_batch", "cst_ans_decode_gaussian_batch", "cst_range_encode_batch", "cst_range_decode_batch",
                 "cst_range_encode_gaussian_batch", "cst_range_decode_gaussian_batch", "cst_chain_decode_gaussian_batch",
                 "cst_scatter_rccl", "cst_gather_rccl", "cst_gather_sizes_rccl"):
        assert must in called, must


def test_wrapper_names_follow_the_reference_and_files_are_balanced():
    lib = (RUST / "src" / "lib.rs").read_text()
    for method in ("encode_iid_symbols_reverse", "encode_symbols_reverse", "decode_iid_symbols", "decode_symbols",       # stream::stack::AnsCoder
                   "encode_iid_symbols", "encode_symbols", "max_words"):                                                  # stream::queue::RangeEncoder
        assert re.search(r"pub fn %s\b" % method, lib), method
    for f in ("lib.rs", "hip.rs", "ffi.rs"):
        text = re.sub(r"//[^\n]*", "", (RUST / "src" / f).read_text())
        text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
        text = re.sub(r"'(?:[^'\\]|\\.)'", "' '", text)
        for a, b in ("()", "[]", "{}"):
            assert text.count(a) == text.count(b), (f, a, text.count(a), text.count(b))
    assert (RUST / "Cargo.toml").exists() and "links = \"constriction_amd\"" in (RUST / "Cargo.toml").read_text()
    assert "rustc-link-lib=dylib=constriction_amd" in (RUST / "build.rs").read_text()
