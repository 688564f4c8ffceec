#!/usr/bin/env python3
"""Writes bindings/rust/src/ffi.rs -- the raw `extern "C"` declarations of EVERY entry point of include/constriction_amd.h --
from the header itself (no Rust toolchain exists in the build image, so nothing can run bindgen; tests/test_rust_binding.py
parses the result again, independently, and compares every function's arity and types with the header).
Usage: python scripts/gen_rust_ffi.py [--check]"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "constriction_amd.h"
OUT = ROOT / "bindings" / "rust" / "src" / "ffi.rs"

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "double": "f64",
           "cst_status": "CstStatus", "cst_layout": "CstLayout", "cst_coder_config": "CstCoderConfig"}
POINTEES = {"void": "c_void", "char": "c_char", "int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "int64_t": "i64", "uint64_t": "u64", "double": "f64",
            "cst_model": "CstModel", "cst_range_state": "CstRangeState", "cst_chain_heads": "CstChainHeads"}
STRUCT_NAMES = {"cst_coder_config": "CstCoderConfig", "cst_range_state": "CstRangeState", "cst_chain_heads": "CstChainHeads"}
ENUM_ALIASES = {"cst_status": "CstStatus", "cst_stream_status": "CstStreamStatus", "cst_layout": "CstLayout", "cst_family": "CstFamily"}


def rust_type(c: str) -> str:
    c = re.sub(r"\s+", " ", c.replace("*", " * ")).strip()
    toks = c.split(" ")
    stars = toks.count("*")
    base = [t for t in toks if t not in ("*", "const")]
    assert len(base) == 1, c
    const = toks[0] == "const"
    if stars == 0:
        return SCALARS[base[0]]
    t = POINTEES[base[0]]
    inner = f"*const {t}" if const else f"*mut {t}"
    for _ in range(stars - 1):
        inner = f"*mut {inner}"
    return inner


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", lambda m: " " * 0, text, flags=re.S)


def parse_functions(text):
    """[(name, ret, [(ctype, argname)], doc)] in header order"""
    out = []
    pat = re.compile(r"(?:/\*((?:(?!\*/).)*)\*/\s*)?^((?:const\s+)?\w+(?:\s*\*+\s*|\s+))(cst_\w+)\s*\(([^;{]*?)\)\s*;", re.S | re.M)
    for m in pat.finditer(text):
        doc, ret, name, args = m.group(1), m.group(2).strip(), m.group(3), strip_comments(m.group(4))
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = re.sub(r"\s+", " ", a).strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                params.append((mm.group(1).strip(), mm.group(2)))
        out.append((name, ret, params, doc))
    return out


def parse_enums(text):
    out = []
    for m in re.finditer(r"typedef enum (\w+) \{(.*?)\} \w+;", strip_comments(text), re.S):
        items = [(k, int(v)) for k, v in re.findall(r"(CST_\w+)\s*=\s*(-?\d+)", m.group(2))]
        out.append((m.group(1), items))
    return out


def parse_structs(text):
    out = []
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \w+;", strip_comments(text), re.S):
        fields = re.findall(r"(\w+)\s+(\w+);", m.group(2))
        out.append((m.group(1), fields))
    return out


def doc_lines(doc, indent):
    if not doc:
        return []
    lines = [re.sub(r"^\s*\*? ?", "", l).rstrip() for l in doc.strip("\n").split("\n")]
    lines = [l for l in lines if not re.fullmatch(r"-{20,}", l.strip())]
    while lines and not lines[0].strip():
        lines.pop(0)
    while lines and not lines[-1].strip():
        lines.pop()
    return [f"{indent}/// {l}".rstrip() for l in lines]


def generate() -> str:
    text = HEADER.read_text()
    version = int(re.search(r"#define CST_ABI_VERSION (\d+)", text).group(1))
    o = ["//! Raw `extern \"C\"` declarations of libconstriction_amd.so -- GENERATED from include/constriction_amd.h by",
         "//! scripts/gen_rust_ffi.py; do not edit.  One declaration per entry point of the header, in header order, with the",
         "//! header's comments (which cite the reference interface each call replaces).  C enums are `i32` aliases with",
         "//! constants: a value the library adds later must not be undefined behaviour on the Rust side.",
         "#![allow(non_camel_case_types, dead_code, clippy::too_many_arguments)]",
         "use core::ffi::{c_char, c_void};", "",
         f"pub const CST_ABI_VERSION: i32 = {version};", ""]
    for name, items in parse_enums(text):
        alias = ENUM_ALIASES[name]
        o.append(f"/// `{name}`")
        o.append(f"pub type {alias} = i32;")
        for k, v in items:
            o.append(f"pub const {k}: {alias} = {v};")
        o.append("")
    for k, v in re.findall(r"#define (CST_FLAG_\w+) (\d+)u", text):
        o.append(f"pub const {k}: u32 = {v};")
    for k, v in re.findall(r"#define (CST_CODER_\w+) (\d+)\b", text):
        o.append(f"pub const {k}: i32 = {v};")
    o.append("")
    o += ["/// `cst_model`: opaque, device-resident model image", "#[repr(C)]", "pub struct CstModel {", "    _private: [u8; 0],", "}", ""]
    for name, fields in parse_structs(text):
        o += [f"/// `{name}`", "#[repr(C)]", "#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]", f"pub struct {STRUCT_NAMES[name]} {{"]
        o += [f"    pub {f}: {SCALARS[t]}," for t, f in fields]
        o += ["}", ""]
    o += ['#[link(name = "constriction_amd")]', 'extern "C" {']
    for name, ret, params, doc in parse_functions(text):
        o += doc_lines(doc, "    ")
        args = ", ".join(f"{'r#' + n if n in ('type', 'ref', 'in') else n}: {rust_type(t)}" for t, n in params)
        rt = "" if ret == "void" else f" -> {rust_type(ret)}"
        line = f"    pub fn {name}({args}){rt};"
        if len(line) > 120:
            line = f"    pub fn {name}(\n" + "".join(f"        {n}: {rust_type(t)},\n" for t, n in params) + f"    ){rt};"
        o.append(line)
        o.append("")
    if o[-1] == "":
        o.pop()
    o.append("}")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    src = generate()
    if "--check" in sys.argv:
        sys.exit(0 if OUT.exists() and OUT.read_text() == src else 1)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(src)
    print(f"wrote {OUT} ({src.count('pub fn ')} functions)")
