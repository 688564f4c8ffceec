"""Builds constriction_amd/lib/libconstriction_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repository snapshot.  Every .hip file is its own translation unit (no relocatable device
code), so the files are compiled to objects in parallel and only the stale ones are rebuilt."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libconstriction_amd.so"
OBJ = PKG.parent / "build" / "obj"

# -ffp-contract=off is REQUIRED: the f64 model arithmetic must round exactly like the CPU reference
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
LINK_LIBS = ["-ldl"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def _headers():
    return (list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + [PKG.parent / "include" / "constriction_amd.h"])


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB, list(CSRC.glob("*.hip")) + _headers() + [Path(__file__)])


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(src: Path) -> list:
    """the files a translation unit really includes (quoted includes, transitively): touching one generated .inc
    recompiles only the kernels built from it"""
    seen, todo = {}, [src]
    while todo:
        f = todo.pop()
        if f in seen or not f.exists():
            continue
        seen[f] = True
        for name in _INCLUDE.findall(f.read_text()):
            todo.append((f.parent / name).resolve())
    return list(seen)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    if force or _stale(obj, _deps(src) + [Path(__file__)]):
        res = subprocess.run([_hipcc(), *FLAGS, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n" + res.stdout + res.stderr)
        if res.stderr.strip():
            print(res.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    OBJ.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, force), srcs))
    tmp = LIB.with_suffix(".so.tmp")
    res = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), *LINK_LIBS, "-o", str(tmp)],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
