"""Builds constriction_amd/lib/libconstriction_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repository snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libconstriction_amd.so"

# -ffp-contract=off is REQUIRED: the f64 model arithmetic must round exactly like the CPU reference
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = (list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) +
            [PKG.parent / "include" / "constriction_amd.h"])
    return any(d.stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    srcs = sorted(str(p) for p in CSRC.glob("*.hip"))
    tmp = LIB.with_suffix(".so.tmp")
    cmd = [_hipcc(), *FLAGS, *srcs, "-o", str(tmp)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
