"""Builds the RCCL test double (tests/rccl_double/fake_rccl.cpp -> tests/_build/libfake_rccl.so).  Test infrastructure."""
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB = HERE.parent / "_build" / "libfake_rccl.so"


def build() -> Path:
    src = HERE / "fake_rccl.cpp"
    if not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        LIB.parent.mkdir(exist_ok=True)
        subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src),
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", str(LIB)], check=True)
    return LIB
