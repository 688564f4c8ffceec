"""The drop-in boundary is a plain C ABI: examples/c_abi_roundtrip.c (C99, no Python, no torch) compiles and links against
include/constriction_amd.h + the in-tree library with gcc (CPU check), and on an MI355X it runs: encode, decode, compare."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_NO_PC_ENCODER", "CST_SMALL_KERNELS"))      # (A/B runs, scripts/alt_paths.sh: other kernels, same results)
LIB = ROOT / "constriction_amd" / "lib"


def build(tmp_path):
    if shutil.which("gcc") is None or not Path("/opt/rocm/include/hip/hip_runtime_api.h").exists():
        pytest.skip("gcc or the HIP headers are not here")
    if not (LIB / "libconstriction_amd.so").exists():
        from constriction_amd import build as b
        b.build_library()
    exe = tmp_path / "c_abi_roundtrip"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
           str(ROOT / "examples" / "c_abi_roundtrip.c"), "-L", str(LIB), "-lconstriction_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{LIB}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_c_example_compiles_and_links(tmp_path):
    assert build(tmp_path).exists()


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    res = subprocess.run([str(build(tmp_path))], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 mismatches" in res.stdout and "words against the int32 call" in res.stdout and " 0 mismatches (symbols" in res.stdout
    assert ALT or "ans_encode_pc_n8_kernel<ckpt>" in res.stdout          # the int8 matrix was read by the encoder loops themselves


@pytest.mark.gpu
def test_python_int8_example_runs():
    """examples/batched_int8_latents.py: int8 latents through the batched coder, packed words, jump points"""
    import sys
    res = subprocess.run([sys.executable, str(ROOT / "examples" / "batched_int8_latents.py"), "1024", "512"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert ALT or ("ans_encode_pc_n8_kernel" in res.stdout and "ans_decode_n8_kernel" in res.stdout)
    assert res.stdout.strip().endswith("the int32 call's")
