"""CPU-only: the C-ABI library builds/loads and exports exactly the symbols include/constriction_amd.h declares.
No compute call is made (there is no GPU in the build container)."""
import ctypes
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "constriction_amd.h"


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cst_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from constriction_amd import build, _native
    build.build_library()
    return _native.load_library()


def test_header_is_plain_c():
    """The boundary is C: the header must compile as C99 with gcc."""
    res = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-x", "c", str(HEADER)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_every_declared_symbol_is_exported_and_bound(lib):
    from constriction_amd import _native
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported by the library"
        assert s in _native.SIGNATURES, f"{s} has no ctypes signature in _native.py"
    assert sorted(_native.SIGNATURES) == syms


def test_abi_version_and_bounds(lib):
    from constriction_amd._native import CoderConfig
    assert lib.cst_abi_version() == 5
    # config C2: ceil(4096*12/32) + 2 = 1538
    assert lib.cst_ans_max_words(4096, CoderConfig(32, 64, 12)) == 1552     # 1538 rounded up to 64 bytes
    assert lib.cst_ans_max_words(5, CoderConfig(32, 64, 24)) == 16        # 4 + 2 rounded up
    assert lib.cst_ans_max_words(0, CoderConfig(16, 32, 12)) == 32        # 2 rounded up (32 x 16-bit words)


def test_no_gpu_means_loud_failure(lib):
    """Without a device the product path raises; it never computes on the CPU."""
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from constriction_amd import _native
    with pytest.raises(_native.BackendUnavailable):
        _native.lib()
    out = ctypes.c_void_p()
    import numpy as np
    cdf = np.array([0, 1000, 4096], dtype=np.uint32)
    st = lib.cst_model_create_table(12, 0, 2, cdf.ctypes.data, ctypes.byref(out))
    assert st == _native.CST_ERR_NO_DEVICE and not out.value


def test_product_never_imports_oracle():
    """No file of the product package may import, link or load the test oracle."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|liboracle|oracle/|oracle\.oracle|cst_oracle_)")
    for p in (ROOT / "constriction_amd").rglob("*"):
        if p.suffix in {".py", ".hip", ".hpp", ".h", ".cpp"}:
            assert not pat.search(p.read_text()), p
