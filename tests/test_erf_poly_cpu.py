"""CPU-only: the polynomial table of the per-symbol Gaussian kernels' fast erf against the oracle's erf (oracle/oracle.c, the
msun restatement the device's exact evaluation is bit-identical to)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O


def test_erf_polynomials_stay_within_the_bound():
    """constriction_amd/csrc/cst_erf_poly.inc (scripts/gen_erf_poly.py): the FAST erf of the per-symbol Gaussian kernels,
    replayed on the CPU with the device's operations (f64 fused multiply-adds emulated with mpmath: exact product and sum,
    one rounding) on a grid of every interval, its ends and the neighbourhood of 0 -- within 2^-50 of the oracle's erf, where
    the kernels assume 2^-46 (cst_math.hpp, kErfFastBound).  The file itself must be what the generator writes."""
    import importlib.util
    import math
    mp = pytest.importorskip("mpmath")
    root = Path(__file__).resolve().parent.parent
    inc = root / "constriction_amd" / "csrc" / "cst_erf_poly.inc"
    spec = importlib.util.spec_from_file_location("gen_erf_poly", root / "scripts" / "gen_erf_poly.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rows = gen.coefficients()
    listed = [[float.fromhex(t.strip()) for t in line.split("//")[0].split(",") if t.strip()]
              for line in inc.read_text().splitlines() if not line.startswith("//")]
    assert listed == rows
    mp.mp.prec = 200

    def fma(a, b, c):
        return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))

    def erf_fast(x):
        s = min(abs(x) * 16.0, float.fromhex("0x1.7ffffffffffffp+6"))
        u = fma(s - math.floor(s), 2.0, -1.0)
        c = rows[int(s)]
        y = c[7]
        for k in range(6, -1, -1):
            y = fma(y, u, c[k])
        return math.copysign(y, x)

    lib = O.load()
    rng = np.random.default_rng(5)
    xs = []
    for r in range(96):
        xs += list((r + rng.random(12)) / 16.0) + [r / 16.0, math.nextafter((r + 1) / 16.0, 0.0), (r + 0.5) / 16.0]
    xs += list(rng.random(50) * 1e-3) + [1e-300, 5.999999, 6.0, 27.0, 1e300]
    worst = max(abs(erf_fast(sgn * x) - lib.cst_oracle_erf(sgn * x)) for x in xs for sgn in (1.0, -1.0))
    assert worst <= 2.0 ** -50, worst
