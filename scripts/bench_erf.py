#!/usr/bin/env python3
"""erf throughput on the device: the bit-exact table-driven evaluation vs the fast one the per-symbol kernels try first"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import _native as N

lib = N.lib()
n = 1 << 26
x = (torch.rand(n, device="cuda", dtype=torch.float64) * 8 - 4)
out = torch.empty_like(x)


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, f in (("exact (branchy)", lambda: lib.cst_debug_erf(x.data_ptr(), out.data_ptr(), n, None)),
                ("exact (table)", lambda: lib.cst_debug_erf_tab(x.data_ptr(), out.data_ptr(), n, None)),
                ("fast", lambda: lib.cst_debug_erf_fast(0, x.data_ptr(), out.data_ptr(), n, None))):
    ms = timed(f)
    print(f"{name:16s} {ms:7.3f} ms for {n} evaluations = {n / ms / 1e6:7.1f} G/s")
