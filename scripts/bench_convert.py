#!/usr/bin/env python3
"""The two conversion kernels of cst_symbols.hip on the 268 M symbols of config C2 (GB/s of HBM traffic: 5 B per int8 symbol, 6 per int16)."""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import _native as N
n = 65536 * 4096
lib = N.lib()
wide = torch.randint(-100, 100, (n,), dtype=torch.int32, device="cuda")
for nb, dt in ((1, torch.int8), (2, torch.int16)):
    narrow = torch.empty(n, dtype=dt, device="cuda")
    back = torch.empty_like(wide)
    t_n = min(bench.event_ms(lambda: N.check(lib.cst_symbols_narrow(C.c_void_p(wide.data_ptr()), n, C.c_void_p(narrow.data_ptr()), nb, None), "narrow"), 10) for _ in range(4))
    t_w = min(bench.event_ms(lambda: N.check(lib.cst_symbols_widen(C.c_void_p(narrow.data_ptr()), nb, n, C.c_void_p(back.data_ptr()), None), "widen"), 10) for _ in range(4))
    ok = bool(torch.equal(back, wide)) and bool(torch.equal(narrow, wide.to(dt)))
    print(f"int{8 * nb}: narrow {t_n:.3f} ms ({n * (4 + nb) / t_n / 1e6:.0f} GB/s)  widen {t_w:.3f} ms ({n * (4 + nb) / t_w / 1e6:.0f} GB/s)  ok={ok}", flush=True)
