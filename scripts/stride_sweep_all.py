#!/usr/bin/env python3
"""Encode / decode time of every shared-table configuration of the bench at each candidate slab stride
(batched.tuned_stride's list): which kernels care how far apart the slabs lie."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n, k = 65536, 4096
for coder, cfg, layout in (("ans", (32, 64, 12), "stream_major"), ("ans", (32, 64, 24), "stream_major"), ("ans", (16, 32, 12), "stream_major"),
                           ("ans", (32, 64, 12), "symbol_major"), ("range", (32, 64, 12), "stream_major"), ("range", (32, 64, 24), "stream_major"),
                           ("range", (32, 64, 12), "symbol_major")):
    P = cfg[2]
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
    sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
    if layout == "symbol_major":
        sym = sym.t().contiguous()
    for rep in range(2):
        rows = []
        best = B.tuned_stride(sym, m, cfg, layout, coder=coder, report=rows)
        print(f"{coder} {cfg} {layout} pass {rep}: best {best}; base {rows[0][0]}: {rows[0][1]:.3f} + {rows[0][2]:.3f}")
        print("   " + "  ".join(f"{c}:{te:.3f}+{td:.3f}" for c, te, td in rows))
