#!/usr/bin/env python3
"""Kernel times of the per-stream-table coder (BASELINE config C3) at 65 536 x 4096: min / median of 6 rounds of 10 launches.
AB_LIB=<path>: an experimental build of the library."""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if os.environ.get("AB_LIB"):
    import constriction_amd._native as _N
    _N.LIB_PATH = Path(os.environ["AB_LIB"]).resolve()
import bench
from constriction_amd import batched as B

n, k = 65536, 4096
mu_d, sigma_d = bench.c3_parameters(bench.SEED, 0, n, k, "cuda")
m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
sym = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, m3.cdfs_device(), 12)
enc = B.ans_encode(sym, m3, (32, 64, 12))
dec = torch.empty_like(sym)
es, ds = [], []
for rep in range(6):
    es.append(bench.event_ms(lambda: B.ans_encode(sym, m3, (32, 64, 12), out=enc), 10))
    ds.append(bench.event_ms(lambda: B.ans_decode(enc, m3, k, out=dec), 10))
print(f"C3 {os.environ.get('AB_LIB', 'lib')}: encode min {min(es):.3f} med {np.median(es):.3f}  decode min {min(ds):.3f} med {np.median(ds):.3f} ms  ok={bool(torch.equal(dec, sym))}")
if os.environ.get("C3_STRIDES"):
    rows = []
    best = B.tuned_stride(sym, m3, (32, 64, 12), report=rows)
    print("tuned", best, "  ".join(f"{c}:{te:.3f}+{td:.3f}" for c, te, td in rows))
