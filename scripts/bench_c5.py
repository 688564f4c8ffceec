#!/usr/bin/env python3
"""Kernel times of the shared-table ANS coder at the C5 shard's shape (131 072 x 4096 by default): min / median of 6 rounds."""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
k = 4096
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, 12)
enc = B.ans_encode(sym, m, (32, 64, 12))
dec = torch.empty_like(sym)
es, ds = [], []
for rep in range(6):
    es.append(bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, 12), out=enc), 10))
    ds.append(bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 10))
print(f"{n} x {k} PC_BIG={os.environ.get('CST_PC_BIG')}: encode min {min(es):.3f} med {np.median(es):.3f}  decode min {min(ds):.3f} med {np.median(ds):.3f} ms  ok={bool(torch.equal(dec, sym))}")
