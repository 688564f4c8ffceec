#!/usr/bin/env python3
"""Kernel times of the shared-table ANS coder (32,64,12) for batch shapes of equal or doubled footprint: separates the
effect of a second wave per SIMD (streams > 65536) from the effect of the data footprint."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
import constriction_amd._native as N
if os.environ.get("LIB"):          # a variant library built by scripts/exp_variants.sh
    N.LIB_PATH = Path(os.environ["LIB"]).resolve()
import bench
from constriction_amd import batched as B

m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
shapes = [(65536, 4096), (131072, 2048), (262144, 1024), (65536, 8192), (131072, 4096), (262144, 4096), (98304, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for n, k in shapes:
    sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, 12)
    enc = B.ans_encode(sym, m, (32, 64, 12)); dec = torch.empty_like(sym)
    e = bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, 12), out=enc), 10)
    d = bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 10)
    ok = bool(torch.equal(dec, sym))
    ns = n * k
    print(f"{n:7d} x {k:5d}: encode {e:7.3f} ms ({ns / e / 1e6:7.1f} Gsym/s)  decode {d:7.3f} ms ({ns / d / 1e6:7.1f} Gsym/s)  ok={ok}", flush=True)
    del sym, enc, dec
    torch.cuda.empty_cache()
