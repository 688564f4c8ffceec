#!/usr/bin/env python3
"""Fixed cost per launch of the coder kernels: times at 65 536 streams x {256, 1024, 2048, 4096} symbols, and the intercept
of the line through them (what staging tables, priming rings and the tail cost next to the per-symbol loop)."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n = 65536
for coder, cfg in (("ans", (32, 64, 12)), ("ans", (32, 64, 24)), ("ans", (16, 32, 12)), ("range", (32, 64, 12)), ("range", (32, 64, 24))):
    P = cfg[2]
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
    enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
    ks, es, ds = [256, 1024, 2048, 4096], [], []
    for k in ks:
        sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
        enc = enc_f(sym, m, cfg)
        dec = torch.empty_like(sym)
        es.append(min(bench.event_ms(lambda: enc_f(sym, m, cfg, out=enc), 10) for _ in range(5)))
        ds.append(min(bench.event_ms(lambda: dec_f(enc, m, k, out=dec), 10) for _ in range(5)))
    fe, fd = np.polyfit(ks, es, 1), np.polyfit(ks, ds, 1)
    print(f"{coder} {cfg}: encode {['%.3f' % x for x in es]} -> {fe[1]*1e3:.1f} us + {fe[0]*1e6:.2f} ns/step;  decode {['%.3f' % x for x in ds]} -> {fd[1]*1e3:.1f} us + {fd[0]*1e6:.2f} ns/step")
