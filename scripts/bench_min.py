#!/usr/bin/env python3
"""Best-of kernel times of one coder / preset at the headline shape (for A/B experiments on a noisy shared box):
usage: bench_min.py ans|range W S P [n_per] [layout] -- min and median over 8 rounds of 10 launches each"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
if os.environ.get("AB_LIB"):                       # an experimental build of the library (A/B runs on one box)
    import constriction_amd._native as _N
    _N.LIB_PATH = Path(os.environ["AB_LIB"]).resolve()
import bench
from constriction_amd import batched as B

coder, W, S, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
layout = sys.argv[6] if len(sys.argv) > 6 else "stream_major"
n = 65536
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
if layout == "symbol_major":
    sym = sym.t().contiguous()
enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
enc = enc_f(sym, m, (W, S, P), layout, stride=int(os.environ.get('STRIDE', 0)) or None)      # STRIDE=<words>: slab stride
dec = torch.empty_like(sym)
es, ds = [], []
for rep in range(8):
    es.append(bench.event_ms(lambda: enc_f(sym, m, (W, S, P), layout, out=enc), 10))
    ds.append(bench.event_ms(lambda: dec_f(enc, m, k, layout, out=dec), 10))
cold = ""
if os.environ.get("COLD"):                         # COLD=1: the same launches after a 1-GiB fill each (nothing of the batch in L2 / Infinity Cache)
    flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")

    def after_flush(fn):
        ts = []
        for _ in range(6):
            if os.environ.get("COLD") == "read":       # a 1-GiB READ instead of a fill: the caches end up full of CLEAN lines
                flush.view(torch.int32).sum()
            else:
                flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts), float(np.median(ts))
    ce, cd = after_flush(lambda: enc_f(sym, m, (W, S, P), layout, out=enc)), after_flush(lambda: dec_f(enc, m, k, layout, out=dec))
    cold = f"  COLD encode min {ce[0]:.3f} med {ce[1]:.3f} decode min {cd[0]:.3f} med {cd[1]:.3f}"
print(f"{os.environ.get('AB_LIB', '').split('/')[-1]} stride {enc.words.shape[1]} {coder} ({W},{S},{P}) {k} {layout}: encode min {min(es):.3f} med {np.median(es):.3f}  decode min {min(ds):.3f} med {np.median(ds):.3f} ms  ok={bool(torch.equal(dec, sym))}{cold}")
