#!/usr/bin/env python3
"""One direction of one coder at the headline shape on an experimental build of the library (AB_LIB=<path>): min / median of 5
rounds of 8 launches.  usage: bench_one_lib.py ans|range P encode|decode"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if os.environ.get("AB_LIB"):
    import constriction_amd._native as _N
    _N.LIB_PATH = Path(os.environ["AB_LIB"]).resolve()
import bench
from constriction_amd import batched as B

coder, P, what = sys.argv[1], int(sys.argv[2]), sys.argv[3]
n, k = 65536, 4096
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, torch.from_numpy(m.cdf().astype(np.int64)).cuda(), P)
enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
enc = enc_f(sym, m, (32, 64, P), stride=2080 if coder == "ans" and P == 12 else None)
dec = torch.empty_like(sym)
fn = (lambda: enc_f(sym, m, (32, 64, P), out=enc)) if what == "encode" else (lambda: dec_f(enc, m, k, out=dec))
ts = [bench.event_ms(fn, 8) for _ in range(5)]
dec_f(enc, m, k, out=dec)
print(f"{os.path.basename(os.environ.get('AB_LIB', 'lib')):14s} {coder} P={P} {what}: min {min(ts):.4f} med {np.median(ts):.4f} ms ok={bool(torch.equal(dec, sym))}", flush=True)
