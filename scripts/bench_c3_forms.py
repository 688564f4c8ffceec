#!/usr/bin/env python3
"""C3 (one table per stream) at 65 536 x 4096: the default call (eight jump points per stream, ans_decode_pt_sub_kernel) beside the
plain pair, int32 and int8 matrices; min / median of 5 rounds of 10 launches.  AB_LIB=<path>: an experimental build."""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if os.environ.get("AB_LIB"):
    import constriction_amd._native as _N
    _N.LIB_PATH = Path(os.environ["AB_LIB"]).resolve()
import bench
from constriction_amd import batched as B

n, k = 65536, int(os.environ.get("C3_N", 4096))
mu_d, sigma_d = bench.c3_parameters(bench.SEED, 0, n, k, "cuda")
m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
sym32 = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, m3.cdfs_device(), 12)


def leg(tag, sym, **kw):
    enc = B.ans_encode(sym, m3, (32, 64, 12), **kw)
    ek = B.last_kernel()
    dec = torch.empty_like(sym)
    es, ds = [], []
    for rep in range(5):
        es.append(bench.event_ms(lambda: B.ans_encode(sym, m3, (32, 64, 12), out=enc, **kw), 10))
        ds.append(bench.event_ms(lambda: B.ans_decode(enc, m3, k, out=dec), 10))
    dk = B.last_kernel()
    pts = enc.jump.pos.shape[1] if enc.jump is not None else 0
    print(f"{tag:24s} k={pts:2d} encode min {min(es):.3f} med {np.median(es):.3f}  decode min {min(ds):.3f} med {np.median(ds):.3f} ms  "
          f"ok={bool(torch.equal(dec, sym))}  {ek} / {dk}", flush=True)


for name, sym in (("int32", sym32), ("int8", sym32.to(torch.int8))):
    leg(f"{name} default", sym)
    for kk in (4, 16):
        leg(f"{name} jump_points={kk}", sym, jump_points=kk)
    leg(f"{name} jump_points=0", sym, jump_points=0)
