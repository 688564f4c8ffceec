#!/usr/bin/env python3
"""Coder kernel timings against the number of streams (4096 symbols each)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B


def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


n_per = 4096
for n_streams in (65536, 66560, 98304, 131072):
    g = torch.Generator(device="cuda").manual_seed(1)
    z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
    sym = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)
    del z
    for P in (12, 24):
        model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
        out = torch.empty_like(sym)
        re, er = timed(lambda: B.range_encode(sym, model, (32, 64, P)))
        rd, _ = timed(lambda: B.range_decode(er, model, n_per, out=out))
        ok_r = bool(torch.equal(out, sym))
        ae, ea = timed(lambda: B.ans_encode(sym, model, (32, 64, P)))
        ad, _ = timed(lambda: B.ans_decode(ea, model, n_per, out=out))
        ok_a = bool(torch.equal(out, sym))
        print(f"streams={n_streams:6d} P={P:2d}  range enc {re:6.3f} dec {rd:6.3f} ok={ok_r}   ans enc {ae:6.3f} dec {ad:6.3f} ok={ok_a}")
    del sym
