#!/usr/bin/env python3
"""Kernel timings at 65 536 + 1 streams against 65 536: what one partial wave costs each kernel family."""
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
for n_streams in (65536, 65537, 65536 + 63):
    g = torch.Generator(device="cuda").manual_seed(1)
    z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
    sym = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)
    del z
    out = torch.empty_like(sym)
    for cfg in ((32, 64, 12), (32, 64, 24), (16, 32, 12)):
        model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, cfg[2])
        ae, ea = timed(lambda: B.ans_encode(sym, model, cfg))
        ad, _ = timed(lambda: B.ans_decode(ea, model, n_per, out=out))
        line = f"streams={n_streams:6d} {cfg}: ans enc {ae:6.3f} dec {ad:6.3f}"
        if cfg[0] == 32:
            re, er = timed(lambda: B.range_encode(sym, model, cfg))
            rd, _ = timed(lambda: B.range_decode(er, model, n_per, out=out))
            line += f"   range enc {re:6.3f} dec {rd:6.3f}"
        print(line)
    del sym, out
