#!/usr/bin/env python3
"""Kernel time against symbols per stream at 65 536 streams: the fixed part (launch, table staging, first HBM round trip,
sealing) and the per-symbol slope of each coder kernel."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


n_streams = 65536
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, 4096), generator=g, device="cuda", dtype=torch.float32)
full = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)
model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
rows = []
for n in (32, 64, 512, 1024, 2048, 4096):
    sym = full[:, :n].contiguous()
    re, enc_r = timed(lambda: B.range_encode(sym, model, (32, 64, 12)))
    ae, enc_a = timed(lambda: B.ans_encode(sym, model, (32, 64, 12)))
    rd, _ = timed(lambda: B.range_decode(enc_r, model, n))
    ad, _ = timed(lambda: B.ans_decode(enc_a, model, n))
    rows.append((n, re, ae, rd, ad))
    print(f"n_per={n:5d}  range enc {re*1e3:7.1f} us  ans enc {ae*1e3:7.1f} us  range dec {rd*1e3:7.1f} us  ans dec {ad*1e3:7.1f} us")
(n0, *a), (n1, *b) = rows[-2], rows[-1]
print("slope ns/symbol/stream-wave and intercept us (from the two largest):")
for name, x, y in zip(("range enc", "ans enc", "range dec", "ans dec"), a, b):
    slope = (y - x) / (n1 - n0)
    print(f"  {name}: {slope * 1e6:6.2f} ns per symbol step, fixed {1e3 * (y - slope * n1):6.1f} us")
