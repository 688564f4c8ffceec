#!/usr/bin/env python3
"""Per-symbol quantized Gaussians (the reference's flagship call, src/pybindings/stream/stack.rs:567-588, 733-751) batched:
n_streams x n_per symbols, every symbol with its own (mean, std) in f64; both coders; round trip checked."""
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


n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
means = (torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) * 20 - 10)
stds = torch.exp(torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) * 3.4 - 0.7)
sym = torch.clamp(torch.round(torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) * stds + means), -127, 127).to(torch.int32)
for name, enc_f, dec_f in (("ans", B.ans_encode_gaussian, B.ans_decode_gaussian), ("range", B.range_encode_gaussian, B.range_decode_gaussian)):
    e, enc = timed(lambda: enc_f(sym, -127, 127, means, stds))
    d, (dec, st) = timed(lambda: dec_f(enc, -127, 127, means, stds))
    ok = bool(torch.equal(dec, sym)) and int(st.abs().sum()) == 0
    ns = n_streams * n_per
    print(f"{name:5s} per-symbol Gaussians {n_streams} x {n_per}: encode {e:8.3f} ms ({ns / e / 1e6:6.1f} Gsym/s)  decode {d:8.3f} ms ({ns / d / 1e6:6.1f} Gsym/s)  roundtrip_ok={ok}")
