#!/usr/bin/env python3
"""Range coder kernel timings at the C4 shape; LIB=<path> times a variant library (scripts/exp_variants.sh)."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import constriction_amd._native as N
if os.environ.get("LIB"):
    N.LIB_PATH = Path(os.environ["LIB"]).resolve()
from constriction_amd import batched as B

n_streams, n_per = int(os.environ.get("STREAMS", 65536)), 4096


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
sym = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)
for P in (12, 24):
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    enc_ms, enc = timed(lambda: B.range_encode(sym, model, (32, 64, P)))
    line = f"{os.environ.get('LIB', 'product'):40s} P={P}: encode {enc_ms:6.3f} ms"
    if not os.environ.get("ENCODE_ONLY"):
        dec_ms, (dec, st) = timed(lambda: B.range_decode(enc, model, n_per))
        line += f"  decode {dec_ms:6.3f} ms  roundtrip_ok={bool(torch.equal(dec, sym))}"
    print(line)
