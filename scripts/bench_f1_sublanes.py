#!/usr/bin/env python3
"""f1 (every symbol its own f64 (mean, std), 65 536 x 4096) decoded plainly and through k jump points per stream."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n, k = 65536, 4096
g = torch.Generator(device="cuda").manual_seed(1)
means = torch.rand((n, k), generator=g, device="cuda", dtype=torch.float64) * 20 - 10
stds = torch.exp(torch.rand((n, k), generator=g, device="cuda", dtype=torch.float64) * 3.4 - 0.7)
sym = torch.clamp(torch.round(torch.randn((n, k), generator=g, device="cuda", dtype=torch.float64) * stds + means), -127, 127).to(torch.int32)
enc = B.ans_encode_gaussian(sym, -127, 127, means, stds)
dec = torch.empty_like(sym)
e = bench.event_ms(lambda: B.ans_encode_gaussian(sym, -127, 127, means, stds, out=enc), 3)
d = bench.event_ms(lambda: B.ans_decode_gaussian(enc, -127, 127, means, stds, out=dec), 3)
print(f"plain: encode {e:.3f} decode {d:.3f} ms ok={bool(torch.equal(dec, sym))}", flush=True)
for chunks in (2, 4):
    pair = B.ans_encode_gaussian_checkpointed(sym, -127, 127, means, stds, k // chunks)
    enc2, ck = pair
    same = bool(torch.equal(enc2.n_words, enc.n_words))
    dec.zero_()
    st = torch.empty((n, chunks), dtype=torch.int32, device="cuda")
    e = bench.event_ms(lambda: B.ans_encode_gaussian_checkpointed(sym, -127, 127, means, stds, k // chunks, out=pair), 3)
    d = bench.event_ms(lambda: B.ans_decode_gaussian_checkpointed(enc2, ck, -127, 127, means, stds, out=dec, status=st), 3)
    print(f"k={chunks}: encode {e:.3f} decode {d:.3f} ms ok={bool(torch.equal(dec, sym))} status0={int(st.abs().sum()) == 0} same_counts={same}", flush=True)
