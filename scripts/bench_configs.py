#!/usr/bin/env python3
"""Kernel timings of the other BASELINE configurations on one GPU (not the bench line; cited in DESIGN.md 4):
C2' = (16,32,12) preset, C3 = per-stream Gaussian tables, C4 = range coder at P = 12 and P = 24,
C2-24 = shared table at P = 24 (generic step path).  Symbols are synthetic, in-support; every round trip is checked."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B

n_streams, n_per = 65536, 4096


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def report(name, enc_ms, dec_ms, words):
    sym = n_streams * n_per
    byts = 4 * sym + 4 * words
    print(f"{name:34s} encode {enc_ms:7.3f} ms ({byts / enc_ms / 1e9:5.2f} TB/s)  decode {dec_ms:7.3f} ms ({byts / dec_ms / 1e9:5.2f} TB/s)  "
          f"round trip {sym / (enc_ms + dec_ms) / 1e6:7.1f} Gsym/s   {words / n_streams:7.1f} words/stream")


g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
sym = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)

for cfg in [(32, 64, 12), (16, 32, 12), (32, 64, 24)]:
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, cfg[2])
    enc_ms, enc = timed(lambda: B.ans_encode(sym, model, cfg))
    dec_ms, (dec, st) = timed(lambda: B.ans_decode(enc, model, n_per))
    assert torch.equal(dec, sym)
    report(f"ANS shared table {cfg}", enc_ms, dec_ms, enc.total_words())

for P in (12, 24):
    cfg = (32, 64, P)
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    enc_ms, enc = timed(lambda: B.range_encode(sym, model, cfg))
    dec_ms, (dec, st) = timed(lambda: B.range_decode(enc, model, n_per))
    assert torch.equal(dec, sym)
    report(f"range coder {cfg}", enc_ms, dec_ms, enc.total_words())

rng = np.random.default_rng(3)
mu = torch.from_numpy(rng.uniform(-10, 10, n_streams)).cuda()
sigma = torch.from_numpy(np.exp(np.log(0.5) + rng.uniform(0, 1, n_streams) * np.log(32))).cuda()
sym3 = torch.clamp(torch.round(z * sigma.float()[:, None] + mu.float()[:, None]), -127, 127).to(torch.int32)
model = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sigma, 12)
enc_ms, enc = timed(lambda: B.ans_encode(sym3, model, (32, 64, 12)))
dec_ms, (dec, st) = timed(lambda: B.ans_decode(enc, model, n_per))
assert torch.equal(dec, sym3)
report("ANS per-stream tables (C3)", enc_ms, dec_ms, enc.total_words())
