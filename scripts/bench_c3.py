#!/usr/bin/env python3
"""C3 (per-stream Gaussian tables) kernel timings on one GPU; round trip checked."""
import sys, os
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B

n_streams = int(os.environ.get("C3_STREAMS", 65536)); n_per = 4096


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


rng = np.random.default_rng(0xC0FFEE)
mu = torch.from_numpy(-10 + 20 * rng.random(n_streams)).cuda()
sigma = torch.from_numpy(np.exp(np.log(0.5) + rng.random(n_streams) * np.log(32))).cuda()
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
sym3 = torch.clamp(torch.round(z * sigma.float()[:, None] + mu.float()[:, None]), -127, 127).to(torch.int32)
model = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sigma, 12)
enc_ms, enc = timed(lambda: B.ans_encode(sym3, model, (32, 64, 12)))
dec_ms, (dec, st) = timed(lambda: B.ans_decode(enc, model, n_per))
ok = bool(torch.equal(dec, sym3)) and int(enc.status.abs().sum()) == 0
words = enc.total_words()
sym = n_streams * n_per
byts = 4 * sym + 4 * words
print(f"C3 per-stream tables: encode {enc_ms:7.3f} ms ({byts / enc_ms / 1e9:5.2f} TB/s)  decode {dec_ms:7.3f} ms ({byts / dec_ms / 1e9:5.2f} TB/s)  "
      f"round trip {sym / (enc_ms + dec_ms) / 1e6:7.1f} Gsym/s  {words / n_streams:7.1f} words/stream  roundtrip_ok={ok}")
