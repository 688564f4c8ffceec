#!/usr/bin/env python3
"""The reference's own usage pattern through the drop-in API: ONE coder, a long message, every symbol with its own
Gaussian (src/pybindings/stream/stack.rs:567-588, 733-751).  Decoding one stream is sequential by construction: this
is the latency of the model search per symbol, not a throughput figure."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from constriction_amd.stream import stack, queue, model as M

rng = np.random.default_rng(3)
fam = M.QuantizedGaussian(-100, 100)
for n in (1000, 100_000, 1_000_000):
    means = rng.uniform(-20, 20, n); stds = np.exp(rng.uniform(-0.5, 3.0, n))
    sym = np.clip(np.rint(means + stds * rng.standard_normal(n)), -100, 100).astype(np.int32)
    for name, make_enc, make_dec in (("ans", stack.AnsCoder, None), ("range", queue.RangeEncoder, queue.RangeDecoder)):
        enc = make_enc()
        (enc.encode_reverse if name == "ans" else enc.encode)(sym[:10], fam, means[:10], stds[:10]); torch.cuda.synchronize()
        enc = make_enc()
        t = time.time()
        (enc.encode_reverse if name == "ans" else enc.encode)(sym, fam, means, stds)
        torch.cuda.synchronize(); e = time.time() - t
        dec = enc if name == "ans" else make_dec(enc.get_compressed())
        t = time.time()
        out = dec.decode(fam, means, stds)
        torch.cuda.synchronize(); d = time.time() - t
        ok = bool(np.array_equal(out, sym))
        print(f"{name:5s} n={n:8d}: encode {e * 1e3:9.3f} ms ({e / n * 1e9:7.1f} ns/sym)  decode {d * 1e3:9.3f} ms ({d / n * 1e9:7.1f} ns/sym)  ok={ok}", flush=True)
