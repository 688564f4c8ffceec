#!/usr/bin/env python3
"""One chain coder (constriction.stream.chain.ChainCoder drop-in), a long message, every symbol its own Gaussian: decode
from random words, re-encode, check that the words are restored.  Host copies included."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import constriction_amd as constriction
from constriction_amd import stream  # noqa: F401

rng = np.random.default_rng(1)
fam = constriction.stream.model.QuantizedGaussian(-100, 100)
for n in (1000, 100_000, 1_000_000):
    data = rng.integers(1, 2**32, n, dtype=np.uint64).astype(np.uint32)
    means = rng.uniform(-20, 20, n); stds = np.exp(rng.uniform(-0.5, 3, n))
    c = constriction.stream.chain.ChainCoder(data, seal=True)
    c.decode(fam, means[:100], stds[:100]); torch.cuda.synchronize()
    c = constriction.stream.chain.ChainCoder(data, seal=True)
    t = time.time(); sym = c.decode(fam, means, stds); torch.cuda.synchronize(); d = time.time() - t
    t = time.time(); c.encode_reverse(sym, fam, means, stds); torch.cuda.synchronize(); e = time.time() - t
    ok = np.array_equal(np.concatenate(c.get_data(unseal=True)), data)
    print(f"chain coder, one chain, n={n:8d}: decode {d / n * 1e9:7.1f} ns/sym, encode {e / n * 1e9:7.1f} ns/sym, restored={ok}", flush=True)
