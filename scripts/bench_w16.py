#!/usr/bin/env python3
"""The (16,32,12) preset at the headline shape: words one per 32-bit slot against CST_FLAG_PACKED_W16 (two per slot)."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n, k, P = 65536, 4096, 12
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, torch.from_numpy(m.cdf().astype(np.int64)).cuda(), P)
dec = torch.empty_like(sym)
for packed in (False, True, False, True):
    enc = B.ans_encode(sym, m, (16, 32, P), packed16=packed)
    e = bench.event_ms(lambda: B.ans_encode(sym, m, (16, 32, P), out=enc), 7)
    d = bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 7)
    pk, off = B.compact(enc)
    c = bench.event_ms(lambda: B.compact(enc, out=(pk, off)), 7)
    print(f"(16,32,12) packed={packed}: encode {e:.3f} decode {d:.3f} compact {c:.3f} ms  ok={bool(torch.equal(dec, sym))}  {B.last_kernel()}", flush=True)
