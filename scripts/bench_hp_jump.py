#!/usr/bin/env python3
"""12 < P <= 24 with jump points: the small-footprint bucket-entry decoders (two waves per SIMD) on the k x 65 536 virtual streams of a batch.
usage: bench_hp_jump.py [P] [n_streams] [n_per]"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

P = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
k_per = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym32 = bench.synth_symbols_device(0xC0FFEE, 0, n, k_per, -50, cdf, P)
for dt in (torch.int32, torch.int8, torch.int16):
    sym = sym32.to(dt)
    for k in (0, 2, 4):
        enc = B.ans_encode(sym, m, (32, 64, P), jump_points=k) if k else B.ans_encode(sym, m, (32, 64, P))
        ek = B.last_kernel()
        out = torch.empty_like(sym)
        B.ans_decode(enc, m, k_per, out=out)
        dk = B.last_kernel()
        ok = bool(torch.equal(out, sym))
        e = min(bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, P), jump_points=k) if k else B.ans_encode(sym, m, (32, 64, P), out=enc), 10) for _ in range(4))
        d = min(bench.event_ms(lambda: B.ans_decode(enc, m, k_per, out=out), 10) for _ in range(4))
        print(f"P={P} {n} x {k_per} {str(dt)[6:]:6s} k={k}: encode {e:.3f} [{ek}]  decode {d:.3f} [{dk}]  ok={ok}", flush=True)
