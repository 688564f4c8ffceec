#!/usr/bin/env python3
"""The C5 shard (131 072 x 4096 per GPU) on the kernels the dispatcher picks and, with CST_SMALL_KERNELS=0 / enc / dec, on the
one-wave-per-SIMD kernels run as two rounds of workgroups; hot and after a 1-GiB fill.  usage: bench_c5.py [n_streams]"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if os.environ.get("AB_LIB"):                       # an experimental build of the library (A/B runs on one box)
    import constriction_amd._native as _N
    _N.LIB_PATH = Path(os.environ["AB_LIB"]).resolve()
import bench
from constriction_amd import batched as B

n, k, P = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 4096, 12
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
stride = os.environ.get("STRIDE")
enc = B.ans_encode(sym, m, (32, 64, P), stride=int(stride) if stride else None)
dec = torch.empty_like(sym)
fill = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
for rep in range(3):
    e = bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, P), out=enc), 7)
    d = bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 7)
    cold = []
    for fn in (lambda: B.ans_encode(sym, m, (32, 64, P), out=enc), lambda: B.ans_decode(enc, m, k, out=dec)):
        ts = []
        for _ in range(3):
            fill.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        cold.append(min(ts))
    print(f"C5 {n} [{os.environ.get('CST_SMALL_KERNELS', 'default')}] stride {enc.words.shape[1]}: encode {e:.3f} decode {d:.3f} ms hot; "
          f"{cold[0]:.3f} / {cold[1]:.3f} after a 1-GiB fill; ok={bool(torch.equal(dec, sym))}", flush=True)
