import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from constriction_amd import batched as B
for P in (24, 16):
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
    sym = bench.synth_symbols_device(0xC0FFEE, 0, 65536, 4096, -50, cdf, P)
    enc = B.ans_encode(sym, m, (32, 64, P)); k = B.last_kernel()
    t = min(bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, P), out=enc), 10) for _ in range(6))
    e2, ck = B.ans_encode_checkpointed(sym, m, 2048, (32, 64, P)); k2 = B.last_kernel()
    t2 = min(bench.event_ms(lambda: B.ans_encode_checkpointed(sym, m, 2048, (32, 64, P)), 10) for _ in range(4))
    print(f"P={P} int32 encode {t:.3f} [{k}]  with jump points {t2:.3f} [{k2}]", flush=True)
