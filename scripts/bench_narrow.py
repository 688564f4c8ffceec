import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from constriction_amd import batched as B
n, k, P = 65536, 4096, 12
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, torch.from_numpy(m.cdf().astype(np.int64)).cuda(), P)
for dt in (torch.int8, torch.int16):
    nar = sym.to(dt); enc = B.ans_encode(nar, m, (32, 64, P)); dec = torch.empty_like(nar)
    e = bench.event_ms(lambda: B.ans_encode(nar, m, (32, 64, P), out=enc), 7)
    d = bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 7)
    print(dt, f"encode {e:.3f} decode {d:.3f} ok={bool(torch.equal(dec, nar))}")
