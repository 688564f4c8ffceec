"""Per-step latency of the ragged kernels: uniform documents of 2000 (400) symbols at 16 waves, one wave and two waves per SIMD."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, bench
from constriction_amd import batched as B
n_sym, P = 64, 24
w = 0.93 ** np.arange(n_sym)
prob = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n_sym)).astype(np.int64)); prob[0] += (1 << P) - int(prob.sum())
cdf = np.concatenate([[0], np.cumsum(prob)]).astype(np.uint32)
model = B.Model.from_cdf(cdf, 0, P)
for n_docs, L in ((1024, 2000), (16384, 2000), (65536, 2000), (131072, 2000), (65536, 400)):
    offsets = torch.arange(n_docs + 1, dtype=torch.int64, device="cuda") * L
    flat = torch.randint(0, n_sym, (n_docs * L,), device="cuda", dtype=torch.int32)
    enc = B.ans_encode_ragged(flat, offsets, model, order=None)
    dec, st = B.ans_decode_ragged(enc, model, offsets, order=None)
    e = min(bench.event_ms(lambda: B.ans_encode_ragged(flat, offsets, model, order=None), 3) for _ in range(2))
    d = min(bench.event_ms(lambda: B.ans_decode_ragged(enc, model, offsets, out=dec, order=None), 3) for _ in range(2))
    print(f"{n_docs} docs x {L}: encode {e:.3f} ms ({e*1e6/L:.0f} ns/step) decode {d:.3f} ms ({d*1e6/L:.0f} ns/step) ok={bool(torch.equal(dec, flat))}", flush=True)
