#!/usr/bin/env python3
"""Big ragged batches: 100 000 and 1 000 000 documents of 20 .. 2000 symbols (log-uniform), 64 symbols at P = 24, in the order
given (slot i = document i), with the schedule of `*_ragged_ordered` (documents sorted by length, `order="sorted"`: the sort
is inside the timed call of the encoder, the decoder reuses it), and with the documents physically sorted."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, bench
from constriction_amd import batched as B
rng = np.random.default_rng(1)
n_sym, P = 64, 24
w = 0.93 ** np.arange(n_sym)
prob = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n_sym)).astype(np.int64)); prob[0] += (1 << P) - int(prob.sum())
cdf = np.concatenate([[0], np.cumsum(prob)]).astype(np.uint32)
model = B.Model.from_cdf(cdf, 0, P)
for n_docs in (100_000, 1_000_000):
    base = np.exp(rng.uniform(np.log(20), np.log(2000), n_docs)).astype(np.int64)
    for label, lengths in (("shuffled", base), ("sorted desc", np.sort(base)[::-1].copy())):
        offsets = np.zeros(n_docs + 1, dtype=np.int64); np.cumsum(lengths, out=offsets[1:])
        n = int(offsets[-1])
        flat = torch.randint(0, n_sym, (n,), device="cuda", dtype=torch.int32)
        off_d = torch.from_numpy(offsets).cuda()
        for order in ((None, "sorted") if label == "shuffled" else (None,)):
            enc = B.ans_encode_ragged(flat, off_d, model, order=order)
            dec, st = B.ans_decode_ragged(enc, model, off_d)
            ok = bool(torch.equal(dec, flat))
            e = min(bench.event_ms(lambda: B.ans_encode_ragged(flat, off_d, model, order=order), 3) for _ in range(2))
            d = min(bench.event_ms(lambda: B.ans_decode_ragged(enc, model, off_d, out=dec), 3) for _ in range(2))
            print(f"{n_docs} docs {label}, order={order}: {n/1e6:.0f} M symbols encode {e:.3f} ms ({n/e/1e6:.0f} Gsym/s) decode {d:.3f} ms ({n/d/1e6:.0f} Gsym/s) ok={ok}", flush=True)
            del enc, dec
        del flat
