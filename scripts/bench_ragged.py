#!/usr/bin/env python3
"""Many small coders (the reference's tests/issue52.rs pattern): 100 000 documents of 20 .. 2000 symbols with one categorical
model at P = 24 through `batched.ans_{encode,decode}_ragged` (one launch each), beside the same documents one
`stream.stack.AnsCoder` each through the drop-in (a device round trip per call; 200 documents, extrapolated)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B
from constriction_amd.stream import model as M, stack

rng = np.random.default_rng(1)
n_docs, n_sym, P = 100_000, 64, 24
probs = rng.dirichlet(np.ones(n_sym) * 0.5)
single = M.Categorical(probs, perfect=False)
from oracle import oracle as O      # (the table only: test infrastructure is not timed)
cdf = O.categorical_fast_cdf(probs, P)
model = B.Model.from_cdf(cdf, 0, P)
for label, lengths in (("20 .. 2000 symbols, shuffled", np.exp(rng.uniform(np.log(20), np.log(2000), n_docs)).astype(np.int64)),
                       ("the same, sorted by length", None), ("200 symbols each", np.full(n_docs, 200, dtype=np.int64))):
    if lengths is None:
        lengths = np.sort(prev)
    prev = lengths
    offsets = np.zeros(n_docs + 1, dtype=np.int64); np.cumsum(lengths, out=offsets[1:])
    flat = torch.from_numpy(rng.choice(n_sym, size=int(offsets[-1]), p=probs).astype(np.int32)).cuda()
    off_d = torch.from_numpy(offsets).cuda()
    enc = B.ans_encode_ragged(flat, off_d, model)
    dec, st = B.ans_decode_ragged(enc, model, off_d)
    ok = bool(torch.equal(dec, flat)) and int(enc.status.abs().sum()) == 0
    e = min(bench.event_ms(lambda: B.ans_encode_ragged(flat, off_d, model), 5) for _ in range(3))
    d = min(bench.event_ms(lambda: B.ans_decode_ragged(enc, model, off_d, out=dec), 5) for _ in range(3))
    n = int(offsets[-1])
    print(f"{n_docs} documents, {label} ({n / 1e6:.1f} M symbols): encode {e:.3f} ms ({e * 1e3 / n_docs:.3f} us/doc, {n / e / 1e6:.1f} Gsym/s)  "
          f"decode {d:.3f} ms ({d * 1e3 / n_docs:.3f} us/doc, {n / d / 1e6:.1f} Gsym/s)  ok={ok}")
docs = [flat[offsets[s]: offsets[s + 1]].cpu().numpy() for s in range(200)]
torch.cuda.synchronize()
t0 = time.perf_counter()
words = []
for doc in docs:
    c = stack.AnsCoder(); c.encode_reverse(doc, single); words.append(c.get_compressed())
t1 = time.perf_counter()
for doc, w in zip(docs, words):
    assert np.array_equal(stack.AnsCoder(w).decode(single, len(doc)), doc)
t2 = time.perf_counter()
print(f"drop-in, one AnsCoder per document: encode {(t1 - t0) / 200 * 1e6:.0f} us/doc, decode {(t2 - t1) / 200 * 1e6:.0f} us/doc")
