import sys, numpy as np, torch
sys.path.insert(0, '.')
from constriction_amd import batched as B
from oracle import oracle as O
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(1)
n_streams, n_per, P, lo, hi = int(sys.argv[1]), int(sys.argv[2]), 12, -127, 127
mu = -10 + 20 * rng.random(n_streams); sigma = np.exp(np.log(0.5) + rng.random(n_streams) * np.log(32))
print("creating model", flush=True)
model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sigma), P)
torch.cuda.synchronize(); print("model ok", model.n_tables, flush=True)
cdfs = np.stack([O.GaussianModel(lo, hi, m, s, P, 32).cdf_table() for m, s in zip(mu, sigma)])
print(model.cdf(0).tolist() == cdfs[0].tolist(), flush=True)
sym = O.synth_symbols(0xC0FFEE, 0, n_streams, n_per, lo, cdfs, P, per_stream_tables=True)
enc = B.ans_encode(dev(sym), model, (32, 64, P))
torch.cuda.synchronize(); print("encode ok", flush=True)
dec, st = B.ans_decode(enc, model, n_per)
torch.cuda.synchronize(); print("decode ok", np.array_equal(dec.cpu().numpy(), sym), flush=True)
