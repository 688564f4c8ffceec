import sys, numpy as np, torch
sys.path.insert(0, ".")
from constriction_amd import batched as B
n_streams, n_per = 65536, 4096
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
rng = np.random.default_rng(3)
mu = torch.from_numpy(rng.uniform(-10, 10, n_streams)).cuda()
sigma = torch.from_numpy(np.exp(np.log(0.5) + rng.uniform(0, 1, n_streams) * np.log(32))).cuda()
sym3 = torch.clamp(torch.round(z * sigma.float()[:, None] + mu.float()[:, None]), -127, 127).to(torch.int32)
model = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sigma, 12)
for _ in range(3):
    enc = B.ans_encode(sym3, model, (32, 64, 12))
    dec, st = B.ans_decode(enc, model, n_per)
torch.cuda.synchronize()
print("ok", bool(torch.equal(dec, sym3)))
