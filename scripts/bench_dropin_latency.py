import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import constriction_amd as constriction
from constriction_amd.stream import stack, model as M
m = M.QuantizedGaussian(-50, 50, 3.2, 9.6)
sym = np.random.default_rng(1).integers(-20, 20, 1000).astype(np.int32)
c = stack.AnsCoder()
c.encode_reverse(sym, m); torch.cuda.synchronize()
for n in (1, 100, 1000):
    t=time.time()
    for _ in range(50): c.encode_reverse(sym[:n], m)
    torch.cuda.synchronize(); e=(time.time()-t)/50
    t=time.time()
    for _ in range(50): c.decode(m, n)
    torch.cuda.synchronize(); d=(time.time()-t)/50
    print(f"n={n:5d}: encode_reverse {e*1e6:7.1f} us/call, decode {d*1e6:7.1f} us/call")
means = np.linspace(-10,10,1000); stds = np.full(1000, 5.0)
fam = M.QuantizedGaussian(-50, 50)
t=time.time()
for _ in range(50): c.encode_reverse(sym, fam, means, stds)
torch.cuda.synchronize(); print(f"per-symbol params n=1000: encode {(time.time()-t)/50*1e6:.1f} us/call")
