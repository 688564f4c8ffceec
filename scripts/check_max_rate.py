import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from constriction_amd import batched as B
P = 12
for n in (4096, 2048):
    cdf = (np.arange(n + 1, dtype=np.uint32) * ((1 << P) // n)).astype(np.uint32)
    m = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(1)
    sym = torch.from_numpy(rng.integers(0, n, (256, 4096), dtype=np.int32)).cuda()
    for cfg in ((32, 64, P),):
        enc = B.ans_encode(sym, m, cfg)
        dec, st = B.ans_decode(enc, m, 4096)
        print(n, cfg, B.last_kernel(), "words/stream", float(enc.n_words.float().mean()), "ok", bool(torch.equal(dec, sym)), int(st.abs().sum()))
        dec, st = B.ans_decode(enc, m, 4096, cold=True)
        print(n, cfg, B.last_kernel(), "ok", bool(torch.equal(dec, sym)), int(st.abs().sum()))
        r = B.range_encode(sym, m, cfg); d2, s2 = B.range_decode(r, m, 4096)
        print(n, "range", B.last_kernel(), float(r.n_words.float().mean()), bool(torch.equal(d2, sym)), int(s2.abs().sum()))
