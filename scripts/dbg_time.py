import sys, torch, numpy as np
sys.path.insert(0, ".")
from constriction_amd import batched as B
n_streams, n_per = 65536, 4096
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
sym = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)
del z
model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
enc = B.ans_encode(sym, model, (32, 64, 12))
out = torch.empty_like(sym)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("decode with out=      ", t(lambda: B.ans_decode(enc, model, n_per, out=out)))
print("decode allocating     ", t(lambda: B.ans_decode(enc, model, n_per)))
print("decode with out= again", t(lambda: B.ans_decode(enc, model, n_per, out=out)))
print("empty 1GiB            ", t(lambda: torch.empty((n_streams, n_per), dtype=torch.int32, device="cuda")))
print("encode                ", t(lambda: B.ans_encode(sym, model, (32, 64, 12))))
print("encode with out=      ", t(lambda: B.ans_encode(sym, model, (32, 64, 12), out=enc)))
