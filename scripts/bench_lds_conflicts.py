import os, sys
sys.path.insert(0, "/root/repo")
import torch
from constriction_amd import batched as B
n_streams, n_per = 65536, 4096
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
sym = torch.clamp(torch.round(z * 9.6 + 3.2), -50, 50).to(torch.int32)
same = sym[:1].expand(n_streams, n_per).contiguous()
per64 = sym[::64].repeat_interleave(64, dim=0).contiguous()      # the 64 streams of a wave are equal
for P in (12,):
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    for name, x in (("random", sym), ("all streams equal", same), ("equal within a wave", per64)):
        ms, enc = timed(lambda: B.range_encode(x, model, (32, 64, P)))
        ms2, enc2 = timed(lambda: B.ans_encode(x, model, (32, 64, P)))
        d1, _ = timed(lambda: B.range_decode(enc, model, n_per)); d2, _ = timed(lambda: B.ans_decode(enc2, model, n_per))
        print(f"{name:22s} range encode {ms:.3f} decode {d1:.3f}   ans encode {ms2:.3f} decode {d2:.3f}")
