#!/usr/bin/env python3
"""Does the SIZE OF THE ALLOCATION a symbol matrix lives in change the kernels' speed?  C2 kernels (65 536 x 4096) on a
stand-alone 1 GiB tensor, on either half of a 2 GiB tensor, on quarters of a 4 GiB tensor."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


n, n_per = 65536, 4096
model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
g = torch.Generator(device="cuda").manual_seed(1)
base = torch.clamp(torch.round(torch.randn((n, n_per), generator=g, device="cuda") * 9.6 + 3.2), -50, 50).to(torch.int32)
for total in (1, 2, 4):
    big_in = torch.empty((total * n, n_per), dtype=torch.int32, device="cuda")
    big_out = torch.empty((total * n, n_per), dtype=torch.int32, device="cuda")
    for part in range(total):
        sym = big_in[part * n:(part + 1) * n]
        sym.copy_(base)
        out = big_out[part * n:(part + 1) * n]
        e, enc = timed(lambda: B.ans_encode(sym, model, (32, 64, 12)))
        d, _ = timed(lambda: B.ans_decode(enc, model, n_per, out=out))
        print(f"allocation of {total} GiB, part {part}: encode {e:.3f} ms  decode {d:.3f} ms  ok={bool(torch.equal(out, base))}"
              f"  (addresses {sym.data_ptr():#x} / {out.data_ptr():#x})")
    del big_in, big_out
    torch.cuda.empty_cache()
