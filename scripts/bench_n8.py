#!/usr/bin/env python3
"""int8 symbol matrices at the headline shape: the loops that read / write int8 themselves (cst_ans_n8.hip) against the conversion
next to the int32 kernels (CST_NO_N8=1, cst_symbols.hip) and against the int32 call.
usage: bench_n8.py [n_streams] [n_per] [P]   (COLD=1: also after a 1-GiB fill)"""
import os
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 12
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym32 = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
sym8 = sym32.to(torch.int8)
stride = int(os.environ.get("STRIDE", 0)) or None
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda") if os.environ.get("COLD") else None


def cold_ms(fn):
    ts = []
    for _ in range(5):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def run(label, sym):
    enc = B.ans_encode(sym, m, (32, 64, P), stride=stride)
    ek = B.last_kernel()
    dec = torch.empty_like(sym)
    B.ans_decode(enc, m, k, out=dec)
    dk = B.last_kernel()
    es, ds = [], []
    for rep in range(6):
        es.append(bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, P), out=enc), 10))
        ds.append(bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 10))
    cold = ""
    if flush is not None:
        cold = f"  COLD encode {cold_ms(lambda: B.ans_encode(sym, m, (32, 64, P), out=enc)):.3f} decode {cold_ms(lambda: B.ans_decode(enc, m, k, out=dec)):.3f}"
    ok = bool(torch.equal(dec, sym)) and bool((enc.status == 0).all())
    print(f"{label:<28} {n} x {k} P={P} stride {enc.words.shape[1]}: encode min {min(es):.3f} med {np.median(es):.3f} [{ek}]  "
          f"decode min {min(ds):.3f} med {np.median(ds):.3f} [{dk}] ms  ok={ok}{cold}", flush=True)
    return enc


e32 = run("int32", sym32)
e8 = run("int8" + (" (CST_NO_N8)" if os.environ.get("CST_NO_N8") else ""), sym8)
same = bool(torch.equal(e32.n_words, e8.n_words))
w32, w8 = e32.words.cpu().numpy(), e8.words.cpu().numpy()
nw = e32.n_words.cpu().numpy()
same = same and all(np.array_equal(w32[s, : nw[s]], w8[s, : nw[s]]) for s in range(0, n, max(1, n // 512)))
print("words of the int8 call == words of the int32 call:", same)
