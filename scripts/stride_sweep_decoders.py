#!/usr/bin/env python3
"""Decode time of the two (32,64), P = 12 decoders -- ans_decode_kernel (lane chunk loads) and ans_decode_dq_kernel (lane-quad 64-byte
groups, CST_FLAG_COLD_WORDS) -- and encode time of the producer / consumer encoder at every slab stride from 97 to 160 x 64 bytes:
cache-resident words (decode follows encode) and words from HBM (a 1-GiB read in between).  65 536 x 4096, best of 4 launches."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n, k, P = 65536, 4096, 12
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
dec = torch.empty_like(sym)
flush = torch.empty(1 << 28, dtype=torch.int32, device="cuda")


def timed(fn, cold):
    best = 1e9
    for _ in range(4):
        if cold:
            flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print("# stride (x 64 bytes): encode | decode, words in cache: lane / quad | decode, words from HBM: lane / quad     (ms)")
rows = []
for units in range(97, 161):
    enc = B.ans_encode(sym, m, (32, 64, P), stride=16 * units)
    te = timed(lambda: B.ans_encode(sym, m, (32, 64, P), out=enc), False)
    r = [te]
    for cold in (False, True):
        for hint in (False, True):
            B.ans_decode(enc, m, k, out=dec, cold=hint)
            r.append(timed(lambda: B.ans_decode(enc, m, k, out=dec, cold=hint), cold))
    rows.append(r)
    print(f"{units} {r[0]:.3f} | {r[1]:.3f} {r[2]:.3f} | {r[3]:.3f} {r[4]:.3f}", flush=True)
a = np.array(rows)
names = ["encode", "decode hot lane", "decode hot quad", "decode cold lane", "decode cold quad"]
for i, nm in enumerate(names):
    print(f"# {nm}: min {a[:, i].min():.3f}  mean {a[:, i].mean():.3f}  max {a[:, i].max():.3f}  (max / min {a[:, i].max() / a[:, i].min():.2f})")
