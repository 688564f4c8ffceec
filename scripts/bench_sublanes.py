#!/usr/bin/env python3
"""What k jump points per stream buy the decoders at the headline shape: the plain decode of 65 536 x 4096 against the
checkpointed decode of the SAME words as 65 536 * k virtual streams (Pos / Seek side information, stack.rs:1107-1139).
usage: bench_sublanes.py [P] [n_per]"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

P = int(sys.argv[1]) if len(sys.argv) > 1 else 12
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = 65536
cfg = (32, 64, P)
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
enc = B.ans_encode(sym, m, cfg)
dec = torch.empty_like(sym)
for rep in range(2):
    d = bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 5)
    print(f"plain ({P}) {k}: decode {d:6.3f} ms ok={bool(torch.equal(dec, sym))}", flush=True)
for chunks in (2, 4, 8):
    interval = k // chunks
    enc2, ck = B.ans_encode_checkpointed(sym, m, interval, cfg)
    same = bool(torch.equal(enc2.n_words, enc.n_words))
    dec.zero_()
    for rep in range(2):
        d = bench.event_ms(lambda: B.ans_decode_checkpointed(enc2, ck, m, k, out=dec), 5)
        print(f"k={chunks} interval {interval}: decode {d:6.3f} ms ok={bool(torch.equal(dec, sym))} same_counts={same}", flush=True)
