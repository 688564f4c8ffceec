#!/usr/bin/env python3
"""Config C3 (one table per stream) with k jump points per stream: the checkpointing encoder against the plain one (same
words), the sub-lane decoder (two waves per SIMD, the lanes of a stream share its LDS table) against the plain decoder.
usage: bench_sublanes_c3.py [n_streams] [n_per]"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = (32, 64, 12)
mu_d, sigma_d = bench.c3_parameters(bench.SEED, 0, n, k, "cuda")
m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
sym = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, m3.cdfs_device(), 12)
enc = B.ans_encode(sym, m3, cfg)
dec = torch.empty_like(sym)
for rep in range(2):
    e = bench.event_ms(lambda: B.ans_encode(sym, m3, cfg, out=enc), 5)
    d = bench.event_ms(lambda: B.ans_decode(enc, m3, k, out=dec), 5)
    print(f"plain: encode {e:6.3f} ms decode {d:6.3f} ms ok={bool(torch.equal(dec, sym))}", flush=True)
for chunks in (2, 4, 8, 16):
    interval = k // chunks
    enc2, ck = B.ans_encode_checkpointed(sym, m3, interval, cfg)
    same = bool(torch.equal(enc2.n_words, enc.n_words)) and bool(torch.equal(enc2.words[:, :600], enc.words[:, :600]))
    dec.zero_()
    for rep in range(2):
        e = bench.event_ms(lambda: B.ans_encode_checkpointed(sym, m3, interval, cfg), 5)
        d = bench.event_ms(lambda: B.ans_decode_checkpointed(enc2, ck, m3, k, out=dec), 5)
        print(f"k={chunks} interval {interval}: encode {e:6.3f} ms (allocating) decode {d:6.3f} ms ok={bool(torch.equal(dec, sym))} same_words={same}", flush=True)
