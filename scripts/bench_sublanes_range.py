#!/usr/bin/env python3
"""Config C4 (range coder) with k jump points per stream: the checkpointing encoder against the plain one (same words), the
sub-lane decoder (two waves per SIMD) against the plain decoder.  usage: bench_sublanes_range.py [P ...]"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n, k = int(__import__("os").environ.get("STREAMS", 65536)), int(__import__("os").environ.get("NPER", 4096))
for P in [int(x) for x in sys.argv[1:]] or [12, 24]:
    cfg = (32, 64, P)
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
    sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
    enc = B.range_encode(sym, m, cfg)
    dec = torch.empty_like(sym)
    for rep in range(2):
        e = bench.event_ms(lambda: B.range_encode(sym, m, cfg, out=enc), 5)
        d = bench.event_ms(lambda: B.range_decode(enc, m, k, out=dec), 5)
        print(f"P={P} plain: encode {e:6.3f} ms decode {d:6.3f} ms ok={bool(torch.equal(dec, sym))}", flush=True)
    for chunks in (2, 4, 8):
        interval = k // chunks
        pair = B.range_encode_checkpointed(sym, m, interval, cfg)
        enc2, ck = pair
        same = bool(torch.equal(enc2.n_words, enc.n_words)) and bool(torch.equal(enc2.words[:, :400], enc.words[:, :400]))
        dec.zero_()
        st = torch.empty((n, chunks), dtype=torch.int32, device="cuda")
        for rep in range(2):
            e = bench.event_ms(lambda: B.range_encode_checkpointed(sym, m, interval, cfg, out=pair), 5)
            d = bench.event_ms(lambda: B.range_decode_checkpointed(enc2, ck, m, k, out=dec, status=st), 5)
            print(f"P={P} k={chunks} interval {interval}: encode {e:6.3f} ms decode {d:6.3f} ms ok={bool(torch.equal(dec, sym))} "
                  f"status0={int(st.abs().sum())==0} same_words={same}", flush=True)
