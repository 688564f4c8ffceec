#!/usr/bin/env python3
"""Kernel times of API variants around the headline shape (65 536 streams x ~4096 symbols): where does a caller fall off the
hand-scheduled paths?  (HIP events; round trips checked.)"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

n = 65536
def model(P, lo=-50, hi=50):
    m = B.Model.quantized_gaussian(lo, hi, 3.2, 9.6, P)
    return m, torch.from_numpy(m.cdf().astype(np.int64)).cuda()

def run(name, coder, cfg, k, layout="stream_major", packed=False):
    m, cdf = model(cfg[2])
    sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, cfg[2])
    if layout == "symbol_major":
        sym = sym.t().contiguous()
    enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
    enc = enc_f(sym, m, cfg, layout)
    dec = torch.empty_like(sym)
    e = bench.event_ms(lambda: enc_f(sym, m, cfg, layout, out=enc), 5)
    if packed:
        pk, off = B.compact(enc)
        d = bench.event_ms(lambda: dec_f((pk, enc.n_words), m, k, layout, offsets=off, config=cfg, out=dec), 5)
    else:
        d = bench.event_ms(lambda: dec_f(enc, m, k, layout, out=dec), 5)
    print(f"{name:58s} encode {e:6.3f} ms  decode {d:6.3f} ms  ok={bool(torch.equal(dec, sym))}", flush=True)
    del sym, enc, dec
    torch.cuda.empty_cache()

if __name__ == "__main__":
  run("ANS (32,64,12) 4096 symbols [headline]", "ans", (32, 64, 12), 4096)
  run("ANS (32,64,12) 4095 symbols (rows not 16-byte aligned)", "ans", (32, 64, 12), 4095)
  run("ANS (32,64,12) 4100 symbols (ragged last tile)", "ans", (32, 64, 12), 4100)
  run("ANS (32,64,12) decode from the packed buffer", "ans", (32, 64, 12), 4096, packed=True)
  run("ANS (32,64,10)", "ans", (32, 64, 10), 4096)
  run("ANS (32,64,8)", "ans", (32, 64, 8), 4096)
  run("ANS (32,64,16)", "ans", (32, 64, 16), 4096)
  run("ANS (32,64,24) symbol-major", "ans", (32, 64, 24), 4096, "symbol_major")
  run("ANS (16,32,12) symbol-major", "ans", (16, 32, 12), 4096, "symbol_major")
  run("range (32,64,12) symbol-major", "range", (32, 64, 12), 4096, "symbol_major")
  run("range (32,64,12) decode from the packed buffer", "range", (32, 64, 12), 4096, packed=True)
  run("range (32,64,12) 4100 symbols", "range", (32, 64, 12), 4100)
  run("ANS (32,64,24) 4100 symbols", "ans", (32, 64, 24), 4100)
  run("ANS (16,32,12) 4100 symbols", "ans", (16, 32, 12), 4100)
  run("ANS (32,64,12) symbol-major", "ans", (32, 64, 12), 4096, "symbol_major")
  run("ANS (32,64,12) symbol-major, 4100 symbols", "ans", (32, 64, 12), 4100, "symbol_major")
