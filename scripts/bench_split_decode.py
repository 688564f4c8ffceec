#!/usr/bin/env python3
"""Is the decoder's slowdown beyond 65 536 streams a property of the LAUNCH (grid size, workgroup order) or of the DATA
(footprint)?  Decodes 131 072 x 4096 as one call and as two calls over the halves of the same buffers."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
n, k = 131072, 4096
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, 12)
enc = B.ans_encode(sym, m, (32, 64, 12))
dec = torch.empty_like(sym)
h = n // 2
halves = [B.EncodedBatch(enc.words[i * h:(i + 1) * h], enc.n_words[i * h:(i + 1) * h], enc.status[i * h:(i + 1) * h], enc.config) for i in (0, 1)]
outs = [dec[:h], dec[h:]]
one = bench.event_ms(lambda: B.ans_decode(enc, m, k, out=dec), 10)
def two():
    B.ans_decode(halves[0], m, k, out=outs[0]); B.ans_decode(halves[1], m, k, out=outs[1])
t2 = bench.event_ms(two, 10)
first = bench.event_ms(lambda: B.ans_decode(halves[0], m, k, out=outs[0]), 10)
second = bench.event_ms(lambda: B.ans_decode(halves[1], m, k, out=outs[1]), 10)
print(f"one call {one:.3f} ms; two calls {t2:.3f} ms; first half alone {first:.3f}; second half alone {second:.3f}; ok={bool(torch.equal(dec, sym))}")
esym = [sym[:h], sym[h:]]
one = bench.event_ms(lambda: B.ans_encode(sym, m, (32, 64, 12), out=enc), 10)
def two_e():
    B.ans_encode(esym[0], m, (32, 64, 12), out=halves[0]); B.ans_encode(esym[1], m, (32, 64, 12), out=halves[1])
t2 = bench.event_ms(two_e, 10)
print(f"encode: one call {one:.3f} ms; two calls {t2:.3f} ms")

# the same 65 536-stream decode with the Infinity Cache (256 MiB, memory side) flushed before every call
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def cold(fn, reps=10):
    tot = 0.0
    for _ in range(reps):
        flush.fill_(1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps
print(f"65536 streams, cache flushed before each call: decode {cold(lambda: B.ans_decode(halves[0], m, k, out=outs[0])):.3f} ms, "
      f"encode {cold(lambda: B.ans_encode(esym[0], m, (32, 64, 12), out=halves[0])):.3f} ms")

# does a STREAMING read of the words (sequential, DRAM-friendly) before the decode bring them into the Infinity Cache?
def cold_then(pre, fn, reps=10):
    tot_pre = tot = 0.0
    for _ in range(reps):
        flush.fill_(1); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); pre(); e[1].record(); fn(); e[2].record(); torch.cuda.synchronize()
        tot_pre += e[0].elapsed_time(e[1]); tot += e[1].elapsed_time(e[2])
    return tot_pre / reps, tot / reps
w0 = halves[0].words
used = int(enc.n_words.max().item())
w0 = w0[:, : (used + 31) // 32 * 32]
pre, d = cold_then(lambda: w0.max(), lambda: B.ans_decode(halves[0], m, k, out=outs[0]))
print(f"flush, then a streaming read of the used part of the 65536 slabs ({w0.numel() * 4 / 1e6:.0f} MB): {pre:.3f} ms, then decode: {d:.3f} ms")

# the same input decoded into two output buffers alternately (the words stay resident, the output does not)
out_b = torch.empty_like(outs[0])
def alt():
    B.ans_decode(halves[0], m, k, out=outs[0]); B.ans_decode(halves[0], m, k, out=out_b)
print(f"same words, two output buffers alternately: {bench.event_ms(alt, 10) / 2:.3f} ms per decode (one output buffer: {first:.3f})")
# and two inputs into ONE output buffer
def alt_in():
    B.ans_decode(halves[0], m, k, out=out_b); B.ans_decode(halves[1], m, k, out=out_b)
print(f"two inputs alternately, one output buffer: {bench.event_ms(alt_in, 10) / 2:.3f} ms per decode")
