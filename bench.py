#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch that is already resident in HBM:
ANS-encode 65 536 independent 4096-symbol streams (int32, stream-major) with one shared 12-bit
QuantizedGaussian(-50, 50, 3.2, 9.6) table into per-stream slabs, then ANS-decode them back
(BASELINE.json configs[1], SURVEY.md 8d "C2").  With N > 1 every rank (one process per GPU,
launched by torch.distributed.run) owns its own 65 536 streams (weak scaling, no data-path
collective); the RCCL gather of the packed words to rank 0 is timed separately and reported as
`gather_ms` (it is not part of `value`).

Rank 0 prints ONE JSON line.  `value` = symbols all ranks processed / max-over-ranks time of the K
timed steps (barrier + synchronize on both sides).  `roofline` is measured live with HIP events on
the launch stream around the dominant kernel; `cpu_baseline` times the CPU oracle ("port": the
repo's C restatement of the reference arithmetic, the Rust crate cannot be built here) on the
host cores of the same box, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SEED = 0xC0FFEE
LO, HI, MEAN, STD, P = -50, 50, 3.2, 9.6, 12
W, S = 32, 64
N_STREAMS, N_PER = 65536, 4096
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_symbols_device(seed, stream_begin, n_streams, n_per, lo, cdf_dev, precision, chunk=4096):
    """SURVEY.md 8(d) recipe on the GPU: q = splitmix64(seed ^ stream).next() >> (64-P); sym = quantile(q).
    Bit-identical to oracle.synth_symbols (checked in tests)."""
    dev = cdf_dev.device
    out = torch.empty((n_streams, n_per), dtype=torch.int32, device=dev)
    G = -7046029254386353131           # 0x9E3779B97F4A7C15 as int64
    C1 = -4658895280553007687          # 0xBF58476D1CE4E5B9
    C2 = -7723592293110705685          # 0x94D049BB133111EB
    t = torch.arange(1, n_per + 1, dtype=torch.int64, device=dev)[None, :] * G
    inner = cdf_dev[1:-1].to(torch.int64).contiguous()
    for a in range(0, n_streams, chunk):
        b = min(a + chunk, n_streams)
        sid = torch.arange(stream_begin + a, stream_begin + b, dtype=torch.int64, device=dev)
        z = (sid ^ seed)[:, None] + t
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * C1
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * C2
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        q = (z >> (64 - precision)) & ((1 << precision) - 1)
        idx = torch.searchsorted(inner, q, right=True)  # largest i with cdf[i] <= q
        out[a:b] = (idx + lo).to(torch.int32)
    return out


def cpu_baseline(cdf, symbols_host, repeats=3):
    """Times the CPU oracle (kind "port": the repo's C restatement of the reference arithmetic, -O3 -march=native,
    one disjoint block of streams per thread) on this box's host cores, on the SAME symbols the GPU coded."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    lut = O.lookup_from_cdf(cdf, P)

    def run(sym, threads):
        best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            words, n_words, status = O.ans_encode_batch(sym, LO, cdf, P, W, S, n_threads=threads, native=True)
            t1 = time.perf_counter()
            dec, dstatus = O.ans_decode_batch(words, n_words, N_PER, LO, cdf, P, W, S, lookup=lut, n_threads=threads, native=True)
            t2 = time.perf_counter()
            if best is None or (t2 - t0) < sum(best):
                best = (t1 - t0, t2 - t1)
        assert np.array_equal(dec, sym) and not status.any() and not dstatus.any()
        return sym.size, best[0], best[1]

    n, te, td = run(symbols_host, cores)
    n1, te1, td1 = run(symbols_host[: max(64, min(1024, len(symbols_host)))], 1)
    return {
        "value": round(n / (te + td) / 1e6, 2), "unit": "Msymbols/s", "cores": cores, "kind": "port",
        "sample": f"{len(symbols_host)} of {N_STREAMS} streams x {N_PER} symbols (the GPU's own input), {cores} threads, best of "
                  f"{repeats} (encode {n / te / 1e6:.0f} + decode {n / td / 1e6:.0f} Msym/s); "
                  f"1 thread: {n1 / (te1 + td1) / 1e6:.1f} Msym/s "
                  f"({te1 / n1 * 1e9:.1f} ns/sym encode, {td1 / n1 * 1e9:.1f} ns/sym decode)",
        "single_thread_value": round(n1 / (te1 + td1) / 1e6, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=N_STREAMS, help="streams per GPU (default: BASELINE config C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1 only: also time the RCCL gather of the packed words to rank 0 (reported as gather_ms, never "
                         "part of `value`; off by default so that the scaling run has no collective at all on its data path)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from constriction_amd import batched as B

    n_streams = args.streams
    model = B.Model.quantized_gaussian(LO, HI, MEAN, STD, P)
    cdf = model.cdf()
    cdf_dev = torch.from_numpy(cdf.astype(np.int64)).cuda()
    symbols = synth_symbols_device(SEED, rank * n_streams, n_streams, N_PER, LO, cdf_dev, P)
    enc = B.ans_encode(symbols, model, (W, S, P))          # allocates slabs / counts once
    decoded = torch.empty_like(symbols)
    torch.cuda.synchronize()

    def step():
        B.ans_encode(symbols, model, (W, S, P), out=enc)
        B.ans_decode(enc, model, N_PER, out=decoded)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel durations with HIP events on the launch stream (torch's current stream) ----
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    for k in range(args.steps):
        ev[k][0].record()
        B.ans_encode(symbols, model, (W, S, P), out=enc)
        ev[k][1].record()
        B.ans_decode(enc, model, N_PER, out=decoded)
        ev[k][2].record()
    torch.cuda.synchronize()
    enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    dec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    total_words = enc.total_words()
    n_sym = n_streams * N_PER
    # algorithmic bytes per launch: 4 B per int32 symbol + 4 B per compressed word, each way
    bytes_per_launch = 4 * n_sym + 4 * total_words
    dominant, dom_ms = ("ans_encode_kernel", enc_ms) if enc_ms >= dec_ms else ("ans_decode_kernel", dec_ms)
    achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9

    ok = True
    if not args.no_check:
        ok = bool(torch.equal(decoded, symbols)) and int(enc.status.abs().sum().item()) == 0
        if rank == 0:
            from oracle import oracle as O
            sample = [0, 1, n_streams // 2, n_streams - 1]
            host = symbols[sample].cpu().numpy()
            ww, wn, _ = O.ans_encode_batch(host, LO, cdf, P, W, S)
            for k, s in enumerate(sample):
                ok = ok and enc.stream(s).tolist() == ww[k, : wn[k]].tolist()

    # ---- compaction and (N > 1) gather of the packed words to rank 0, timed separately ----
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    packed, offsets = B.compact(enc)
    e1.record()
    torch.cuda.synchronize()
    compact_ms = e0.elapsed_time(e1)
    gather_ms = None
    if dist is not None and args.gather:
        from constriction_amd import dist as D
        sync_all()
        g0 = time.perf_counter()
        D.gather_packed(packed, offsets, dst=0)
        sync_all()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        traffic = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get(dominant, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Msymbols/s encode+decode, 64k x 4k-symbol streams, bit-exact vs CPU",
            "value": round(world * n_sym * args.steps / elapsed / 1e6, 1),
            "unit": "Msymbols/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"C2: {n_streams} streams/GPU x {N_PER} int32 symbols (stream-major), shared "
                                   f"12-bit QuantizedGaussian({LO},{HI},{MEAN},{STD}), AnsCoder (W,S,P)=({W},{S},{P}), "
                                   "encode into slabs + decode; u64 coder state, u32 words, i32 symbols", "streams_per_gpu": n_streams, "symbols_per_stream": N_PER,
                       "parallelism": f"streams sharded over {world} GPU(s), no data-path collective"},
            "bit_exact": ok,
            "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "compact_ms": round(compact_ms, 4),
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "words_per_stream": round(total_words / n_streams, 2),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(dom_ms, 4),
                         "encode_GBps": round(bytes_per_launch / (enc_ms * 1e-3) / 1e9, 1),
                         "decode_GBps": round(bytes_per_launch / (dec_ms * 1e-3) / 1e9, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cdf, symbols.cpu().numpy())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bit-exactness check FAILED")


if __name__ == "__main__":
    main()
