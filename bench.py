#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch that is already resident in HBM:
ANS-encode 65 536 independent 4096-symbol streams (int32, stream-major) with one shared 12-bit
QuantizedGaussian(-50, 50, 3.2, 9.6) table into per-stream slabs, then ANS-decode them back
(BASELINE.json configs[1], SURVEY.md 8d "C2").  With N > 1 every rank (one process per GPU) owns its own
65 536 streams (weak scaling, no data-path collective).  `python bench.py --gpus N` starts its own ranks
(re-executes itself under torch.distributed.run on 127.0.0.1) unless a launcher already did.

Rank 0's LAST stdout line is the bench line: the contract's keys only, under 4 KB (contract_line below).  Everything
else -- `configs`, `after_cache_flush`, `end_to_end`, `rate` -- is the detail record: bench_detail.json next to this script
and the stdout line before the last one.  `value` = symbols all ranks processed / max-over-ranks time of the K
timed steps (barrier + synchronize on both sides).  `roofline` is measured live with HIP events on
the launch stream around the dominant kernel; `cpu_baseline` times the CPU oracle ("port": the
repo's C restatement of the reference arithmetic, the Rust crate cannot be built here) on the
host cores of the same box.  The `configs` block carries, on the same clock, every other single-GPU
configuration of BASELINE.json (the 16-bit-word and 24-bit presets of C2, C3 = one table per stream, C4 = range
coder at P = 12 / 24, f1 = every symbol its own f64 (mean, std) -- the reference's flagship Python call, batched --, the C5
shard of 131 072 streams per GPU with compaction and -- N > 1 -- the RCCL gather of the packed words to rank 0):
encode_ms / decode_ms / achieved fraction of the HBM roofline / bit_exact, where bit_exact compares EVERY stream's words
with the CPU oracle's and the decoded symbols with the input.  With N > 1 the line also carries `per_rank` (every rank's
own encode_ms / decode_ms) and `rccl_ranks_seen`, so that a scaling run diagnoses itself.
"""
import argparse
import json
import os
import socket
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SEED = 0xC0FFEE
LO, HI, MEAN, STD, P = -50, 50, 3.2, 9.6, 12
W, S = 32, 64
N_STREAMS, N_PER = 65536, 4096
C5_STREAMS = 131072       # per GPU: 1 048 576 streams over 8 GPUs
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_symbols_device(seed, stream_begin, n_streams, n_per, lo, cdf_dev, precision, chunk=4096):
    """SURVEY.md 8(d) recipe on the GPU: q = splitmix64(seed ^ stream).next() >> (64-P); sym = quantile(q).
    Bit-identical to oracle.synth_symbols (tests/test_gpu_ans_batch.py::test_bench_symbols_match_the_oracle)."""
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


def splitmix_draws(seed, stream_begin, n_streams, first, count, dev):
    """draws first .. first + count - 1 (1-based, as oracle.synth_symbols counts them) of the per-stream generators
    splitmix64(seed ^ stream_id): int64 tensor [n_streams, count] holding the 64-bit outputs"""
    G, C1, C2 = -7046029254386353131, -4658895280553007687, -7723592293110705685
    t = torch.arange(first, first + count, dtype=torch.int64, device=dev)[None, :] * G
    sid = torch.arange(stream_begin, stream_begin + n_streams, dtype=torch.int64, device=dev)
    z = (sid ^ seed)[:, None] + t
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * C1
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * C2
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def unit_draws(z):
    """u = (x >> 11) * 2^-53 in [0, 1): the 53-bit mantissa draw of SURVEY.md 8(d)"""
    return ((z >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))


def c3_parameters(seed, stream_begin, n_streams, n_per, dev):
    """mu_s = -10 + 20 u1, sigma_s = exp(ln 0.5 + u2 ln 32) with u1, u2 the two draws of stream s's
    generator that follow its n_per symbol draws"""
    u = unit_draws(splitmix_draws(seed, stream_begin, n_streams, n_per + 1, 2, dev))
    return -10.0 + 20.0 * u[:, 0], torch.exp(float(np.log(0.5)) + u[:, 1] * float(np.log(32.0)))


def synth_symbols_per_stream(seed, stream_begin, n_per, lo, cdf_rows, precision, chunk=2048):
    """the 8(d) recipe under one table per stream: sym = quantile_function_s(draw_t >> (64 - P)); cdf_rows int32/int64
    [n_streams, n + 1] on the device"""
    n_streams = cdf_rows.shape[0]
    dev = cdf_rows.device
    out = torch.empty((n_streams, n_per), dtype=torch.int32, device=dev)
    inner = cdf_rows[:, 1:-1].to(torch.int64).contiguous()
    for a in range(0, n_streams, chunk):
        b = min(a + chunk, n_streams)
        z = splitmix_draws(seed, stream_begin + a, b - a, 1, n_per, dev)
        q = (z >> (64 - precision)) & ((1 << precision) - 1)
        out[a:b] = (torch.searchsorted(inner[a:b], q, right=True) + lo).to(torch.int32)
    return out


def cpu_quota():
    """CPUs' worth of time the container may use per scheduling period (cgroup cpu.max / cfs_quota_us), or None if unlimited.
    The GPU boxes of this pool show 256 logical CPUs and a quota of 16: a pass of 256 threads spends the period's 1.6 CPU-seconds
    in ~6 ms and is then SUSPENDED until the next 100-ms period -- which is what round 4's "decode scales to 4 % of ideal" was
    (scripts/cpu_scaling.py, profiles/r05_cpu_scaling.txt: up to 32 threads the port scales at 0.94 - 0.99 per thread, the decoder
    like the encoder; beyond the quota a pass runs at whatever is left of the period's budget, 3 - 21 Gsym/s from run to run)."""
    try:
        p = Path("/sys/fs/cgroup/cpu.max")
        if p.exists():
            quota, period = p.read_text().split()[:2]
            return None if quota == "max" else float(quota) / float(period)
        q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        return None if q <= 0 else q / int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
    except (OSError, ValueError):
        return None


def host_cores():
    """host threads this rank may use for the oracle checks: the box's cores shared among the ranks of the job"""
    return max(1, (os.cpu_count() or 1) // max(1, int(os.environ.get("WORLD_SIZE", "1"))))


def baseline_threads():
    """threads of the cpu_baseline leg: the logical CPUs the container can keep busy (its CPU quota, if it has one)"""
    q = cpu_quota()
    n = host_cores()
    return n if q is None else max(1, min(n, int(q + 0.5)))


def _blocks(n, parts):
    step = max(1, (n + parts - 1) // parts)
    return [(a, min(a + step, n)) for a in range(0, n, step)]


def cpu_words_match(kind, sym_host, gpu_words, gpu_n_words, lo, cdf, precision, w_bits, s_bits):
    """Encodes EVERY stream with the CPU oracle (one block of streams per host thread; the C code runs without the
    GIL) and compares word counts and words with the GPU's slabs.  cdf: [n+1] shared or [n_streams][n+1]."""
    from oracle import oracle as O
    cores = host_cores()
    per_stream = cdf.ndim == 2

    def work(ab):
        a, b = ab
        c = cdf[a:b] if per_stream else cdf
        if kind == "ans":
            words, n_words, status = O.ans_encode_batch(sym_host[a:b], lo, c, precision, w_bits, s_bits)
        else:
            words, n_words, status = O.rc_encode_batch(sym_host[a:b], lo, c, precision, w_bits, s_bits)
        if status.any() or not np.array_equal(n_words, gpu_n_words[a:b]):
            return False
        width = min(words.shape[1], gpu_words.shape[1])
        if int(n_words.max(initial=0)) > width:
            return False
        mask = np.arange(width, dtype=np.uint32)[None, :] < n_words[:, None]
        return bool(np.array_equal(np.where(mask, words[:, :width], 0), np.where(mask, gpu_words[a:b, :width], 0)))

    with ThreadPoolExecutor(max_workers=cores) as pool:
        return all(pool.map(work, _blocks(len(sym_host), 4 * cores)))


def cpu_tables(lo, hi, mu, sigma, precision):
    """one quantized-Gaussian cdf per stream from the oracle (host threads)"""
    from oracle import oracle as O
    cores = host_cores()

    def work(ab):
        return np.stack([O.GaussianModel(lo, hi, float(m), float(s), precision, 32).cdf_table() for m, s in zip(mu[ab[0]:ab[1]], sigma[ab[0]:ab[1]])])

    with ThreadPoolExecutor(max_workers=cores) as pool:
        return np.concatenate(list(pool.map(work, _blocks(len(mu), 2 * cores))))


def cpu_baseline(cdf, symbols_host, repeats=3):
    """Times the CPU oracle (kind "port": the repo's C restatement of the reference arithmetic, -O3 -march=native,
    one disjoint block of streams per thread) on this box's host cores, on the SAME symbols the GPU coded.
    Output buffers are allocated once and written by an untimed first pass (each thread touches its own block first, pinned to
    one logical CPU: the pages of a block live on its thread's NUMA node), so a timed pass measures coding, not the kernel's
    page-fault path (round 3 allocated a gigabyte of zero pages inside the timed region: all-core decode ran at half the
    encoder's rate although one thread decodes as fast as it encodes)."""
    from oracle import oracle as O
    cores = baseline_threads()
    lut = O.lookup_from_cdf(cdf, P)

    def run(sym, threads):
        os.environ["CST_ORACLE_PIN"] = "1" if threads > 1 else "0"
        enc_out = O.ans_encode_batch(sym, LO, cdf, P, W, S, n_threads=threads, native=True)                       # untimed: first touch
        dec_out = O.ans_decode_batch(enc_out[0], enc_out[1], N_PER, LO, cdf, P, W, S, lookup=lut, n_threads=threads, native=True)
        best_e = best_d = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            words, n_words, status = O.ans_encode_batch(sym, LO, cdf, P, W, S, n_threads=threads, native=True, out=enc_out)
            t1 = time.perf_counter()
            dec, dstatus = O.ans_decode_batch(words, n_words, N_PER, LO, cdf, P, W, S, lookup=lut, n_threads=threads, native=True, out=dec_out)
            t2 = time.perf_counter()
            best_e = (t1 - t0) if best_e is None else min(best_e, t1 - t0)
            best_d = (t2 - t1) if best_d is None else min(best_d, t2 - t1)
        os.environ["CST_ORACLE_PIN"] = "0"
        assert np.array_equal(dec, sym) and not status.any() and not dstatus.any()
        return sym.size, best_e, best_d

    n, te, td = run(symbols_host, cores)
    n1, te1, td1 = run(symbols_host[: max(64, min(2048, len(symbols_host)))], 1)
    enc_rate, dec_rate, enc1, dec1 = n / te / 1e6, n / td / 1e6, n1 / te1 / 1e6, n1 / td1 / 1e6
    return {
        "value": round(n / (te + td) / 1e6, 2), "unit": "Msymbols/s", "cores": cores, "kind": "port",
        "sample": f"{len(symbols_host)} of {N_STREAMS} streams x {N_PER} symbols (the GPU's own input), {cores} pinned threads, best of {repeats}",
        "sample_detail": f"buffers allocated and first touched outside the timed region (encode {enc_rate:.0f} + decode {dec_rate:.0f} Msym/s); "
                         f"1 thread: {n1 / (te1 + td1) / 1e6:.1f} Msym/s "
                         f"({te1 / n1 * 1e9:.1f} ns/sym encode, {td1 / n1 * 1e9:.1f} ns/sym decode)",
        "single_thread_value": round(n1 / (te1 + td1) / 1e6, 2),
        "encode_Msymbols_per_s": round(enc_rate, 1), "decode_Msymbols_per_s": round(dec_rate, 1),
        "scaling_efficiency": {"encode": round(enc_rate / (cores * enc1), 3), "decode": round(dec_rate / (cores * dec1), 3),
                               "what": f"{cores}-thread rate / ({cores} x the one-thread rate)"},
        "logical_cpus": os.cpu_count(), "cgroup_cpu_quota": cpu_quota(),
        "threads_note": "threads = min(logical CPUs, the container's CPU quota): more threads than the quota are suspended by the scheduler "
                        "for the rest of every 100-ms period once its budget is spent (profiles/r05_cpu_scaling.txt); on a box without a "
                        "quota every logical CPU is used",
    }


def model_entropy_bits(cdf, precision):
    """entropy_base2 of a quantized model (src/stream/model.rs:576-591): the bit rate an optimal coder reaches on symbols drawn
    from the model itself, which is how every batch here is drawn"""
    p = np.diff(np.asarray(cdf, dtype=np.float64)) / float(1 << precision)
    p = p[p > 0]
    return float(-(p * np.log2(p)).sum())


def rate_report(total_words, n_sym, word_bits, entropy_bits):
    """bits per symbol of the compressed words (final state words included, as in get_compressed) against the model's entropy:
    the reference's own headline figure (README: "bit-rate overhead")"""
    bps = word_bits * total_words / n_sym
    return {"bits_per_symbol": round(bps, 5), "model_entropy_bits": round(entropy_bits, 5),
            "overhead_vs_entropy": round(bps / entropy_bits - 1.0, 6)}


def event_ms(fn, reps):
    """typical duration of fn() in ms (the MEDIAN of `reps` launches timed one by one: on a shared box one launch in a hundred
    is preempted for milliseconds, and a mean of five would report that), HIP events on torch's current stream (the launch
    stream of the library calls).  The bench line itself is the mean over its timed region, as the contract says.
    The launches are preceded by 30 ms of the same work: after seconds of host-side checking the GPU sits at its idle clocks,
    and five launches right away measure the ramp (C3 encode 0.42 instead of 0.39 ms, scripts/c3_in_bench.py)."""
    t0 = time.perf_counter()
    while True:
        fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 0.03:
            break
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))


def run_config(B, name, coder, cfg, model, symbols, reps, check, cdf_host=None, lo=LO, layout="stream_major", stride=None, packed16=False):
    """One entry of the `configs` block: kernel times of encode and decode (HIP events), achieved fraction of the HBM
    roofline (algorithmic bytes 4 B / symbol + 4 B / word per direction), bit-exactness of every stream.
    layout "symbol_major": `symbols` is [n_per, n_streams]."""
    n_streams, n_per = symbols.shape if layout == "stream_major" else symbols.shape[::-1]
    enc_fn = B.ans_encode if coder == "ans" else B.range_encode
    dec_fn = B.ans_decode if coder == "ans" else B.range_decode
    kw = {} if layout == "stream_major" else {"layout": layout}
    if packed16:       # CST_FLAG_PACKED_W16: two 16-bit words per slot (the stride tuner measures the unpacked form: default stride here)
        enc = enc_fn(symbols, model, cfg, packed16=True)
    else:
        enc = enc_fn(symbols, model, cfg, stride=stride, **kw)      # stride "tuned": batched.tuned_stride, like the headline batch
    decoded = torch.empty_like(symbols)
    enc_ms = event_ms(lambda: enc_fn(symbols, model, cfg, out=enc, **kw), reps)
    enc_kernel = B.last_kernel()
    dec_ms = event_ms(lambda: dec_fn(enc, model, n_per, out=decoded, **kw), reps)
    dec_kernel = B.last_kernel()
    total_words = enc.total_words()
    n_sym = n_streams * n_per
    byts = 4 * n_sym + (cfg[0] // 8) * total_words       # algorithmic: W-bit words (16-bit words sit in 32-bit slots: the TRAFFIC is larger)
    entry = {
        "workload": name, "coder": coder, "config": list(cfg), "streams": n_streams, "symbols_per_stream": n_per,
        "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "encode_kernel": enc_kernel, "decode_kernel": dec_kernel,
        # the calls above are the DEFAULT calls (jump_points="auto"): how many jump points per stream the library took for this batch
        "jump_points": int(enc.jump.pos.shape[1]) if enc.jump is not None else 0,
        "Msymbols_per_s": round(n_sym / (enc_ms + dec_ms) / 1e3, 1), "words_per_stream": round(total_words / n_streams, 2),
        "slab_stride_words": int(enc.words.shape[1]),
        "encode_frac": round(byts / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "decode_frac": round(byts / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
    }
    if cdf_host is not None and np.asarray(cdf_host).ndim == 1:
        entry.update(rate_report(total_words, n_sym, cfg[0], model_entropy_bits(cdf_host, cfg[2])))
    plain_ok = True
    if enc.jump is not None:
        # for the record: the same batch WITHOUT jump points (jump_points=0: what the default call was before round 6); same words
        plain = enc_fn(symbols, model, cfg, stride=int(enc.words.shape[1]), jump_points=0, **kw)
        plain_dec = torch.empty_like(symbols)
        pe = event_ms(lambda: enc_fn(symbols, model, cfg, out=plain, jump_points=0, **kw), reps)
        pd = event_ms(lambda: dec_fn(plain, model, n_per, out=plain_dec, **kw), reps)
        entry["without_jump_points"] = {"encode_ms": round(pe, 4), "decode_ms": round(pd, 4), "decode_kernel": B.last_kernel(),
                                        "decode_speedup_of_the_default": round(pd / dec_ms, 3)}
        if check:
            used = torch.arange(plain.words.shape[1], device=symbols.device)[None, :] < plain.n_words[:, None]
            plain_ok = bool(torch.equal(plain_dec, symbols)) and bool(torch.equal(plain.n_words, enc.n_words)) and \
                bool(((plain.words == enc.words) | ~used).all())
        del plain, plain_dec
    if coder == "ans" and (packed16 or layout != "stream_major"):
        # Pos / Seek on the two batch forms whose fast encoders do not note jump points on their way (round 6): the table comes from the
        # generic checkpointing encoder (one lane per stream; for packed words: run beside the packed encoder), decode = the plain
        # batched decoder on the chunks as streams of their own.  Asked for explicitly -- "auto" takes none here; the words are the same
        try:
            ej = enc_fn(symbols, model, cfg, jump_points=2, **({"packed16": True} if packed16 else kw))
            jdec = torch.empty_like(symbols)
            je = event_ms(lambda: enc_fn(symbols, model, cfg, out=ej, jump_points=2, **kw), max(2, reps // 4))
            jek = B.last_kernel()
            jd = event_ms(lambda: dec_fn(ej, model, n_per, out=jdec, **kw), reps)
            sub = {"encode_ms": round(je, 4), "decode_ms": round(jd, 4), "encode_kernel": jek, "decode_kernel": B.last_kernel(),
                   "what": "explicit jump_points=2 (random access, stack.rs:1107-1139): the jump table from the generic checkpointing encoder"
                           + (" run beside the packed encoder" if packed16 else "") + ", decode = the plain batched decoder on the chunks with raw states"}
            if check:
                used = torch.arange(min(ej.words.shape[1], enc.words.shape[1]), device=symbols.device)[None, :] < ej.n_words[:, None]
                w = used.shape[1]
                jok = bool(torch.equal(jdec, symbols)) and bool(torch.equal(ej.n_words, enc.n_words)) and bool(((ej.words[:, :w] == enc.words[:, :w]) | ~used).all())
                if jok and cdf_host is not None:
                    from oracle import oracle as O
                    rows = [0, 1, n_streams // 2, n_streams - 1]
                    hs = symbols[rows].cpu().numpy() if layout == "stream_major" else np.ascontiguousarray(symbols[:, rows].cpu().numpy().T)
                    wp, ws = O.ans_jump_table(hs, lo, np.asarray(cdf_host, dtype=np.uint32), cfg[2], n_per // 2, W=cfg[0], S=cfg[1])
                    jok = np.array_equal(ej.jump.pos[rows].cpu().numpy().view(np.uint32), wp) and np.array_equal(ej.jump.state[rows].cpu().numpy().view(np.uint64), ws)
                sub["bit_exact"] = bool(jok)
                plain_ok = plain_ok and bool(jok)
            entry["with_2_jump_points"] = sub
            del ej, jdec
        except Exception as exc:      # noqa: BLE001
            entry["with_2_jump_points"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            plain_ok = False
    if check:
        ok = plain_ok and bool(torch.equal(decoded, symbols)) and int(enc.status.abs().sum().item()) == 0
        if ok and cdf_host is not None:
            words, n_words, _ = enc.to_numpy()
            host_sym = symbols.cpu().numpy() if layout == "stream_major" else np.ascontiguousarray(symbols.cpu().numpy().T)
            ok = cpu_words_match(coder, host_sym, words, n_words, lo, cdf_host, cfg[2], cfg[0], cfg[1])
            entry["bit_exact_scope"] = f"all {n_streams} streams: words and counts vs CPU oracle, decoded symbols vs input"
        entry["bit_exact"] = ok
    return entry, enc


def jump_config(B, name, coder, cfg, model, symbols, reps, check, cdf_host=None, lo=LO, ks=(2, 4)):
    """Decode with k JUMP POINTS per stream (the reference's Pos / Seek side information, stack.rs:1107-1139, queue.rs:172-196,
    900-926): the encoder notes them on its way, the decoder runs k lanes per stream -- two resident waves per SIMD where the
    plain decoder of a 65 536-stream batch has one.  The words are the plain encoder's (compared), the jump tables of sampled
    streams are compared with the CPU oracle's, every chunk's symbols with the input.  Algorithmic bytes as for the plain entry
    (the jump table adds 12 k or 20 k bytes to a stream's 16 KiB of symbols)."""
    from oracle import oracle as O
    n_streams, n_per = symbols.shape
    ans = coder == "ans"
    enc_ck = B.ans_encode_checkpointed if ans else B.range_encode_checkpointed
    dec_ck = B.ans_decode_checkpointed if ans else B.range_decode_checkpointed
    plain = (B.ans_encode if ans else B.range_encode)(symbols, model, cfg, jump_points=0)
    decoded = torch.empty_like(symbols)
    plain_dec_ms = event_ms(lambda: (B.ans_decode if ans else B.range_decode)(plain, model, n_per, out=decoded), reps)
    plain_kernel = B.last_kernel()
    total_words = plain.total_words()
    byts = symbols.element_size() * n_streams * n_per + (cfg[0] // 8) * total_words
    entry = {"workload": name, "coder": coder, "config": list(cfg), "streams": n_streams, "symbols_per_stream": n_per,
             "symbol_bytes": symbols.element_size(), "plain_decode_ms": round(plain_dec_ms, 4), "plain_decode_kernel": plain_kernel, "by_k": {}}
    ok = True
    for k in ks:
        interval = n_per // k
        pair = enc_ck(symbols, model, interval, cfg)
        enc, ck = pair
        status = torch.empty((n_streams, k), dtype=torch.int32, device=symbols.device)
        decoded.zero_()
        e_ms = event_ms(lambda: enc_ck(symbols, model, interval, cfg, out=pair), reps)
        e_kernel = B.last_kernel()
        d_ms = event_ms(lambda: dec_ck(enc, ck, model, n_per, out=decoded, status=status), reps)
        entry["by_k"][str(k)] = {"encode_ms": round(e_ms, 4), "decode_ms": round(d_ms, 4), "kernels": [e_kernel, B.last_kernel()],
                                 "decode_frac": round(byts / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                 "decode_speedup": round(plain_dec_ms / d_ms, 3)}
        if check:
            good = bool(torch.equal(decoded, symbols)) and int(status.abs().sum().item()) == 0 and int(enc.status.abs().sum().item()) == 0
            good = good and bool(torch.equal(enc.n_words, plain.n_words))
            w = min(int(plain.words.shape[1]), int(enc.words.shape[1]), 512)
            used = torch.arange(w, device=symbols.device)[None, :] < plain.n_words[:, None]  # (the first 512 words of every stream)
            good = good and bool(((enc.words[:, :w] == plain.words[:, :w]) | ~used).all())
            if good and cdf_host is not None:
                rows = sorted({0, 1, 63, 64, n_streams // 2, n_streams - 1})
                host = symbols[rows].cpu().numpy().astype(np.int32)
                tabs = np.asarray(cdf_host)
                tab = tabs[rows] if tabs.ndim == 2 else tabs
                if ans:
                    pos, state = O.ans_jump_table(host, lo, tab, cfg[2], interval)
                    good = np.array_equal(ck.pos[rows].cpu().numpy().view(np.uint32), pos) and \
                        np.array_equal(ck.state[rows].cpu().numpy().view(np.uint64), state)
                else:
                    pos, lower, rng = O.range_jump_table(host, lo, tab, cfg[2], interval)
                    good = np.array_equal(ck.pos[rows].cpu().numpy().view(np.uint32), pos) and \
                        np.array_equal(ck.lower[rows].cpu().numpy().view(np.uint64), lower) and \
                        np.array_equal(ck.range[rows].cpu().numpy().view(np.uint64), rng)
            ok = ok and bool(good)
        del pair, enc, ck
    if check:
        entry["bit_exact"] = ok
        entry["bit_exact_scope"] = ("every chunk's symbols vs input, counts and the first 512 words of every stream vs the plain encoder's (themselves "
                                    "compared with the CPU oracle in the plain entry), jump tables of six streams vs the CPU oracle")
    return entry


def narrow_config(B, model, symbols, reps, check, dtype=torch.int8, cfg=None, name="C2", coder="ans"):
    """C2 with a NARROW symbol matrix (the reference's Symbol type is generic, quantize.rs:229-255; C2's alphabet fits int8).
    int8 / int16 (round 5): the hand-scheduled loops read / write the narrow matrix themselves (ans_encode_pc_n8_kernel / ans_decode_n8_kernel
    and their n16 forms: cst_ans_pc.hip, cst_ans_n8.hip) -- algorithmic bytes 1 or 2 B per symbol + 4 B per word each way, and that is the traffic; the
    kernels are bound by instruction issue, not by HBM (one wave per SIMD: DESIGN.md 4.13), so their HBM fractions are low by
    construction.  `conversion_path`: the same call with CST_NO_N8=1 -- widened / narrowed by a streaming kernel next to the int32
    coder kernels (what the shapes the native kernels do not take -- symbol-major, other presets, rows that are not whole lines -- still use)."""
    import os
    enc_fn, dec_fn = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)      # (C4: round 6, cst_range_*_batch_sym)
    n_streams, n_per = symbols.shape
    cfg = cfg or (W, S, P)
    narrow = symbols.to(dtype)
    nb = narrow.element_size()
    from constriction_amd import _native
    enc = enc_fn(narrow, model, cfg)
    enc_kernel = B.last_kernel()
    decoded = torch.empty_like(narrow)
    dec_fn(enc, model, n_per, out=decoded)
    dec_kernel = B.last_kernel()
    enc_ms = event_ms(lambda: enc_fn(narrow, model, cfg, out=enc), reps)
    dec_ms = event_ms(lambda: dec_fn(enc, model, n_per, out=decoded), reps)
    total_words = enc.total_words()
    byts = nb * n_streams * n_per + 4 * total_words
    native = "n8_kernel" in enc_kernel or "n16_kernel" in enc_kernel
    entry = {"workload": f"{name} with {str(dtype).replace('torch.', '')} symbol matrices " +
                         ("(read / written by the coder loops themselves)" if native else "(widened / narrowed on the device next to the coder call)"),
             "coder": coder, "config": list(cfg), "streams": n_streams, "symbols_per_stream": n_per, "symbol_bytes": nb,
             "encode_kernel": enc_kernel, "decode_kernel": dec_kernel, "jump_points": int(enc.jump.pos.shape[1]) if enc.jump is not None else 0,
             "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "Msymbols_per_s": round(n_streams * n_per / (enc_ms + dec_ms) / 1e3, 1),
             "slab_stride_words": int(enc.words.shape[1]),
             "algorithmic_bytes_per_symbol": round(byts / (n_streams * n_per), 3),
             "encode_frac": round(byts / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "decode_frac": round(byts / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    if native:
        entry["bound"] = "instruction issue of one wave per SIMD (not HBM): see DESIGN.md 4.13"
    if enc.jump is not None:
        plain8 = enc_fn(narrow, model, cfg, jump_points=0)
        pe = event_ms(lambda: enc_fn(narrow, model, cfg, out=plain8, jump_points=0), reps)
        pd = event_ms(lambda: dec_fn(plain8, model, n_per, out=decoded), reps)
        entry["without_jump_points"] = {"encode_ms": round(pe, 4), "decode_ms": round(pd, 4), "decode_kernel": B.last_kernel(),
                                        "decode_speedup_of_the_default": round(pd / dec_ms, 3)}
        del plain8
        dec_fn(enc, model, n_per, out=decoded)
    if check:
        plain = enc_fn(symbols, model, cfg, jump_points=0)
        used = torch.arange(plain.words.shape[1], device=symbols.device)[None, :] < plain.n_words[:, None]
        entry["bit_exact"] = bool(torch.equal(decoded, narrow)) and bool(torch.equal(enc.n_words, plain.n_words)) and \
            bool(((enc.words == plain.words) | ~used).all()) and int(enc.status.abs().sum().item()) == 0
        entry["bit_exact_scope"] = "words and counts of every stream vs the int32 call's (compared with the CPU oracle in the headline check), decoded symbols vs input"
    if native:
        os.environ["CST_NO_N8"] = "1"
        _native.reload_knobs()           # (the library reads its debug switches once, when it is loaded)
        try:
            c_enc = event_ms(lambda: enc_fn(narrow, model, cfg, out=enc), reps)
            ck = B.last_kernel()
            c_dec = event_ms(lambda: dec_fn(enc, model, n_per, out=decoded), reps)
            entry["conversion_path"] = {"encode_ms": round(c_enc, 4), "decode_ms": round(c_dec, 4), "coder_kernels": [ck, B.last_kernel()],
                                        "Msymbols_per_s": round(n_streams * n_per / (c_enc + c_dec) / 1e3, 1)}
        finally:
            del os.environ["CST_NO_N8"]
            _native.reload_knobs()
    return entry


def per_symbol_config(B, reps, check, n_streams=N_STREAMS, n_per=N_PER, lo=-100, hi=100):
    """f1: every symbol its own f64 (mean, std) -- coder.encode_reverse(symbols, QuantizedGaussian(lo, hi),
    means, stds) / coder.decode(family, means, stds), src/pybindings/stream/stack.rs:567-588, 733-751 -- for all streams at
    once, preset (32, 64, 24).  Algorithmic bytes per symbol and direction: 4 (symbol) + 16 (two f64 parameters) + 4 per
    compressed word.  Parameters from the streams' splitmix64 generators (draws n_per + 3 ...), symbols = the model's
    rounded normal deviate; every stream's words are compared with the CPU oracle's."""
    from oracle import oracle as O
    dev = "cuda"
    mu = torch.empty((n_streams, n_per), dtype=torch.float64, device=dev)
    sd = torch.empty_like(mu)
    sym = torch.empty((n_streams, n_per), dtype=torch.int32, device=dev)
    for a in range(0, n_streams, 4096):
        b = min(a + 4096, n_streams)
        u = unit_draws(splitmix_draws(SEED, a, b - a, n_per + 3, 3 * n_per, dev)).view(b - a, n_per, 3)
        mu[a:b] = -30.0 + 60.0 * u[:, :, 0]
        sd[a:b] = torch.exp(float(np.log(0.5)) + u[:, :, 1] * float(np.log(32.0)))
        z = torch.special.ndtri(u[:, :, 2].clamp(1e-12, 1 - 1e-12))
        sym[a:b] = torch.clamp(torch.round(mu[a:b] + sd[a:b] * z), lo, hi).to(torch.int32)
        del u, z
    cfg = (32, 64, 24)
    enc = B.ans_encode_gaussian(sym, lo, hi, mu, sd, cfg)
    decoded = torch.empty_like(sym)
    enc_ms = event_ms(lambda: B.ans_encode_gaussian(sym, lo, hi, mu, sd, cfg, out=enc), reps)
    dec_ms = event_ms(lambda: B.ans_decode_gaussian(enc, lo, hi, mu, sd, out=decoded), reps)
    total_words = enc.total_words()
    n_sym = n_streams * n_per
    byts = 20 * n_sym + 4 * total_words
    entry = {"workload": f"f1: every symbol its own f64 (mean, std), support {lo}..{hi} (the reference's flagship Python call, batched)",
             "coder": "ans", "config": list(cfg), "streams": n_streams, "symbols_per_stream": n_per,
             "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "Msymbols_per_s": round(n_sym / (enc_ms + dec_ms) / 1e3, 1),
             "words_per_stream": round(total_words / n_streams, 2), "algorithmic_bytes_per_symbol": round(byts / n_sym, 3),
             "encode_frac": round(byts / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
             "decode_frac": round(byts / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    entry["jump_points"] = int(enc.jump.pos.shape[1]) if enc.jump is not None else 0
    entry["encode_kernel"], entry["decode_kernel"] = "", ""
    B.ans_encode_gaussian(sym, lo, hi, mu, sd, cfg, out=enc); entry["encode_kernel"] = B.last_kernel()
    B.ans_decode_gaussian(enc, lo, hi, mu, sd, out=decoded); entry["decode_kernel"] = B.last_kernel()
    # the calls above are the DEFAULT calls: the fused encoder notes two jump points per stream on its way and the small-geometry lane
    # decoder runs two waves per SIMD on the 131 072 (stream, chunk) pairs.  For the record, the same batch without (jump_points=0:
    # the default before round 6); the words are the same
    jump_ok = True
    if enc.jump is not None:
        try:
            plain = B.ans_encode_gaussian(sym, lo, hi, mu, sd, cfg, jump_points=0)
            dec2 = torch.zeros_like(sym)
            pe = event_ms(lambda: B.ans_encode_gaussian(sym, lo, hi, mu, sd, cfg, out=plain, jump_points=0), reps)
            pd = event_ms(lambda: B.ans_decode_gaussian(plain, lo, hi, mu, sd, out=dec2), reps)
            entry["without_jump_points"] = {"encode_ms": round(pe, 4), "decode_ms": round(pd, 4), "decode_kernel": B.last_kernel(),
                                            "decode_speedup_of_the_default": round(pd / dec_ms, 3),
                                            "decode_frac": round(byts / (pd * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            if check:
                used = torch.arange(min(512, int(enc.words.shape[1])), device=dev)[None, :] < enc.n_words[:, None]
                w = used.shape[1]
                jump_ok = bool(torch.equal(dec2, sym)) and bool(torch.equal(plain.n_words, enc.n_words)) and \
                    bool(((plain.words[:, :w] == enc.words[:, :w]) | ~used).all())
            del plain, dec2
        except Exception as exc:      # noqa: BLE001
            entry["without_jump_points"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            jump_ok = False
    if check:
        ok = bool(torch.equal(decoded, sym)) and int(enc.status.abs().sum().item()) == 0 and jump_ok
        if ok:
            words, n_words, _ = enc.to_numpy()
            h_sym, h_mu, h_sd = sym.cpu().numpy(), mu.cpu().numpy(), sd.cpu().numpy()

            def work(ab):
                for s in range(*ab):
                    c = O.AnsCoder()
                    c.encode_gaussian_reverse(h_sym[s], lo, hi, h_mu[s], h_sd[s], 24, 32)
                    w = c.get_compressed()
                    if len(w) != n_words[s] or not np.array_equal(w, words[s, : n_words[s]]):
                        return False
                return True
            cores = host_cores()
            with ThreadPoolExecutor(max_workers=cores) as pool:
                ok = all(pool.map(work, _blocks(n_streams, 8 * cores)))
            entry["bit_exact_scope"] = f"all {n_streams} streams: words and counts vs CPU oracle, decoded symbols vs input"
        entry["bit_exact"] = ok
    # the same batch through the RANGE coder's per-symbol calls (round 6: they carry jump points by default too)
    try:
        renc = B.range_encode_gaussian(sym, lo, hi, mu, sd, cfg)
        rdec = torch.empty_like(sym)
        re_ms = event_ms(lambda: B.range_encode_gaussian(sym, lo, hi, mu, sd, cfg, out=renc), reps)
        rek = B.last_kernel()
        rd_ms = event_ms(lambda: B.range_decode_gaussian(renc, lo, hi, mu, sd, out=rdec), reps)
        r = {"encode_ms": round(re_ms, 4), "decode_ms": round(rd_ms, 4), "encode_kernel": rek, "decode_kernel": B.last_kernel(),
             "jump_points": int(renc.jump.pos.shape[1]) if renc.jump is not None else 0}
        if check:
            r["bit_exact"] = bool(torch.equal(rdec, sym)) and int(renc.status.abs().sum().item()) == 0
        if renc.jump is not None:
            rplain = B.range_encode_gaussian(sym, lo, hi, mu, sd, cfg, jump_points=0)
            rpd = event_ms(lambda: B.range_decode_gaussian(rplain, lo, hi, mu, sd, out=rdec), reps)
            used = torch.arange(min(512, int(renc.words.shape[1])), device=dev)[None, :] < renc.n_words[:, None]
            w = used.shape[1]
            r["without_jump_points"] = {"decode_ms": round(rpd, 4), "decode_speedup_of_the_default": round(rpd / rd_ms, 3)}
            if check:
                r["bit_exact"] = r["bit_exact"] and bool(torch.equal(rdec, sym)) and bool(torch.equal(rplain.n_words, renc.n_words)) and \
                    bool(((rplain.words[:, :w] == renc.words[:, :w]) | ~used).all())
            del rplain
        if check and r["bit_exact"]:
            h_sym, h_mu, h_sd = sym[:48].cpu().numpy(), mu[:48].cpu().numpy(), sd[:48].cpu().numpy()
            for s_ in range(0, 48, 4):      # a dozen streams against the CPU oracle (the ANS entry above compares all of them)
                c = O.RangeEncoder()
                c.encode(h_sym[s_], [O.GaussianModel(lo, hi, float(m), float(v), 24, 32) for m, v in zip(h_mu[s_], h_sd[s_])], 24)
                if renc.stream(s_).tolist() != c.get_compressed().tolist():
                    r["bit_exact"] = False
            r["bit_exact_scope"] = "decoded symbols vs input (all streams), words vs the plain call's, twelve streams' words vs the CPU oracle"
        entry["range_coder"] = r
        if check:
            entry["bit_exact"] = bool(entry.get("bit_exact", True)) and r["bit_exact"]
        del renc, rdec
    except Exception as exc:      # noqa: BLE001
        entry["range_coder"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        if check:
            entry["bit_exact"] = False
    B.release_scratch()
    return entry


def ragged_config(B, reps, check, n_docs=100_000, n_sym=64, precision=24):
    """Many small coders in one launch (the reference's second usage pattern, tests/issue52.rs:27-80: one DefaultAnsCoder per
    document of a compressed index, a shared model): 100 000 documents of 20 .. 2000 symbols (log-uniform lengths, seeded),
    a 64-symbol categorical model at P = 24, `cst_ans_{encode,decode}_ragged`.  Algorithmic bytes per direction: 4 per symbol +
    4 per compressed word + 16 per document (its two offsets).  These batches are bound by the latency of the longest
    document's chain, not by traffic: the fractions are reported for completeness.  Check: every 50th document's words
    against the CPU oracle coding that document alone; all decoded symbols against the input."""
    from oracle import oracle as O
    rng = np.random.default_rng(SEED)
    w = 0.93 ** np.arange(n_sym)
    prob = np.maximum(1, np.floor(w / w.sum() * ((1 << precision) - n_sym)).astype(np.int64))
    prob[0] += (1 << precision) - int(prob.sum())
    cdf = np.concatenate([[0], np.cumsum(prob)]).astype(np.uint32)
    model = B.Model.from_cdf(cdf, 0, precision)
    lengths = np.exp(rng.uniform(np.log(20), np.log(2000), n_docs)).astype(np.int64)
    offsets = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    n_total = int(offsets[-1])
    gen = torch.Generator(device="cuda").manual_seed(SEED)
    q = torch.randint(0, 1 << precision, (n_total,), generator=gen, device="cuda", dtype=torch.int64)
    flat = (torch.searchsorted(torch.from_numpy(cdf.astype(np.int64)).cuda(), q, right=True) - 1).to(torch.int32)
    del q
    off_d = torch.from_numpy(offsets).cuda()
    cfg = (32, 64, precision)
    enc = B.ans_encode_ragged(flat, off_d, model, cfg)
    decoded, status = B.ans_decode_ragged(enc, model, off_d)
    enc_ms = event_ms(lambda: B.ans_encode_ragged(flat, off_d, model, cfg), reps)
    dec_ms = event_ms(lambda: B.ans_decode_ragged(enc, model, off_d, out=decoded), reps)
    dec_kernel = B.last_kernel()
    total_words = int(enc.n_words.sum().item())
    byts = 4 * n_total + 4 * total_words + 16 * n_docs
    # the calls above are the DEFAULT calls: the encoder notes a jump point every batched.RAGGED_JUMP_EVERY symbols of every document on
    # its way and the decoder runs the chunks side by side (round 6) -- a launch lasts as long as its longest CHAIN.  For the record,
    # the same batch without (jump_every=0: the default before round 6; same words):
    plain = B.ans_encode_ragged(flat, off_d, model, cfg, jump_every=0)
    plain_dec = torch.empty_like(flat)
    pe = event_ms(lambda: B.ans_encode_ragged(flat, off_d, model, cfg, jump_every=0), reps)
    pd = event_ms(lambda: B.ans_decode_ragged(plain, model, off_d, out=plain_dec), reps)
    pos = torch.arange(enc.words.numel(), device=flat.device)
    owner = torch.searchsorted(enc.word_offsets[1:].contiguous(), pos, right=True).clamp_(max=n_docs - 1)
    used = pos - enc.word_offsets[owner] < enc.n_words[owner]          # (the slabs are only partly used)
    plain_ok = bool(torch.equal(plain_dec, flat)) and bool(torch.equal(plain.n_words, enc.n_words)) and \
        bool(((plain.words == enc.words) | ~used).all())
    del pos, owner, used
    entry = {"workload": f"many small coders (tests/issue52.rs pattern): {n_docs} documents of 20..2000 symbols in one launch, {n_sym}-symbol categorical model",
             "coder": "ans", "config": list(cfg), "streams": n_docs, "symbols_total": n_total,
             "jump_every": enc.jump.interval if enc.jump is not None else 0, "decode_kernel": dec_kernel,
             "jump_table_bytes_per_symbol": round(12 * int(enc.jump.chunk_offsets[-1].item()) / n_total, 4) if enc.jump is not None else 0,
             "without_jump_points": {"encode_ms": round(pe, 4), "decode_ms": round(pd, 4), "decode_speedup_of_the_default": round(pd / dec_ms, 3),
                                     "same_words_and_symbols": plain_ok},
             "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "Msymbols_per_s": round(n_total / (enc_ms + dec_ms) / 1e3, 1),
             "ns_per_document": [round(enc_ms * 1e6 / n_docs, 2), round(dec_ms * 1e6 / n_docs, 2)],
             "words_per_stream": round(total_words / n_docs, 2),
             "encode_frac": round(byts / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
             "decode_frac": round(byts / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    if check:
        ok = plain_ok and bool(torch.equal(decoded, flat)) and int(enc.status.abs().sum().item()) == 0 and int(status.abs().sum().item()) == 0
        if ok:
            h_flat, n_words = flat.cpu().numpy(), enc.n_words.cpu().numpy()
            h_words, h_woff = enc.words.cpu().numpy().view(np.uint32), enc.word_offsets.cpu().numpy()
            for d in range(0, n_docs, 50):
                doc = h_flat[offsets[d]: offsets[d + 1]]
                want, n, st = O.ans_encode_batch(doc[None, :], 0, cdf, precision, 32, 64)
                if st[0] != 0 or n[0] != n_words[d] or not np.array_equal(want[0, : n[0]], h_words[h_woff[d]: h_woff[d] + n[0]]):
                    ok = False
                    break
            entry["bit_exact_scope"] = f"every 50th of {n_docs} documents: words and counts vs CPU oracle; all decoded symbols vs input"
        entry["bit_exact"] = ok
    return entry


def end_to_end(B, model, symbols, n_chunks=8, n_hip_streams=3, reps=3, dtype=torch.int32):
    """the same batch from HOST memory and back -- what the reference's bindings do around every call (they copy
    the numpy input in, src/pybindings/mod.rs:240-243, and the compressed words out, pybindings/stream/stack.rs:422-427).
        encode leg:  pinned host symbols -> H2D -> cst_ans_encode_batch -> cst_compact_words -> D2H of the packed words
        decode leg:  pinned host words + offsets -> H2D -> cst_ans_decode_batch -> D2H of the symbols
    in `n_chunks` chunks of streams over `n_hip_streams` HIP streams, so that the copies of one chunk overlap the kernels of
    another.  Never part of `value`: the link, not the coder, sets these numbers.  dtype torch.int8 / int16: the symbol matrices
    cross the link in the narrow type and are widened / narrowed on the device (cst_ans_*_batch_sym)."""
    n_streams, n_per = symbols.shape
    per = n_streams // n_chunks
    cfg = (W, S, P)
    sym_bytes = torch.empty((), dtype=dtype).element_size()
    host_sym = torch.empty((n_streams, n_per), dtype=dtype, pin_memory=True)
    host_sym.copy_(symbols.to(dtype))
    stride = B.max_words(n_per, cfg)
    host_words = torch.empty(n_streams * stride, dtype=torch.int32, pin_memory=True)
    host_off = torch.empty((n_chunks, per + 1), dtype=torch.int64, pin_memory=True)
    host_back = torch.empty((n_streams, n_per), dtype=dtype, pin_memory=True)
    streams = [torch.cuda.Stream() for _ in range(n_hip_streams)]
    dsym = [torch.empty((per, n_per), dtype=dtype, device="cuda") for _ in range(n_hip_streams)]
    encs = [B.ans_encode(dsym[k], model, cfg) for k in range(n_hip_streams)]
    packs = [B.compact(encs[k]) for k in range(n_hip_streams)]
    dwords = [torch.empty(per * stride, dtype=torch.int32, device="cuda") for _ in range(n_hip_streams)]
    doff = [torch.empty(per + 1, dtype=torch.int64, device="cuda") for _ in range(n_hip_streams)]
    dnw = [torch.empty(per, dtype=torch.int32, device="cuda") for _ in range(n_hip_streams)]
    torch.cuda.synchronize()
    totals = [0] * n_chunks

    def encode_leg():
        for c in range(n_chunks):
            k = c % n_hip_streams
            with torch.cuda.stream(streams[k]):
                dsym[k].copy_(host_sym[c * per:(c + 1) * per], non_blocking=True)
                B.ans_encode(dsym[k], model, cfg, out=encs[k])
                B.compact(encs[k], out=packs[k])
                host_off[c].copy_(packs[k][1], non_blocking=True)
                streams[k].synchronize()                       # (this stream only: the chunk's word count decides the size of its copy)
                totals[c] = int(host_off[c, -1])
                base = c * per * stride
                host_words[base: base + totals[c]].copy_(packs[k][0][: totals[c]], non_blocking=True)
        torch.cuda.synchronize()

    def decode_leg():
        for c in range(n_chunks):
            k = c % n_hip_streams
            with torch.cuda.stream(streams[k]):
                base = c * per * stride
                dwords[k][: totals[c]].copy_(host_words[base: base + totals[c]], non_blocking=True)
                doff[k].copy_(host_off[c], non_blocking=True)
                dnw[k].copy_(doff[k][1:] - doff[k][:-1])
                B.ans_decode((dwords[k], dnw[k]), model, n_per, offsets=doff[k], config=cfg, out=dsym[k])   # (packed words without provenance -- they have just come over PCIe: the library reads them as cold)
                host_back[c * per:(c + 1) * per].copy_(dsym[k], non_blocking=True)
        torch.cuda.synchronize()

    def wall(fn):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best
    encode_leg()
    te = wall(encode_leg)
    td = wall(decode_leg)
    ok = bool(torch.equal(host_back, host_sym))
    n_sym = n_streams * n_per
    words_bytes = 4 * sum(totals)
    return {"what": f"pinned host memory -> device -> host, {n_chunks} chunks of {per} streams over {n_hip_streams} HIP streams; best of {reps}",
            "symbol_dtype": str(dtype).replace("torch.", ""),
            "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
            "Msymbols_per_s": round(n_sym / (te + td) / 1e6, 1),
            "encode_link_GBps": round((sym_bytes * n_sym + words_bytes) / te / 1e9, 1),
            "decode_link_GBps": round((sym_bytes * n_sym + words_bytes) / td / 1e9, 1),
            "bit_exact": ok}


def c1_config(B, check):
    """BASELINE config C1: ONE stream of 10^6 symbols, QuantizedGaussian(-50, 50, 3.2, 9.6) at P = 24 -- through the drop-in
    single coder (constriction_amd.stream.stack.AnsCoder: host arrays in, host arrays out, like the reference's Python API) and
    through the checkpointed batched calls (a jump point every 800 symbols -- whole 32-symbol tiles, so that the producer / consumer
    encoder notes them on its way: the one-lane-per-stream kernel that takes any interval needs 255 ms for this stream -- and the
    decoder runs on 1250 lanes)."""
    from oracle import oracle as O
    from constriction_amd.stream import stack, model as M
    lo, hi, mean, std, prec, n = -50, 50, 3.2, 9.6, 24, 1_000_000
    cdf = O.GaussianModel(lo, hi, mean, std, prec, 32).cdf_table()
    sym = O.synth_symbols(0xC0FFEE, 0, 1, n, lo, cdf, prec)[0]
    entry = {"workload": "C1: one stream of 1 000 000 symbols, QuantizedGaussian(-50,50,3.2,9.6), DefaultAnsCoder (32,64,24)", "coder": "ans",
             "config": [32, 64, prec], "streams": 1, "symbols_per_stream": n}
    fam = M.QuantizedGaussian(lo, hi, mean, std)
    coder = stack.AnsCoder()
    coder.encode_reverse(sym[:1000], fam)                                          # (first call: library and tables warm)
    coder = stack.AnsCoder()
    t0 = time.perf_counter()
    coder.encode_reverse(sym, fam)
    words = coder.get_compressed()
    t1 = time.perf_counter()
    out = stack.AnsCoder(words).decode(fam, n)
    t2 = time.perf_counter()
    entry["dropin_encode_ns_per_symbol"] = round((t1 - t0) / n * 1e9, 1)
    entry["dropin_decode_ns_per_symbol"] = round((t2 - t1) / n * 1e9, 1)
    entry.update(rate_report(len(words), n, 32, model_entropy_bits(cdf, prec)))
    model = B.Model.quantized_gaussian(lo, hi, mean, std, prec)
    dsym = torch.from_numpy(sym[None, :].copy()).cuda()
    enc, ck = B.ans_encode_checkpointed(dsym, model, 800, (32, 64, prec))
    entry["checkpoint_interval"], entry["checkpointed_encode_kernel"] = 800, B.last_kernel()
    dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n)
    entry["checkpointed_encode_ms"] = round(event_ms(lambda: B.ans_encode_checkpointed(dsym, model, 800, (32, 64, prec)), 3), 3)
    entry["checkpointed_decode_ms"] = round(event_ms(lambda: B.ans_decode_checkpointed(enc, ck, model, n), 3), 3)
    ok = bool(np.array_equal(out, sym)) and bool(torch.equal(dec[0].cpu(), torch.from_numpy(sym))) and int(dstatus.abs().sum().item()) == 0
    if check and ok:
        want_words, want_n, _ = O.ans_encode_batch(sym[None, :], lo, cdf, prec)
        ok = words.tolist() == want_words[0, : want_n[0]].tolist() and enc.stream(0).tolist() == words.tolist()
        entry["bit_exact_scope"] = "drop-in and checkpointed words vs CPU oracle, decoded symbols vs input"
    entry["bit_exact"] = ok
    return entry


def other_configs(B, rank, world, dist, args, reps=5):
    """Every other single-GPU configuration of BASELINE.json, same clock, same checks (see the module docstring)."""
    out = []
    check = not args.no_check
    from oracle import oracle as O

    def gaussian(precision, lo=LO, hi=HI):
        m = B.Model.quantized_gaussian(lo, hi, MEAN, STD, precision)
        cdf = m.cdf()
        if check:
            assert cdf.tolist() == O.GaussianModel(lo, hi, MEAN, STD, precision, 32).cdf_table().tolist(), "device table != oracle"
        return m, cdf

    def add(name, *a, **k):
        """one entry; whatever goes wrong inside it is reported in its place (and fails the run's check) instead of taking the
        other entries with it"""
        try:
            entry = run_config(B, name, *a, stride="tuned" if args.slab_stride == "tuned" else None, **k)[0]
        except Exception as exc:      # noqa: BLE001
            entry = {"workload": name, "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False}
        out.append(entry)

    def jump(name, *a, **k):
        try:
            entry = jump_config(B, name, *a, **k)
        except Exception as exc:      # noqa: BLE001
            entry = {"workload": name, "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False}
        out.append(entry)

    m12, cdf12 = gaussian(12)
    cdf12_dev = torch.from_numpy(cdf12.astype(np.int64)).cuda()
    sym12 = synth_symbols_device(SEED, rank * N_STREAMS, N_STREAMS, N_PER, LO, cdf12_dev, 12)
    if world == 1:
        try:
            out.append(c1_config(B, check))
        except Exception as exc:      # noqa: BLE001
            out.append({"workload": "C1: one stream of 1 000 000 symbols", "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False})
        symT = sym12.t().contiguous()
        add("C2 as symbols[t][stream] (symbol-major layout)", "ans", (32, 64, 12), m12, symT, reps, check, cdf12,
                          layout="symbol_major")
        del symT
        add("C2 with 16-bit words (SmallAnsCoder preset)", "ans", (16, 32, 12), m12, sym12, reps, check, cdf12)
        add("C2 with 16-bit words, PACKED two per slot as the reference's Vec<u16> (CST_FLAG_PACKED_W16)", "ans", (16, 32, 12), m12, sym12, reps, check,
            cdf12, packed16=True)
        for narrow_dtype in (torch.int8, torch.int16):
            try:
                out.append(narrow_config(B, m12, sym12, reps, check, dtype=narrow_dtype))
            except Exception as exc:      # noqa: BLE001
                out.append({"workload": f"C2 with {str(narrow_dtype).replace('torch.', '')} symbol matrices", "error": f"{type(exc).__name__}: {exc}"[:200],
                            "bit_exact": False})
        m24, cdf24 = gaussian(24)
        sym24 = synth_symbols_device(SEED, rank * N_STREAMS, N_STREAMS, N_PER, LO, torch.from_numpy(cdf24.astype(np.int64)).cuda(), 24)
        add("C2 at P = 24 (DefaultAnsCoder preset)", "ans", (32, 64, 24), m24, sym24, reps, check, cdf24)
        try:
            out.append(narrow_config(B, m24, sym24, reps, check, dtype=torch.int8, cfg=(32, 64, 24), name="C2 at P = 24 (DefaultAnsCoder preset)"))
        except Exception as exc:      # noqa: BLE001
            out.append({"workload": "C2 at P = 24 with int8 symbol matrices", "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False})
        add("C4 range coder, P = 12", "range", (32, 64, 12), m12, sym12, reps, check, cdf12)
        add("C4 range coder, P = 24", "range", (32, 64, 24), m24, sym24, reps, check, cdf24)
        for mN, symN, PN in ((m12, sym12, 12), (m24, sym24, 24)):      # round 6: the range coder's loops read / write int8 themselves
            try:
                out.append(narrow_config(B, mN, symN, reps, check, dtype=torch.int8, cfg=(32, 64, PN), name=f"C4 range coder, P = {PN},", coder="range"))
            except Exception as exc:      # noqa: BLE001
                out.append({"workload": f"C4 range coder, P = {PN}, with int8 symbol matrices", "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False})
        jump("C2 decode with k jump points per stream (small-footprint decoder on 65 536 k virtual streams; the producer / consumer encoder "
             "notes the jump points on its way)", "ans", (32, 64, 12), m12, sym12, reps, check, cdf12)
        jump("C2 with int8 symbol matrices: decode with k jump points per stream (the loops read / write int8 themselves; 65 536 k virtual "
             "streams on the small-footprint int8 decoder, two waves per SIMD)", "ans", (32, 64, 12), m12, sym12.to(torch.int8), reps, check, cdf12)
        jump("C2 at P = 24 (DefaultAnsCoder preset): decode with k jump points per stream (small-footprint bucket-entry decoder, two waves per SIMD)",
             "ans", (32, 64, 24), m24, sym24, reps, check, cdf24)
        jump("C2 at P = 24 with int8 symbol matrices: decode with k jump points per stream (small-footprint bucket-entry decoder that writes int8)",
             "ans", (32, 64, 24), m24, sym24.to(torch.int8), reps, check, cdf24)
        jump("C4 range coder, P = 12: decode with k jump points per stream (sub-lane decoder)", "range", (32, 64, 12), m12, sym12, reps, check, cdf12)
        jump("C4 range coder, P = 24: decode with k jump points per stream (sub-lane decoder)", "range", (32, 64, 24), m24, sym24, reps, check, cdf24)
        del sym24, m24
        symT = sym12.t().contiguous()
        add("C4 range coder, P = 12, symbols[t][stream] (symbol-major layout)", "range", (32, 64, 12), m12, symT, reps, check, cdf12,
                          layout="symbol_major")
        del symT
        # an alphabet of 700 symbols at P = 16 and rows of 4100 symbols: no 2^P-entry lookup table, more symbols than an 8-bit bucket
        # index addresses, rows that are not cache-line aligned (the shapes next to the headline one: scripts/bench_variants.py)
        big = B.Model.quantized_gaussian(-350, 349, 3.2, 96.0, 16)
        cdf_big = big.cdf()
        sym_big = synth_symbols_device(SEED, rank * N_STREAMS, N_STREAMS, 4100, -350, torch.from_numpy(cdf_big.astype(np.int64)).cuda(), 16)
        add("ANS, 700 symbols at P = 16, rows of 4100 symbols (10-bit bucket index, second-level tables, row skew)", "ans", (32, 64, 16), big, sym_big,
                          reps, check, cdf_big, lo=-350)
        del sym_big, big
        # C3: one (mean, std) per stream, support -127..127
        # (parameters and symbols from the per-stream splitmix64 generators, like C2's)
        mu_d, sigma_d = c3_parameters(SEED, rank * N_STREAMS, N_STREAMS, N_PER, "cuda")
        mu, sigma = mu_d.cpu().numpy(), sigma_d.cpu().numpy()
        m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
        sym3 = synth_symbols_per_stream(SEED, rank * N_STREAMS, N_PER, -127, m3.cdfs_device(), 12)
        cdfs = cpu_tables(-127, 127, mu, sigma, 12) if check else None
        add("C3 per-stream (mean, std) tables, support -127..127", "ans", (32, 64, 12), m3, sym3, reps, check, cdfs, lo=-127)
        try:      # round 6: C3's support fits int8 -- the compact-row encoder and the sub-lane decoder read / write the int8 matrix themselves
            out.append(narrow_config(B, m3, sym3, reps, check, dtype=torch.int8, cfg=(32, 64, 12), name="C3 per-stream (mean, std) tables"))
        except Exception as exc:      # noqa: BLE001
            out.append({"workload": "C3 with int8 symbol matrices", "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False})
        jump("C3: decode with k jump points per stream (sub-lane decoder: the lanes of a stream share its table in LDS)", "ans", (32, 64, 12),
             m3, sym3, reps, check, cdfs, lo=-127, ks=(2, 4, 8))
        del sym3, m3, cdfs
        out.append(per_symbol_config(B, reps, check))
        try:
            out.append(ragged_config(B, reps, check))
        except Exception as exc:      # noqa: BLE001
            out.append({"workload": "many small coders (ragged)", "error": f"{type(exc).__name__}: {exc}"[:200], "bit_exact": False})
    del sym12
    torch.cuda.empty_cache()
    # C5 shard: 131 072 streams per GPU, compaction, gather of the packed words to rank 0
    n5 = args.c5_streams
    sym5 = synth_symbols_device(SEED, rank * n5, n5, N_PER, LO, cdf12_dev, 12)
    e, enc5 = run_config(B, f"C5 shard: {n5} streams/GPU x {world} GPU(s)", "ans", (32, 64, 12), m12, sym5, reps, check and world == 1, cdf12,
                         stride="tuned" if args.slab_stride == "tuned" else None)
    packed, offsets = B.compact(enc5)
    e["compact_ms"] = round(event_ms(lambda: B.compact(enc5, out=(packed, offsets)), reps), 4)
    if world == 1:
        for key, narrow_dtype in (("int8_symbols", torch.int8), ("int16_symbols", torch.int16)):
            try:      # the shard as a narrow matrix: a quarter / half of the symbol bytes of a batch that never fits the caches
                e8 = narrow_config(B, m12, sym5, reps, check, dtype=narrow_dtype)
                e[key] = {k: e8[k] for k in ("encode_ms", "decode_ms", "Msymbols_per_s", "encode_kernel", "decode_kernel", "bit_exact",
                                             "conversion_path") if k in e8}
            except Exception as exc:      # noqa: BLE001
                e[key] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    if dist is not None and not args.no_gather:
        from constriction_amd import dist as D
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = D.gather_packed(packed[: int(offsets[-1].item())], offsets, dst=0)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        e["gather_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        if rank == 0:
            e["gathered_words"] = int(res[0].numel())
            e["gather_GBps"] = round(4 * res[0].numel() * (world - 1) / world / (e["gather_ms"] * 1e-3) / 1e9, 1)
        # the same exchange through the C ABI's own RCCL communicator (what a non-Python caller uses); never fatal
        try:
            comm = D.RcclComm()
            comm.gather_packed(packed, offsets, dst=0)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            res2 = comm.gather_packed(packed, offsets, dst=0)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            e["gather_c_abi_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            if rank == 0:
                e["gather_c_abi_matches"] = bool(torch.equal(res2[0], res[0]) and torch.equal(res2[1], res[1]))
            comm.close()
        except Exception as exc:      # noqa: BLE001
            e["gather_c_abi_error"] = f"{type(exc).__name__}: {exc}"[:200]
    out.append(e)
    return out


def headline_side_legs(B, args, world, model, symbols, enc, decoded):
    """what the line reports beside the timed step, on the same batch: compaction, the host -> device -> host leg, the same two
    kernels after cache flushes.  Returns (compact_ms, end_to_end or None, after_cache_flush)."""
    packed, offsets = B.compact(enc)
    compact_ms = event_ms(lambda: B.compact(enc, out=(packed, offsets)), 5)
    e2e = None
    if world == 1 and not args.no_end_to_end:
        try:
            e2e = end_to_end(B, model, symbols)
        except Exception as exc:      # noqa: BLE001
            e2e = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        try:      # the same with int8 symbol matrices (the alphabet has 101 symbols): a quarter of the symbol bytes on the link
            e2e["int8_symbols"] = end_to_end(B, model, symbols, dtype=torch.int8)
        except Exception as exc:      # noqa: BLE001
            e2e["int8_symbols"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # The timed steps decode what the encoder has just written.  For the record, the same two kernels after a 1-GiB fill
    # (nothing of the batch left in L2 or the 256-MiB Infinity Cache; DESIGN.md 3.8 "working sets beyond the caches"):
    flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")

    def after_flush_ms(fn, reps=3, clean=False):
        total = 0.0
        for _ in range(reps):
            if clean:
                flush.view(torch.int32).sum()      # a 1-GiB READ: the caches end up full of clean lines
            else:
                flush.fill_(1)                     # a 1-GiB fill: ... of DIRTY lines, whose write-back the launch then shares HBM with
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            total += e0.elapsed_time(e1)
        return total / reps
    # words WITHOUT provenance: the same buffers as a caller holds them who got the words from the host, a peer or a file (the
    # library then decides for itself: batched._words_are_cold)
    foreign = B.EncodedBatch(enc.words, enc.n_words, enc.status, enc.config)
    cold = {"encode_ms": round(after_flush_ms(lambda: B.ans_encode(symbols, model, (W, S, P), out=enc)), 4),
            "decode_ms": round(after_flush_ms(lambda: B.ans_decode(foreign, model, N_PER, out=decoded)), 4),
            "decode_kernel": B.last_kernel(),
            "decode_chunk_loads_ms": round(after_flush_ms(lambda: B.ans_decode(enc, model, N_PER, out=decoded, cold=False)), 4),
            "decode_cold_words_hint_ms": round(after_flush_ms(lambda: B.ans_decode(enc, model, N_PER, out=decoded, cold=True)), 4),
            "what": "same batch, a 1-GiB fill before every launch (nothing of it left in L2 or the Infinity Cache).  decode_ms: the default "
                    "call on words whose provenance the library does not know (as they arrive from the host, a peer, a file): it takes the "
                    "lane-quad decoder (cst_ans_dq.hip) by itself; decode_chunk_loads_ms: the decoder of the timed steps forced (cold=False); "
                    "decode_cold_words_hint_ms: CST_FLAG_COLD_WORDS passed explicitly"}
    cold["hint_bit_exact"] = bool(torch.equal(decoded, symbols))
    cold["after_a_1GiB_read_instead"] = {
        "encode_ms": round(after_flush_ms(lambda: B.ans_encode(symbols, model, (W, S, P), out=enc), clean=True), 4),
        "decode_ms": round(after_flush_ms(lambda: B.ans_decode(foreign, model, N_PER, out=decoded), clean=True), 4),
        "decode_chunk_loads_ms": round(after_flush_ms(lambda: B.ans_decode(enc, model, N_PER, out=decoded, cold=False), clean=True), 4),
        "decode_cold_words_hint_ms": round(after_flush_ms(lambda: B.ans_decode(enc, model, N_PER, out=decoded, cold=True), clean=True), 4),
        "what": "the same with the caches full of clean lines (a fill leaves 288 MiB of dirty lines whose write-back competes with the launch)"}
    del flush

    return compact_ms, e2e, cold


# ---- the output contract: ONE small last line, everything else beside it ----
LINE_LIMIT = 4096          # bytes; round 5's 21-KB line was not parsed by the driver (BENCH_r05.json: parsed null)
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "bit_exact", "encode_ms", "decode_ms", "compact_ms", "roofline", "cpu_baseline",
                 "per_rank", "rccl_ranks_seen", "configs_error")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_cold", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample", "single_thread_value", "logical_cpus", "cgroup_cpu_quota")


def contract_line(full):
    """The LAST stdout line: the contract's keys only, prose cut short, <= LINE_LIMIT bytes whatever the run added to `full`.
    The reports (`configs`, `after_cache_flush`, `end_to_end`, `rate`, notes) travel in the detail record instead."""
    short = {k: full[k] for k in CONTRACT_KEYS if k in full}
    if "roofline" in short:
        short["roofline"] = {k: full["roofline"][k] for k in ROOFLINE_KEYS if k in full["roofline"]}
    if "cpu_baseline" in short:
        cb = {k: full["cpu_baseline"][k] for k in CPU_BASELINE_KEYS if k in full["cpu_baseline"]}
        if isinstance(cb.get("sample"), str):
            cb["sample"] = cb["sample"][:160]
        short["cpu_baseline"] = cb
    if "configs" in full:
        cfgs = full["configs"]
        short["configs_reported"] = len(cfgs)
        short["configs_bit_exact"] = all(c.get("bit_exact", True) for c in cfgs)
    short["detail"] = "bench_detail.json (also the previous stdout line)"
    # whatever else grows: the line must stay parseable
    for victim in ("per_rank", "configs_error", "detail"):
        if len(json.dumps(short)) < LINE_LIMIT:
            break
        short.pop(victim, None)
    if len(json.dumps(short)) >= LINE_LIMIT and isinstance(short.get("config"), dict):
        short["config"] = {k: (v[:120] if isinstance(v, str) else v) for k, v in short["config"].items()}
    assert len(json.dumps(short)) < LINE_LIMIT, "bench line outgrew its reader"
    return short


def emit(full, detail_path=None):
    """detail record -> bench_detail.json next to this script AND an earlier stdout line; then the contract line, LAST"""
    detail = json.dumps({"bench_detail": full})
    try:
        Path(detail_path or (ROOT / "bench_detail.json")).write_text(detail + "\n")
    except OSError as exc:
        print(json.dumps({"bench_detail_write_error": str(exc)[:200]}), flush=True)
    print(detail, flush=True)
    print(json.dumps(contract_line(full)), flush=True)


def plumbing(args, rank, world, dist):
    """The multi-rank skeleton of this script on host tensors (see --plumbing): every rank owns streams
    [rank * n, (rank + 1) * n) whose "compressed words" are a function of the global stream id, packs them, gathers them
    to rank 0 with the product's gather and reports like the real run."""
    from constriction_amd import dist as D
    n = args.streams
    sid = np.arange(rank * n, (rank + 1) * n, dtype=np.int64)
    lens = (7 * sid + 3) % 11                                          # some streams are empty
    words = np.concatenate([np.arange(l, dtype=np.int64) + 1000 * s for s, l in zip(sid, lens)] + [np.zeros(0, np.int64)])
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
    packed = torch.from_numpy(words.astype(np.int32))
    t0 = time.perf_counter()
    res = D.gather_packed(packed, offsets, dst=0) if dist is not None else (packed, offsets)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = True
    if rank == 0:
        all_sid = np.arange(0, world * n, dtype=np.int64)
        all_len = (7 * all_sid + 3) % 11
        want = np.concatenate([np.arange(l, dtype=np.int64) + 1000 * s for s, l in zip(all_sid, all_len)] + [np.zeros(0, np.int64)])
        ok = np.array_equal(res[0].numpy().astype(np.int64), want.astype(np.int32).astype(np.int64)) and \
            res[1].numpy().tolist() == np.concatenate([[0], np.cumsum(all_len)]).tolist()
        print(json.dumps({"plumbing": "ok" if ok else "FAILED", "n_gpus": world, "streams_per_rank": n, "gathered_words": int(res[0].numel()),
                          "gather_ms": round(elapsed * 1e3, 3), "backend": args.backend if dist is not None else None}), flush=True)
    else:
        ok = res is None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("plumbing check FAILED")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=N_STREAMS, help="streams per GPU of the headline step (default: BASELINE config C2)")
    ap.add_argument("--c5-streams", type=int, default=C5_STREAMS, help="streams per GPU of the C5 shard entry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host -> device -> host leg (pinned buffers, chunked over HIP streams)")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the `configs` block")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed step and nothing else on the GPU: no configs, no end-to-end leg, no cache-flush legs, no compaction, no CPU "
                         "baseline -- the run whose rocprofv3 kernel trace holds ONLY the headline pair's hot launches (scripts/profile_round.sh)")
    ap.add_argument("--graph", action="store_true", help="replay the timed step as a HIP graph instead of launching eagerly")
    ap.add_argument("--slab-stride", default="tuned",
                    help="words between the headline batch's slabs: 'tuned' (batched.tuned_stride measures it before the warmup), "
                         "'default' (max_words), or a number")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the RCCL gather of the C5 shard's packed words")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--detail-out", default=None, help="where the detail record goes (default: bench_detail.json next to this script)")
    ap.add_argument("--plumbing", action="store_true",
                    help="no GPU work: run only the multi-rank plumbing of this script (self-launch, rendezvous, stream sharding, "
                         "variable-length gather of per-stream words to rank 0, max-over-ranks timing) on host tensors -- "
                         "what tests/test_dist_cpu.py drives with --backend gloo in the CPU-only build container")
    args = ap.parse_args()
    if args.headline_only:
        args.no_configs = args.no_end_to_end = args.no_cpu_baseline = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start our own ranks: one process per GPU on this node, rendezvous on 127.0.0.1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not args.plumbing:
        if os.environ.get("CST_BENCH_SHARE_GPU"):        # dry runs of the N > 1 path on a one-GPU box (--backend gloo --no-gather)
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, **kw)
    if args.plumbing:
        return plumbing(args, rank, world, dist)

    from constriction_amd import batched as B

    n_streams = args.streams
    model = B.Model.quantized_gaussian(LO, HI, MEAN, STD, P)
    cdf = model.cdf()
    cdf_dev = torch.from_numpy(cdf.astype(np.int64)).cuda()
    symbols = synth_symbols_device(SEED, rank * n_streams, n_streams, N_PER, LO, cdf_dev, P)
    # allocates slabs / counts once; how far apart the slabs lie is the caller's choice at the C ABI (stride_words)
    slab_stride = {"tuned": "tuned", "default": None}.get(args.slab_stride) if not args.slab_stride.isdigit() else int(args.slab_stride)
    enc = B.ans_encode(symbols, model, (W, S, P), stride=slab_stride)
    decoded = torch.empty_like(symbols)
    torch.cuda.synchronize()

    # one step = cst_ans_encode_batch + cst_ans_decode_batch through the C ABI, arguments converted once (the Python
    # wrappers cost ~40 us per call: nothing next to a kernel, but a busy host core then shows up as gaps between them)
    eager_step = B.ans_roundtrip_launcher(symbols, model, enc, decoded)

    # One step = two kernel launches of ~0.3 ms.  Eager launches pipeline (the host is two launches ahead of the GPU);
    # replaying the step as a HIP graph was measured SLOWER (0.65 vs 0.59 ms per step: ~60 us of fixed cost per replay
    # with nothing to amortise it over), so it is opt-in.
    step, launch_mode = eager_step, "eager"
    if args.graph:
        try:
            eager_step()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eager_step()
            graph.replay()
            torch.cuda.synchronize()
            step, launch_mode = graph.replay, "hipGraph replay of (encode, decode)"
        except Exception as exc:      # noqa: BLE001
            launch_mode = f"eager (graph capture failed: {type(exc).__name__})"
            torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # the GPU leaves the host-side setup at idle clocks: ~30 ms of the same launches first (the stride tuner does that as a side
    # effect, `--slab-stride default` does not: 20 steps right behind 3 warmup steps measured the ramp, 0.61 instead of 0.56 ms)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.03:
        step()
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
    # which kernels the dispatcher took for this shape on this box (the C2 shape: the producer / consumer encoder, cst_ans_pc.hip)
    B.ans_encode(symbols, model, (W, S, P), out=enc)
    enc_kernel = B.last_kernel()
    B.ans_decode(enc, model, N_PER, out=decoded, cold=False)
    dec_kernel = B.last_kernel()
    dominant, dom_ms = (enc_kernel, enc_ms) if enc_ms >= dec_ms else (dec_kernel, dec_ms)
    achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9

    ok = True
    scope = "not checked"
    if not args.no_check:
        ok = bool(torch.equal(decoded, symbols)) and int(enc.status.abs().sum().item()) == 0
        scope = "decoded symbols vs input on every rank"
        if rank == 0 and ok:
            words, n_words, _ = enc.to_numpy()
            ok = cpu_words_match("ans", symbols.cpu().numpy(), words, n_words, LO, cdf, P, W, S)
            scope = f"all {n_streams} streams of rank 0: words and counts vs CPU oracle; decoded symbols vs input on every rank"
            del words

    compact_ms, cold = None, None
    e2e = None
    if not args.headline_only:
        compact_ms, e2e, cold = headline_side_legs(B, args, world, model, symbols, enc, decoded)

    # N > 1: every rank's own kernel times and how many ranks RCCL really connected (a scaling run diagnoses itself)
    per_rank, ranks_seen = None, None
    if dist is not None:
        mine = torch.tensor([float(rank), enc_ms, dec_ms, float(total_words)], dtype=torch.float64, device="cuda")
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rows = torch.stack(gathered).cpu().numpy()
        per_rank = [{"rank": int(r[0]), "encode_ms": round(float(r[1]), 4), "decode_ms": round(float(r[2]), 4), "words": int(r[3])} for r in rows]
        ranks_seen = len({int(r[0]) for r in rows})

    # the `configs` block is a report beside the bench line: whatever goes wrong in it (a collective of the gather leg on a
    # box this script has never seen, say) must not take the line itself -- measured above -- with it
    configs, configs_error = None, None
    if not args.no_configs:
        try:
            configs = other_configs(B, rank, world, dist, args)
        except Exception as exc:      # noqa: BLE001
            configs_error = f"{type(exc).__name__}: {exc}"[:300]

    if rank == 0:
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc cannot run inside this process): the committed summary
        # of the same command, named as the source
        traffic, traffic_source = None, None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get(dominant, {}).get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic.json (rocprofv3 --pmc passes of this command, scripts/pmc_all.sh)"
            except Exception:
                traffic = None
        dom_cold_ms = None if cold is None else (cold["encode_ms"] if dominant == enc_kernel else cold["decode_ms"])
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
            "bit_exact": ok, "bit_exact_scope": scope, "launch": launch_mode,
            "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "compact_ms": None if compact_ms is None else round(compact_ms, 4), "after_cache_flush": cold,
            "words_per_stream": round(total_words / n_streams, 2), "rate": rate_report(total_words, n_sym, W, model_entropy_bits(cdf, P)),
            "slab_stride_words": int(enc.words.shape[1]), "slab_stride_source": args.slab_stride,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "frac_cold": None if dom_cold_ms is None else round(bytes_per_launch / (dom_cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(dom_ms, 4),
                         "encode_GBps": round(bytes_per_launch / (enc_ms * 1e-3) / 1e9, 1),
                         "decode_GBps": round(bytes_per_launch / (dec_ms * 1e-3) / 1e9, 1)},
        }
        if e2e is not None:
            line["end_to_end"] = e2e
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["rccl_ranks_seen"] = ranks_seen
        if configs is not None:
            line["configs"] = configs
            ok = ok and all(c.get("bit_exact", True) for c in configs)
        if configs_error is not None:
            line["configs_error"] = configs_error
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cdf, symbols.cpu().numpy())
        emit(line, args.detail_out)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:             # noqa: BLE001  (a rank that lost its peers in the configs block still exits cleanly)
            pass
    if not ok:
        raise SystemExit("bit-exactness check FAILED")


if __name__ == "__main__":
    main()
