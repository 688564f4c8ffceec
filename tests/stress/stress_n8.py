#!/usr/bin/env python3
"""Randomised parity stress of the kernels that read / write INT8 symbol matrices themselves (round 5: ans_encode_pc_n8_kernel with and
without jump points, ans_decode_n8_kernel, ans_decode_small_n8_kernel) against the CPU oracle (not part of the test suite: minutes of
GPU time).  Only shapes those kernels take: any number of streams, rows of whole 128-byte lines (int8 and, since later in the round, int16), (32,64),
8 <= P <= 24 (12 < P: the wide coders and bucket-entry decoders, int32 matrices too), supports inside the type; random tables (model-distributed, uniform and all-tail data: up to P bits per symbol), slab strides (some too
small: CST_STREAM_CAPACITY), impossible symbols, jump points of every interval that divides the rows, and batches of more than 256
streams per CU (the two-waves-per-SIMD decoder).
usage: python tests/stress/stress_n8.py [seconds] [seed]"""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from constriction_amd import batched as B, _native as N
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_cases = n_streams_total = n_jump = n_small = 0
lib = N.lib()
cus = torch.cuda.get_device_properties(0).multi_processor_count
while time.time() < t_end:
    P = int(rng.integers(8, 13)) if rng.random() < 0.5 else int(rng.integers(13, 25))      # 12 < P <= 24: the wide step, bucket-entry decoders
    nb = int(rng.choice([1, 1, 2] + ([4] if P > 12 else [])))      # int8 or int16 matrices (12 < P: also the int32 form of the same coder waves)
    n = int(rng.choice([2, 3, 17, 101, 128, 255, 256] if nb == 1 else [2, 17, 101, 256, 300, 601, 1024]))
    n = min(n, (1 << P) // 2)
    lo = int(rng.integers(-128, 128 - n + 1)) if nb == 1 else int(rng.choice([-32768, 32768 - n, -n // 2, int(rng.integers(-32768, 32768 - n + 1))]))
    if nb == 4 and rng.random() < 0.5:
        lo = int(rng.choice([-2 ** 31, 2 ** 31 - n, 10 ** 6]))
    tmin, tmax, np_t, t_t, line = {1: (-128, 127, np.int8, torch.int8, 128), 2: (-32768, 32767, np.int16, torch.int16, 64),
                                   4: (-2 ** 31, 2 ** 31 - 1, np.int32, torch.int32, 32)}[nb]
    bits = {1: "_n8", 2: "_n16", 4: ""}[nb]
    if P <= 12:
        names = {"enc": f"ans_encode_pc{bits}_kernel", "ck": f"ans_encode_pc{bits}_kernel<ckpt>", "dec": f"ans_decode{bits}_kernel",
                 "small": f"ans_decode_small{bits}_kernel"}
    else:
        names = {"enc": f"ans_encode_pc{bits}_kernel<wide>", "ck": f"ans_encode_pc{bits}_kernel<wide, ckpt>", "dec": f"ans_decode_b16{bits}_kernel",
                 "small": f"ans_decode_b16_small{bits}_kernel"}
    w = rng.gamma(0.3, 1.0, n) + 1e-9
    p = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n)).astype(np.int64))
    p[int(np.argmax(p))] += (1 << P) - int(p.sum())
    cdf = np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)
    model = B.Model.from_cdf(cdf, lo, P)
    big = rng.random() < 0.15                           # more than 256 streams per CU: the small-footprint decoder
    n_streams = cus * 256 + 256 * int(rng.integers(1, 4)) if big else 256 * int(rng.choice([1, 2, 3, 8]))
    n_per = line * int(rng.choice([1, 2]) if big else rng.choice([1, 2, 3, 4, 5, 8, 32]))
    if nb == 4:
        n_per = 32 * int(rng.choice([2, 4]) if big else rng.choice([2, 3, 4, 5, 8, 16, 64]))     # (the int32 coder's read-ahead wants two tiles)
    if rng.random() < 0.3 and not big:
        n_streams = int(rng.integers(1, 700))           # partial workgroups / waves
    kind = rng.random()
    if kind < 0.4:
        idx = rng.choice(n, size=(n_streams, n_per), p=p / float(1 << P))
    elif kind < 0.7:
        idx = rng.integers(0, n, (n_streams, n_per))
    else:                                               # the rarest symbols only: the maximum rate of the word windows
        rare = np.flatnonzero(p == p.min())
        idx = rng.choice(rare, size=(n_streams, n_per))
    sym = (idx + lo).astype(np_t)
    bad_rows = []
    if not big:
        for _ in range(int(rng.choice([0, 0, 1, 5]))):      # impossible symbols (where the type has room for them)
            cand = [v for v in (lo - 1, lo + n, tmin, tmax) if tmin <= v <= tmax and not (lo <= v < lo + n)]
            if cand:
                r = int(rng.integers(n_streams))
                sym[r, rng.integers(n_per)] = int(rng.choice(cand))
                bad_rows.append(r)
    check = np.arange(n_streams) if not big else np.unique(np.concatenate([rng.integers(0, n_streams, 600), [0, n_streams - 1]]))
    want_words, want_n, want_st = O.ans_encode_batch(sym[check].astype(np.int32), lo, cdf, P)
    full = B.max_words(n_per, (32, 64, P))
    stride = full if big else int(rng.choice([full, full + 16, 16 * max(1, int(want_n.max()) // 16), 16, 48]))
    want_st = np.where((want_st == 0) & (want_n > stride), 2, want_st)
    guard = torch.full((n_streams * stride + 1024,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    d = torch.from_numpy(sym).cuda()
    assert d.data_ptr() % 128 == 0 and guard.data_ptr() % 64 == 0
    n_words = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    status = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    divisors = [k for k in (1, 2, 3, 4, 8) if n_per % k == 0 and (n_per // k) % 32 == 0]
    k = int(rng.choice(divisors)) if rng.random() < 0.5 else 0
    tag = f"bytes={nb} P={P} n={n} lo={lo} streams={n_streams} n_per={n_per} stride={stride} k={k} kind={kind:.2f}"
    if k:
        interval = n_per // k
        pos = torch.zeros((n_streams, k), dtype=torch.int32, device="cuda")
        state = torch.zeros((n_streams, k), dtype=torch.int64, device="cuda")
        N.check(lib.cst_ans_encode_batch_ckpt_sym(model._h, N.CoderConfig(32, 64, P), C.c_void_p(d.data_ptr()), nb, n_streams, n_per, 0,
                                                  C.c_void_p(guard.data_ptr()), stride, C.c_void_p(n_words.data_ptr()), interval,
                                                  C.c_void_p(pos.data_ptr()), C.c_void_p(state.data_ptr()), C.c_void_p(status.data_ptr()), None,
                                                  None), "cst_ans_encode_batch_ckpt_sym")
        assert B.last_kernel() == names["ck"], (tag, B.last_kernel())
    else:
        N.check(lib.cst_ans_encode_batch_sym(model._h, N.CoderConfig(32, 64, P), C.c_void_p(d.data_ptr()), nb, n_streams, n_per, 0,
                                             C.c_void_p(guard.data_ptr()), stride, C.c_void_p(n_words.data_ptr()), None,
                                             C.c_void_p(status.data_ptr()), 0, None, None), "cst_ans_encode_batch_sym")
        assert B.last_kernel() == names["enc"], (tag, B.last_kernel())
    torch.cuda.synchronize()
    got_st, got_n = status.cpu().numpy(), n_words.cpu().numpy()
    assert got_st[check].tolist() == want_st.tolist(), tag
    assert got_n[check].tolist() == np.where(want_st == 0, want_n, 0).tolist(), tag
    words = guard.cpu().numpy().view(np.uint32)
    assert (words[n_streams * stride:] == 0x5A5A5A5A).all(), (tag, "words behind the last slab")
    rows = words[: n_streams * stride].reshape(n_streams, stride)
    for i, s in enumerate(check):
        if want_st[i] == 0:
            assert np.array_equal(rows[s, : want_n[i]], want_words[i, : want_n[i]]), (tag, int(s))
    good = got_st == 0
    if k and good.all():
        some = check[:: max(1, len(check) // 64)]
        wp, ws = O.ans_jump_table(sym[some].astype(np.int32), lo, cdf, P, interval)
        assert np.array_equal(pos.cpu().numpy().view(np.uint32)[some], wp) and np.array_equal(state.cpu().numpy().view(np.uint64)[some], ws), tag
    # decode: whole streams, and chunk by chunk where jump points were noted
    if good.all():
        enc = B.EncodedBatch(guard[: n_streams * stride].view(n_streams, stride), n_words, status, (32, 64, P))
        out = torch.full((n_streams, n_per), 99, dtype=t_t, device="cuda")
        dec, dst = B.ans_decode(enc, model, n_per, out=out, cold=bool(rng.random() < 0.5))
        bucket_entries = n <= 256 or (n <= 1024 and P <= 22)     # (cst_common.hpp bucket16_usable: what the 12 < P decoders' entries can hold)
        want_kernel = names["small"] if big and (n <= 256 or P > 12) else names["dec"]
        assert B.last_kernel() == want_kernel or (nb == 4 and not big) or (P > 12 and not bucket_entries), (tag, B.last_kernel())
        assert int(dst.abs().sum().item()) == 0 and torch.equal(dec, d), tag
        n_small += int(big)
        if k and interval % line == 0:
            out.fill_(98)
            dec, dst = B.ans_decode_checkpointed(enc, B.Checkpoints(interval, pos, state), model, n_per, out=out)
            assert P > 12 or B.last_kernel() in (names["dec"], names["small"]), (tag, B.last_kernel())
            assert int(dst.abs().sum().item()) == 0 and torch.equal(dec, d), tag
            n_jump += 1
    n_cases += 1
    n_streams_total += n_streams
print(f"stress_n8: {n_cases} cases, {n_streams_total} streams agree with the oracle ({n_jump} through jump points, {n_small} on the two-wave decoder)")
