"""ctypes front end of the CPU oracle (oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package ``constriction_amd`` never imports
this module.  The oracle restates the reference's arithmetic (citations in oracle.c) and is
pinned by tests/golden/reference_vectors.json.

The shared library is (re)built on demand with the Makefile next to this file.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIBS: dict = {}

u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(native: bool = False) -> Path:
    """Compile oracle.c + oracle_families.c if the .so is missing or older than the sources."""
    name = "liboracle_native.so" if native else "liboracle.so"
    so = _HERE / "_build" / name
    newest = max((_HERE / f).stat().st_mtime for f in ("oracle.c", "oracle_families.c", "Makefile"))
    if not so.exists() or so.stat().st_mtime < newest:
        subprocess.run(["make", "-C", str(_HERE), "native" if native else "all"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def load(native: bool = False):
    key = bool(native)
    if key in _LIBS:
        return _LIBS[key]
    try:
        so = build(native)
    except Exception:
        if not native:
            raise
        so = build(False)
    lib = C.CDLL(str(so))
    d, i, u, z, vp = C.c_double, C.c_int, C.c_uint32, C.c_size_t, C.c_void_p
    i32, i64, u64 = C.c_int32, C.c_int64, C.c_uint64
    sig = {
        "cst_oracle_exp": (d, [d]),
        "cst_oracle_erf": (d, [d]),
        "cst_oracle_gaussian_cdf": (d, [d, d, d]),
        "cst_oracle_leaky_gaussian_lcp": (i, [i32, i32, i32, i, i, d, d, C.POINTER(u), C.POINTER(u)]),
        "cst_oracle_leaky_gaussian_cdf_table": (i, [i32, i32, i, i, d, d, u32p]),
        "cst_oracle_leaky_gaussian_quantile": (i, [u, i32, i32, i, i, d, d, C.POINTER(i32), C.POINTER(u), C.POINTER(u)]),
        "cst_oracle_categorical_fast_cdf_f64": (i, [f64p, i, i, u32p]),
        "cst_oracle_categorical_fast_cdf_f32": (i, [f32p, i, i, u32p]),
        "cst_oracle_lookup_from_cdf": (None, [u32p, i, i, u16p]),
        "cst_oracle_ans_new": (vp, [i, i]),
        "cst_oracle_ans_free": (None, [vp]),
        "cst_oracle_ans_clear": (None, [vp]),
        "cst_oracle_ans_state": (u64, [vp]),
        "cst_oracle_ans_bulk_len": (z, [vp]),
        "cst_oracle_ans_is_empty": (i, [vp]),
        "cst_oracle_ans_from_compressed": (vp, [i, i, u32p, z]),
        "cst_oracle_ans_from_binary": (vp, [i, i, u32p, z]),
        "cst_oracle_ans_num_words": (z, [vp]),
        "cst_oracle_ans_num_valid_bits": (z, [vp]),
        "cst_oracle_ans_get_compressed": (z, [vp, vp]),
        "cst_oracle_ans_encode_cp": (None, [vp, u, u, i]),
        "cst_oracle_ans_peek_quantile": (u, [vp, i]),
        "cst_oracle_ans_decode_advance": (None, [vp, u, u, i]),
        "cst_oracle_ans_seek": (i, [vp, z, u64]),
        "cst_oracle_ans_encode_iid_table_reverse": (i64, [vp, i32p, z, i32, u32p, i, i]),
        "cst_oracle_ans_decode_iid_table": (None, [vp, i32p, z, i32, u32p, i, i]),
        "cst_oracle_ans_encode_gaussian_reverse": (i64, [vp, i32p, z, i32, i32, f64p, f64p, i, i, i]),
        "cst_oracle_ans_decode_gaussian": (None, [vp, i32p, z, i32, i32, f64p, f64p, i, i, i]),
        "cst_oracle_rc_encoder_new": (vp, [i, i]),
        "cst_oracle_rc_encoder_free": (None, [vp]),
        "cst_oracle_rc_encode_cp": (i, [vp, u, u, i]),
        "cst_oracle_rc_encoder_pos": (z, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "cst_oracle_rc_get_compressed": (z, [vp, vp]),
        "cst_oracle_rc_decoder_new": (vp, [i, i, u32p, z]),
        "cst_oracle_rc_decoder_free": (None, [vp]),
        "cst_oracle_rc_peek_quantile": (u, [vp, i]),
        "cst_oracle_rc_decode_advance": (None, [vp, u, u, i]),
        "cst_oracle_rc_maybe_exhausted": (i, [vp]),
        "cst_oracle_ans_encode_batch": (None, [i, i, i, i32p, z, z, i32, i, u32p, i, u32p, z, u32p, i32p, i]),
        "cst_oracle_ans_decode_batch": (None, [i, i, i, i32p, z, z, i32, i, u32p, vp, i, u32p, z, u32p, i32p, i]),
        "cst_oracle_synth_symbols": (None, [u64, z, z, z, i32, i, u32p, i, i, i32p]),
        "cst_oracle_rc_encode_batch": (None, [i, i, i, i32p, z, z, i32, i, u32p, u32p, z, u32p, i32p]),
        "cst_oracle_rc_decode_batch": (None, [i, i, i, i32p, z, z, i32, i, u32p, u32p, z, u32p, i32p]),
        "cst_oracle_ans_jump_table": (None, [i, i, i, i32p, z, z, i32, i, u32p, i, z, u32p, u64p]),
        "cst_oracle_ans_decode_from": (None, [i, i, i, u32p, z, u64, i32p, z, i32, i, u32p]),
        "cst_oracle_rc_jump_table": (None, [i, i, i, i32p, z, z, i32, i, u32p, z, u32p, u64p, u64p]),
        "cst_oracle_rc_decode_from": (i, [i, i, i, u32p, z, z, u64, u64, i32p, z, i32, i, u32p]),
        "cst_oracle_fill_threads": (None, [vp, z, i]),
        # oracle_families.c
        "cst_oracle_log": (d, [d]),
        "cst_oracle_log1p": (d, [d]),
        "cst_oracle_atan": (d, [d]),
        "cst_oracle_lgamma": (d, [d]),
        "cst_oracle_laplace_cdf": (d, [d, d, d]),
        "cst_oracle_cauchy_cdf": (d, [d, d, d]),
        "cst_oracle_binomial_cdf": (d, [d, i32, d]),
        "cst_oracle_inc_beta": (d, [d, d, d, d]),
        "cst_oracle_leaky_family_cdf_table": (i, [i, i32, i32, i, d, d, u32p]),
        "cst_oracle_categorical_perfect_cdf": (i, [f64p, i64, i, u32p]),
        "cst_oracle_lazy_categorical_lcp": (i, [vp, i64, i, i, i64, C.POINTER(u), C.POINTER(u)]),
        "cst_oracle_lazy_categorical_quantile": (i, [vp, i64, i, i, u, C.POINTER(i64), C.POINTER(u), C.POINTER(u)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIBS[key] = lib
    return lib


# ---------------------------------------------------------------------------------------------
# models: every model answers (left, prob) for a symbol and (symbol, left, prob) for a quantile
# ---------------------------------------------------------------------------------------------

class GaussianModel:
    """LeakyQuantizer<f64,i32,u{prob_bits},P>(lo..=hi) x Gaussian(mean, std)
    (src/stream/model/quantize.rs:284-308, 525-568)."""

    def __init__(self, lo, hi, mean, std, P=24, prob_bits=32):
        self.lo, self.hi, self.mean, self.std, self.P, self.prob_bits = int(lo), int(hi), float(mean), float(std), P, prob_bits

    def lcp(self, sym):
        lib = load()
        l, p = C.c_uint32(), C.c_uint32()
        rc = lib.cst_oracle_leaky_gaussian_lcp(int(sym), self.lo, self.hi, self.P, self.prob_bits, self.mean,
                                               self.std, C.byref(l), C.byref(p))
        if rc == 1:
            raise KeyError("impossible symbol")
        if rc == 2:
            raise ArithmeticError("zero probability")
        return l.value, p.value

    def quantile(self, q):
        lib = load()
        s, l, p = C.c_int32(), C.c_uint32(), C.c_uint32()
        lib.cst_oracle_leaky_gaussian_quantile(int(q), self.lo, self.hi, self.P, self.prob_bits, self.mean,
                                               self.std, C.byref(s), C.byref(l), C.byref(p))
        return s.value, l.value, p.value

    def cdf_table(self):
        n = self.hi - self.lo + 1
        cdf = np.zeros(n + 1, dtype=np.uint32)
        rc = load().cst_oracle_leaky_gaussian_cdf_table(self.lo, self.hi, self.P, self.prob_bits, self.mean,
                                                        self.std, cdf)
        if rc:
            raise ArithmeticError("zero probability")
        return cdf


class TableModel:
    """Tabulated model: symbols lo..lo+n-1, cdf[n+1] (contiguous.rs:673-700 encode,
    lookup_contiguous.rs:564-605 decode)."""

    def __init__(self, cdf, lo=0, P=24):
        self.cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        self.lo, self.P = int(lo), P
        self.n = len(self.cdf) - 1
        self._mask = (1 << P) - 1 if P < 32 else 0xFFFFFFFF

    def lcp(self, sym):
        i = int(sym) - self.lo
        if i < 0 or i >= self.n:
            raise KeyError("impossible symbol")
        return int(self.cdf[i]), (int(self.cdf[i + 1]) - int(self.cdf[i])) & 0xFFFFFFFF

    def quantile(self, q):
        i = int(np.searchsorted(self.cdf[: self.n], q, side="right")) - 1
        return self.lo + i, int(self.cdf[i]), (int(self.cdf[i + 1]) - int(self.cdf[i])) & 0xFFFFFFFF


def categorical_fast_cdf(probs, P=24):
    """fast_quantized_cdf + trailing 2^P (categorical.rs:16-54, contiguous.rs:203-214)."""
    probs = np.asarray(probs)
    n = len(probs)
    cdf = np.zeros(n + 1, dtype=np.uint32)
    if probs.dtype == np.float32:
        rc = load().cst_oracle_categorical_fast_cdf_f32(np.ascontiguousarray(probs), n, P, cdf)
    else:
        rc = load().cst_oracle_categorical_fast_cdf_f64(np.ascontiguousarray(probs, dtype=np.float64), n, P, cdf)
    if rc:
        raise ValueError("Probability distribution not normalizable")
    return cdf


FAMILY_LAPLACE, FAMILY_CAUCHY, FAMILY_BINOMIAL = 1, 2, 3


def leaky_family_cdf(family, lo, hi, a, b=0.0, P=24):
    """LeakyQuantizer<f64,i32,u32,P>(lo..=hi) over Laplace(a, b) / Cauchy(a, b) / Binomial(n = hi, p = a), tabulated
    (oracle_families.c; quantize.rs:525-568 over the `probability` crate's CDFs)."""
    cdf = np.zeros(int(hi) - int(lo) + 2, dtype=np.uint32)
    rc = load().cst_oracle_leaky_family_cdf_table(int(family), int(lo), int(hi), P, float(a), float(b), cdf)
    if rc:
        raise ArithmeticError("zero probability")
    return cdf


def categorical_perfect_cdf(probs, P=24):
    """perfectly_quantized_probabilities + cumulation (categorical.rs:56-177, contiguous.rs:301-313); f32 inputs are
    widened to f64 first (`F: Into<f64>`)."""
    probs = np.ascontiguousarray(np.asarray(probs), dtype=np.float64)
    cdf = np.zeros(len(probs) + 1, dtype=np.uint32)
    if load().cst_oracle_categorical_perfect_cdf(probs, len(probs), P, cdf):
        raise ValueError("Probability distribution not normalizable")
    return cdf


class LazyCategoricalModel:
    """LazyContiguousCategoricalEntropyModel<u32, F, _, P> (lazy_contiguous.rs:132-331): nothing is tabulated, every
    call sums the probabilities it needs in F = the dtype of `probs` (f32 stays f32)."""

    def __init__(self, probs, P=24, lo=0):
        probs = np.asarray(probs)
        self.is_f32 = probs.dtype == np.float32
        self.probs = np.ascontiguousarray(probs, dtype=np.float32 if self.is_f32 else np.float64)
        self.P, self.lo, self.n = P, int(lo), len(self.probs)

    def lcp(self, sym):
        l, p = C.c_uint32(), C.c_uint32()
        rc = load().cst_oracle_lazy_categorical_lcp(self.probs.ctypes.data, self.n, int(self.is_f32), self.P,
                                                    int(sym) - self.lo, C.byref(l), C.byref(p))
        if rc == 1:
            raise KeyError("impossible symbol")
        if rc:
            raise ValueError("Probability distribution not normalizable")
        return l.value, p.value

    def quantile(self, q):
        s, l, p = C.c_int64(), C.c_uint32(), C.c_uint32()
        if load().cst_oracle_lazy_categorical_quantile(self.probs.ctypes.data, self.n, int(self.is_f32), self.P, int(q),
                                                       C.byref(s), C.byref(l), C.byref(p)):
            raise ValueError("Probability distribution not normalizable")
        return self.lo + s.value, l.value, p.value


def lookup_from_cdf(cdf, P):
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    lut = np.zeros(1 << P, dtype=np.uint16)
    load().cst_oracle_lookup_from_cdf(cdf, len(cdf) - 1, P, lut)
    return lut


# ---------------------------------------------------------------------------------------------
# coders
# ---------------------------------------------------------------------------------------------

class AnsCoder:
    """stream::stack::AnsCoder<W,S> (src/stream/stack.rs) restated; single stream."""

    def __init__(self, compressed=None, seal=False, W=32, S=64):
        self.W, self.S = W, S
        lib = load()
        if compressed is None:
            self._h = lib.cst_oracle_ans_new(W, S)
        else:
            words = np.ascontiguousarray(compressed, dtype=np.uint32)
            if seal:
                self._h = lib.cst_oracle_ans_from_binary(W, S, words, len(words))
            else:
                self._h = lib.cst_oracle_ans_from_compressed(W, S, words, len(words))
                if not self._h:
                    raise ValueError("Invalid compressed data: ANS compressed data never ends in a zero word.")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and load is not None:            # (module globals are torn down before objects at interpreter exit)
            load().cst_oracle_ans_free(h)

    # model-generic per-symbol paths
    def encode_reverse(self, symbols, models, P=24):
        """models: one model (iid) or a list with one model per symbol."""
        symbols = np.atleast_1d(np.asarray(symbols))
        lib = load()
        for t in range(len(symbols) - 1, -1, -1):
            m = models[t] if isinstance(models, (list, tuple)) else models
            l, p = m.lcp(symbols[t])
            lib.cst_oracle_ans_encode_cp(self._h, l, p, P)

    def decode(self, models, n=None, P=24):
        lib = load()
        if isinstance(models, (list, tuple)):
            n = len(models)
        out = np.zeros(n, dtype=np.int32)
        for t in range(n):
            m = models[t] if isinstance(models, (list, tuple)) else models
            q = lib.cst_oracle_ans_peek_quantile(self._h, P)
            s, l, p = m.quantile(q)
            out[t] = s
            lib.cst_oracle_ans_decode_advance(self._h, l, p, P)
        return out

    # fast whole-array paths (C loops)
    def encode_iid_table_reverse(self, symbols, cdf, lo, P):
        symbols = np.ascontiguousarray(symbols, dtype=np.int32)
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        rc = load().cst_oracle_ans_encode_iid_table_reverse(self._h, symbols, len(symbols), lo, cdf, len(cdf) - 1, P)
        if rc:
            raise KeyError(f"impossible symbol at index {rc - 1}")

    def decode_iid_table(self, n, cdf, lo, P):
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        out = np.zeros(n, dtype=np.int32)
        load().cst_oracle_ans_decode_iid_table(self._h, out, n, lo, cdf, len(cdf) - 1, P)
        return out

    def encode_gaussian_reverse(self, symbols, lo, hi, means, stds, P=24, prob_bits=32):
        symbols = np.ascontiguousarray(symbols, dtype=np.int32)
        means = np.ascontiguousarray(np.atleast_1d(means), dtype=np.float64)
        stds = np.ascontiguousarray(np.atleast_1d(stds), dtype=np.float64)
        iid = int(len(means) == 1 and len(symbols) != 1)
        rc = load().cst_oracle_ans_encode_gaussian_reverse(self._h, symbols, len(symbols), lo, hi, means, stds, iid,
                                                           P, prob_bits)
        if rc:
            raise KeyError(f"impossible symbol at index {rc - 1}")

    def decode_gaussian(self, n, lo, hi, means, stds, P=24, prob_bits=32):
        means = np.ascontiguousarray(np.atleast_1d(means), dtype=np.float64)
        stds = np.ascontiguousarray(np.atleast_1d(stds), dtype=np.float64)
        iid = int(len(means) == 1 and n != 1)
        out = np.zeros(n, dtype=np.int32)
        load().cst_oracle_ans_decode_gaussian(self._h, out, n, lo, hi, means, stds, iid, P, prob_bits)
        return out

    def get_compressed(self):
        lib = load()
        n = lib.cst_oracle_ans_get_compressed(self._h, None)
        out = np.zeros(n, dtype=np.uint32)
        if n:
            lib.cst_oracle_ans_get_compressed(self._h, out.ctypes.data)
        return out

    def num_words(self):
        return load().cst_oracle_ans_num_words(self._h)

    def num_bits(self):
        return self.W * self.num_words()

    def num_valid_bits(self):
        return load().cst_oracle_ans_num_valid_bits(self._h)

    def is_empty(self):
        return bool(load().cst_oracle_ans_is_empty(self._h))

    def pos(self):
        lib = load()
        return lib.cst_oracle_ans_bulk_len(self._h), lib.cst_oracle_ans_state(self._h)

    def seek(self, pos, state):
        if load().cst_oracle_ans_seek(self._h, pos, state):
            raise ValueError("Tried to seek past end of stream.")

    def clear(self):
        load().cst_oracle_ans_clear(self._h)


class RangeEncoder:
    """stream::queue::RangeEncoder<W,S> (src/stream/queue.rs:612-705, seal 458-523)."""

    def __init__(self, W=32, S=64):
        self.W, self.S = W, S
        self._h = load().cst_oracle_rc_encoder_new(W, S)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and load is not None:            # (module globals are torn down before objects at interpreter exit)
            load().cst_oracle_rc_encoder_free(h)

    def encode(self, symbols, models, P=24):
        symbols = np.atleast_1d(np.asarray(symbols))
        lib = load()
        for t in range(len(symbols)):
            m = models[t] if isinstance(models, (list, tuple)) else models
            l, p = m.lcp(symbols[t])
            if lib.cst_oracle_rc_encode_cp(self._h, l, p, P):
                raise KeyError("impossible symbol")

    def pos(self):
        """(words emitted incl. held-back ones, (lower, range)): RangeEncoder::pos, queue.rs:182-196"""
        lo, rg = C.c_uint64(), C.c_uint64()
        n = load().cst_oracle_rc_encoder_pos(self._h, C.byref(lo), C.byref(rg))
        return (int(n), (int(lo.value), int(rg.value)))

    def get_compressed(self):
        lib = load()
        n = lib.cst_oracle_rc_get_compressed(self._h, None)
        out = np.zeros(n, dtype=np.uint32)
        if n:
            lib.cst_oracle_rc_get_compressed(self._h, out.ctypes.data)
        return out

    def num_words(self):
        return load().cst_oracle_rc_get_compressed(self._h, None)

    def num_bits(self):
        return self.W * self.num_words()


class RangeDecoder:
    """stream::queue::RangeDecoder<W,S> (src/stream/queue.rs:847-868, 968-1033)."""

    def __init__(self, compressed, W=32, S=64):
        self.W, self.S = W, S
        self._words = np.ascontiguousarray(compressed, dtype=np.uint32)
        self._h = load().cst_oracle_rc_decoder_new(W, S, self._words, len(self._words))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and load is not None:            # (module globals are torn down before objects at interpreter exit)
            load().cst_oracle_rc_decoder_free(h)

    def decode(self, models, n=None, P=24):
        lib = load()
        if isinstance(models, (list, tuple)):
            n = len(models)
        out = np.zeros(n, dtype=np.int32)
        for t in range(n):
            m = models[t] if isinstance(models, (list, tuple)) else models
            q = lib.cst_oracle_rc_peek_quantile(self._h, P)
            if q == 0xFFFFFFFF:
                raise AssertionError("Tried to decode from compressed data that is invalid for the employed entropy model.")
            s, l, p = m.quantile(q)
            out[t] = s
            lib.cst_oracle_rc_decode_advance(self._h, l, p, P)
        return out

    def maybe_exhausted(self):
        return bool(load().cst_oracle_rc_maybe_exhausted(self._h))


# ---------------------------------------------------------------------------------------------
# batched drivers (CPU baseline) and the synthetic workload
# ---------------------------------------------------------------------------------------------

class ChainCoder:
    """stream::chain::ChainCoder<Word, State, Vec<Word>, Vec<Word>, P> (src/stream/chain.rs:231-246) restated in Python
    integers -- small cases only.  `models`: one model (iid) or one per symbol; a model answers lcp(sym) and quantile(q).
    Constructors: from_binary (:326-346, seal), from_compressed (:358-377), from_remainders (:430-456)."""

    class OutOfCompressedData(Exception):
        pass

    class OutOfRemainders(Exception):
        pass

    def __init__(self, data, is_remainders=False, seal=False, W=32, S=64, P=24):
        self.W, self.S, self.P = W, S, P
        data = [int(x) for x in np.asarray(data).tolist()]
        if is_remainders:
            if seal:
                raise AssertionError("Cannot seal remainders data.")
            if not data or data[-1] == 0:                       # chain.rs:438-441
                raise ValueError("Too little data provided, or provided data ends in zero word and `is_remainders==True`.")
            head = data.pop()
            self.remainders, self.compressed = data, []
            self.rem_head = self._new_heads(self.remainders, False)
            self.comp_head = head
        else:
            self.compressed, self.remainders = data, []
            self.rem_head = self._new_heads(self.compressed, seal)
            self.comp_head = 1

    def _new_heads(self, source, push_one):                     # ChainCoderHeads::new, chain.rs:270-303
        threshold = 1 << (self.S - self.W - self.P)
        if push_one:
            head = 1
        else:
            if not source or source[-1] == 0:
                raise ValueError("Too little data provided, or provided data ends in zero word.")
            head = source.pop()
        while head < threshold:
            if not source:
                raise ValueError("Too little data provided.")
            head = (head << self.W) | source.pop()
        return head

    def _models(self, models, n):
        return models if isinstance(models, (list, tuple)) else [models] * n

    def decode(self, models, n=None):                           # decode_symbol, chain.rs:1044-1122
        W, S, P = self.W, self.S, self.P
        if n is None:
            n = len(models)
        out = []
        for m in self._models(models, n):
            if P == W or self.comp_head < (1 << P):
                if not self.compressed:
                    raise ChainCoder.OutOfCompressedData()
                word = self.compressed.pop()
                if P != W:
                    self.comp_head = ((self.comp_head << (W - P)) | (word >> P)) & ((1 << W) - 1)
            else:
                word = self.comp_head
                self.comp_head >>= P
            quantile = word & ((1 << P) - 1) if P != W else word
            sym, left, prob = m.quantile(quantile)
            self.rem_head = self.rem_head * prob + (quantile - left)
            if self.rem_head >= (1 << (S - P)):
                self.remainders.append(self.rem_head & ((1 << W) - 1))     # flush_remainders_head, :784-796
                self.rem_head >>= W
            out.append(sym)
        return np.array(out, dtype=np.int32)

    def encode_reverse(self, symbols, models):                  # encode_symbol, chain.rs:1140-1209, last symbol first
        W, S, P = self.W, self.S, self.P
        symbols = [int(x) for x in np.atleast_1d(np.asarray(symbols)).tolist()]
        ms = self._models(models, len(symbols))
        for sym, m in zip(reversed(symbols), reversed(ms)):
            left, prob = m.lcp(sym)
            if self.rem_head < (prob << (S - W - P)):
                if not self.remainders:
                    raise ChainCoder.OutOfRemainders()
                self.rem_head = ((self.rem_head << W) | self.remainders.pop()) & ((1 << S) - 1)   # refill, :799-815
            remainder = self.rem_head % prob
            quantile = left + remainder
            self.rem_head //= prob
            if P != W and self.comp_head < (1 << (W - P)):
                self.comp_head = (self.comp_head << P) | quantile
            else:
                if P == W:
                    word = quantile
                else:
                    word = ((self.comp_head << P) | quantile) & ((1 << W) - 1)
                    self.comp_head >>= (W - P)
                self.compressed.append(word)

    def is_whole(self):
        return self.comp_head == 1

    def get_remainders(self):                                   # into_remainders, chain.rs:406-423 -> (compressed, remainders)
        rem, head = list(self.remainders), self.rem_head
        while head != 0:
            rem.append(head & ((1 << self.W) - 1))
            head >>= self.W
        rem.append(self.comp_head)
        return np.array(self.compressed, dtype=np.uint32), np.array(rem, dtype=np.uint32)

    def get_data(self, unseal=False):                           # into_compressed :475-496 / into_binary :516-541 -> (remainders, compressed)
        if not self.is_whole() or (unseal and (self.rem_head.bit_length() - 1) % self.W != 0):
            raise AssertionError("Fractional number of words in compressed or remainders data.")
        comp, head = list(self.compressed), self.rem_head
        while head > (1 if unseal else 0):
            comp.append(head & ((1 << self.W) - 1))
            head >>= self.W
        return np.array(self.remainders, dtype=np.uint32), np.array(comp, dtype=np.uint32)


def synth_symbols(seed, stream_begin, n_streams, n_per_stream, lo, cdf, P, per_stream_tables=False):
    """q = splitmix64(seed ^ stream).next() >> (64-P); sym = quantile(q)   (SURVEY.md 8d)."""
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    n_sym = (cdf.shape[-1]) - 1
    out = np.zeros((n_streams, n_per_stream), dtype=np.int32)
    load().cst_oracle_synth_symbols(seed, stream_begin, n_streams, n_per_stream, lo, n_sym, cdf.reshape(-1),
                                    int(per_stream_tables), P, out.reshape(-1))
    return out


def slab_stride(n_per_stream, P, W=32, S=64):
    """Upper bound on words per stream: at most one word per symbol, and at most
    ceil(n*P/W) from the information content, plus the S/W state words."""
    return min(n_per_stream, (n_per_stream * P + W - 1) // W) + S // W


def ans_encode_batch(symbols, lo, cdf, P, W=32, S=64, stride=None, n_threads=1, native=False, out=None):
    """out = (words, n_words, status) of an earlier call: reuse those buffers (bench.py's cpu_baseline leg keeps the
    allocation and the first touch of a gigabyte of pages out of its timed region)"""
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    n_streams, n = symbols.shape
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    per_stream = cdf.ndim == 2
    n_sym = cdf.shape[-1] - 1
    stride = stride or slab_stride(n, P, W, S)
    if out is not None:
        words, n_words, status = out
        assert words.shape == (n_streams, stride) and words.dtype == np.uint32 and words.flags.c_contiguous
    else:
        words = np.zeros((n_streams, stride), dtype=np.uint32)
        n_words = np.zeros(n_streams, dtype=np.uint32)
        status = np.zeros(n_streams, dtype=np.int32)
    load(native).cst_oracle_ans_encode_batch(W, S, P, symbols.reshape(-1), n_streams, n, lo, n_sym, cdf.reshape(-1),
                                             int(per_stream), words.reshape(-1), stride, n_words, status, n_threads)
    return words, n_words, status


def ans_jump_table(symbols, lo, cdf, P, interval, W=32, S=64):
    """AnsCoder::pos() in front of every chunk of `interval` symbols (stack.rs:1130-1139): (pos [n_streams, n_chunks] uint32,
    state [n_streams, n_chunks] uint64).  cdf: one table or one per stream."""
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    n_streams, n = symbols.shape
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    n_chunks = (n + interval - 1) // interval
    pos = np.zeros((n_streams, n_chunks), np.uint32)
    state = np.zeros((n_streams, n_chunks), np.uint64)
    load().cst_oracle_ans_jump_table(W, S, P, symbols.reshape(-1), n_streams, n, lo, cdf.shape[-1] - 1, cdf.reshape(-1), int(cdf.ndim == 2),
                                     interval, pos.reshape(-1), state.reshape(-1))
    return pos, state


def ans_decode_from(words, pos, state, n, lo, cdf, P, W=32, S=64):
    """AnsCoder::seek(pos, state) + n symbols on ONE stream's words (stack.rs:1117-1128)"""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    out = np.zeros(n, np.int32)
    load().cst_oracle_ans_decode_from(W, S, P, words, int(pos), int(state), out, n, lo, len(cdf) - 1, cdf)
    return out


def range_jump_table(symbols, lo, cdf, P, interval, W=32, S=64):
    """RangeEncoder::pos() in front of every chunk (queue.rs:182-196): (pos, lower, range), each [n_streams, n_chunks]"""
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    n_streams, n = symbols.shape
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    n_chunks = (n + interval - 1) // interval
    pos = np.zeros((n_streams, n_chunks), np.uint32)
    lower, rng = np.zeros((n_streams, n_chunks), np.uint64), np.zeros((n_streams, n_chunks), np.uint64)
    load().cst_oracle_rc_jump_table(W, S, P, symbols.reshape(-1), n_streams, n, lo, len(cdf) - 1, cdf, interval, pos.reshape(-1),
                                    lower.reshape(-1), rng.reshape(-1))
    return pos, lower, rng


def range_decode_from(words, pos, lower, rng, n, lo, cdf, P, W=32, S=64):
    """RangeDecoder::seek((pos, (lower, range))) + n symbols on ONE stream's words (queue.rs:911-926); returns (symbols, status)"""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    out = np.zeros(n, np.int32)
    st = load().cst_oracle_rc_decode_from(W, S, P, words, len(words), int(pos), int(lower), int(rng), out, n, lo, len(cdf) - 1, cdf)
    return out, st


def ans_decode_batch(words, n_words, n_per_stream, lo, cdf, P, W=32, S=64, lookup=None, n_threads=1, native=False, out=None):
    """out = (decoded, status) of an earlier call: reuse those buffers (see ans_encode_batch)"""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    n_streams, stride = words.shape
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    per_stream = cdf.ndim == 2
    n_sym = cdf.shape[-1] - 1
    if out is not None:
        out, status = out
        assert out.shape == (n_streams, n_per_stream) and out.dtype == np.int32 and out.flags.c_contiguous
    else:
        out = np.zeros((n_streams, n_per_stream), dtype=np.int32)
        status = np.zeros(n_streams, dtype=np.int32)
    lut_ptr = None
    if lookup is not None:
        lookup = np.ascontiguousarray(lookup, dtype=np.uint16)
        lut_ptr = lookup.ctypes.data
    load(native).cst_oracle_ans_decode_batch(W, S, P, out.reshape(-1), n_streams, n_per_stream, lo, n_sym,
                                             cdf.reshape(-1), lut_ptr, int(per_stream), words.reshape(-1), stride,
                                             np.ascontiguousarray(n_words, dtype=np.uint32), status, n_threads)
    return out, status


def range_max_words(n_per_stream, P, W=32, S=64):
    return min(n_per_stream, (n_per_stream * P + W - 1) // W) + 2


def rc_encode_batch(symbols, lo, cdf, P, W=32, S=64, stride=None):
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    n_streams, n = symbols.shape
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    stride = stride or range_max_words(n, P, W, S)
    words = np.zeros((n_streams, stride), dtype=np.uint32)
    n_words = np.zeros(n_streams, dtype=np.uint32)
    status = np.zeros(n_streams, dtype=np.int32)
    load().cst_oracle_rc_encode_batch(W, S, P, symbols.reshape(-1), n_streams, n, lo, len(cdf) - 1, cdf, words.reshape(-1),
                                      stride, n_words, status)
    return words, n_words, status


def rc_decode_batch(words, n_words, n_per_stream, lo, cdf, P, W=32, S=64):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    n_streams, stride = words.shape
    cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    out = np.zeros((n_streams, n_per_stream), dtype=np.int32)
    status = np.zeros(n_streams, dtype=np.int32)
    load().cst_oracle_rc_decode_batch(W, S, P, out.reshape(-1), n_streams, n_per_stream, lo, len(cdf) - 1, cdf,
                                      words.reshape(-1), stride, np.ascontiguousarray(n_words, dtype=np.uint32), status)
    return out, status
