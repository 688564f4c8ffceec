"""Entropy models of `constriction.stream.model` that the stream-coder hot path uses.

Mirrors src/pybindings/stream/model.rs: every model fixes PRECISION = 24 bits, Symbol = i32, Probability = u32
(src/pybindings/stream/model/internals.rs:26-39).  A model is either *concrete* (all parameters given to the
constructor; usable for i.i.d. symbols) or a *family* (parameters passed per symbol to encode/decode).
The cumulative tables of concrete models are built on the GPU (Gaussian, bit-exact f64) or on the host
(categorical "fast" quantisation, a handful of floats) and live in HBM as a `cst_model`.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

PRECISION = 24


def _as_float_params(a, name):
    """f32 arrays are widened to f64 before use (src/pybindings/mod.rs:211-216)."""
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        raise TypeError(f"{name} must be a numpy array with dtype float32 or float64")
    if a.ndim != 1:
        raise ValueError(f"{name} must be a rank-1 array")
    return np.ascontiguousarray(a, dtype=np.float64)


def fast_quantized_cdf(probabilities: np.ndarray, precision: int = PRECISION) -> np.ndarray:
    """`fast_quantized_cdf` + trailing 2^P (src/stream/model/categorical.rs:16-54, contiguous.rs:203-214),
    evaluated in the dtype of `probabilities` exactly as the reference does (f32 stays f32)."""
    p = np.asarray(probabilities)
    if p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    if p.ndim != 1:
        raise ValueError("probabilities must be rank 1")
    n = p.shape[0]
    err = ValueError("Probability distribution not normalizable (the array of probabilities\n"
                     "might be empty, contain negative values or NaNs, or sum to infinity).")
    if n < 2 or n >= (1 << precision) - 1:
        raise err
    dt = p.dtype.type
    csum = np.cumsum(p, dtype=p.dtype)          # sequential accumulation, like Iterator::sum
    norm = csum[-1]
    if not np.isfinite(norm) or not norm >= np.finfo(p.dtype).tiny:
        raise err
    free_weight = dt((1 << precision) - n)
    scale = dt(free_weight / norm)
    cum = np.concatenate(([dt(0)], csum[:-1])).astype(p.dtype)
    prod = (cum * scale).astype(p.dtype)
    with np.errstate(invalid="ignore"):
        left = np.where(prod > 0, np.minimum(prod, dt(4294967295.0)), dt(0)).astype(np.float64)
    left = np.trunc(left).astype(np.uint64)
    cdf = np.empty(n + 1, dtype=np.uint32)
    cdf[:n] = ((left + np.arange(n, dtype=np.uint64)) & 0xFFFFFFFF).astype(np.uint32)
    cdf[n] = 1 << precision
    if np.any(np.diff(cdf.astype(np.int64)) <= 0):
        raise err
    return cdf


class Model:
    """Base class (constriction.stream.model.Model)."""
    _n_params = 0

    def _device_model(self):
        raise ValueError("This model family needs its parameters to be passed to `encode`/`decode`.")

    def is_concrete(self) -> bool:
        return False


class QuantizedGaussian(Model):
    """constriction.stream.model.QuantizedGaussian(min_symbol_inclusive, max_symbol_inclusive, mean=None, std=None)
    (src/pybindings/stream/model.rs:645-708): LeakyQuantizer<f64,i32,u32,24>(min..=max) applied to Gaussian(mean, std)."""
    _n_params = 2

    def __init__(self, min_symbol_inclusive, max_symbol_inclusive, mean=None, std=None):
        lo, hi = int(min_symbol_inclusive), int(max_symbol_inclusive)
        if not hi > lo:
            raise ValueError("The support must contain at least two symbols.")   # quantize.rs:292-294 (assert!)
        if hi - lo + 1 > (1 << PRECISION):
            raise ValueError("The support is too large to assign a nonzero probability to each element.")
        if (mean is None) != (std is None):
            raise ValueError("Either none or both of `mean` and `std` must be specified.")
        self.min_symbol, self.max_symbol = lo, hi
        self.mean = None if mean is None else float(mean)
        self.std = None if std is None else float(std)
        if self.std is not None and not self.std > 0.0:
            raise ValueError("Invalid model parameter: `std` must be positive.")   # model.rs:654-657 (assert!)
        self._dev = None

    def is_concrete(self):
        return self.mean is not None

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.quantized_gaussian(self.min_symbol, self.max_symbol, self.mean, self.std, PRECISION)
        return self._dev


class Categorical(Model):
    """constriction.stream.model.Categorical(probabilities=None, lazy=None, perfect=None)
    (src/pybindings/stream/model.rs:455-578).  Only the `perfect=False` quantisation is on the hot path
    (lazy and non-lazy fast tables are identical, tests/python/test_lazy_f32.py); `perfect=True` is not
    implemented."""
    _n_params = 1

    def __init__(self, probabilities=None, lazy: Optional[bool] = None, perfect: Optional[bool] = None):
        if lazy and perfect:
            raise ValueError("Both arguments `lazy` and `perfect` cannot be set to `True` at the same time.\n"
                             "Lazy categorical entropy models cannot perfectly quantize probabilities.")
        if perfect or (perfect is None and lazy is None):
            raise NotImplementedError(
                "Categorical(perfect=True) (also the reference's legacy default when neither `perfect` nor `lazy` "
                "is given) is outside the accelerated hot path; pass perfect=False.")
        self.probabilities = None
        self._dev = None
        if probabilities is not None:
            p = np.asarray(probabilities)
            if p.dtype not in (np.float32, np.float64):
                raise TypeError("probabilities must have dtype float32 or float64")
            self.probabilities = np.ascontiguousarray(p)
            self.cdf = fast_quantized_cdf(self.probabilities, PRECISION)

    def is_concrete(self):
        return self.probabilities is not None

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(self.cdf, 0, PRECISION)
        return self._dev

    @staticmethod
    def cdf_rows(prob_matrix) -> np.ndarray:
        """One fast-quantised cdf row per symbol for the family form (rank-2 probabilities)."""
        m = np.asarray(prob_matrix)
        if m.ndim != 2:
            raise ValueError("expected a rank-2 array of probabilities (one row per symbol)")
        if m.dtype not in (np.float32, np.float64):
            raise TypeError("probabilities must have dtype float32 or float64")
        return np.stack([fast_quantized_cdf(row, PRECISION) for row in m]) if len(m) else np.zeros((0, m.shape[1] + 1), np.uint32)


def leaky_cdf_table(cdf, min_symbol: int, max_symbol: int, params=(), precision: int = PRECISION) -> np.ndarray:
    """`LeakilyQuantizedDistribution::left_cumulative_and_probability` for every symbol of the support with an
    arbitrary continuous CDF (src/stream/model/quantize.rs:525-568): L[0] = 0,
    L[i] = trunc_sat(free_weight * cdf(sym_i - 0.5)) + i, L[n] = 2^P, in f64 like the reference (the CDF is a Python
    callable there too: src/pybindings/stream/model/internals.rs:283-398).  Evaluated on the host; the table then
    goes to the GPU like any other tabulated model."""
    lo, hi = int(min_symbol), int(max_symbol)
    n = hi - lo + 1
    free_weight = float(((1 << precision) - 1) - (hi - lo))
    out = np.empty(n + 1, dtype=np.uint32)
    out[0] = 0
    for i in range(1, n):
        x = float(cdf(float(lo + i) - 0.5, *params)) * free_weight
        # Rust `as u32`: truncate toward zero, saturate, NaN -> 0
        v = 0 if not x > 0.0 else (0xFFFFFFFF if x >= 4294967295.0 else int(x))
        out[i] = (v + i) & 0xFFFFFFFF
    out[n] = 1 << precision
    if np.any(np.diff(out.astype(np.int64)) <= 0):
        raise ValueError("Invalid model: the cumulative distribution function is not monotonically increasing "
                         "on the support (quantize.rs:560-566).")
    return out


class CustomModel(Model):
    """constriction.stream.model.CustomModel(cdf, approximate_inverse_cdf, min_symbol_inclusive, max_symbol_inclusive)
    (src/pybindings/stream/model.rs:264-318): LeakyQuantizer<f64,i32,u32,24> over a user-provided CDF.  Usable as a
    concrete model (no parameters at encode/decode time) or as a family whose parameters (any number of rank-1
    float arrays) are forwarded to `cdf(x, *params_i)`.  `approximate_inverse_cdf` is only a search hint in the
    reference (quantize.rs:204-214) and cannot change results; it is accepted and ignored."""

    def __init__(self, cdf, approximate_inverse_cdf, min_symbol_inclusive, max_symbol_inclusive):
        lo, hi = int(min_symbol_inclusive), int(max_symbol_inclusive)
        if not hi > lo:
            raise ValueError("The support must contain at least two symbols.")
        if hi - lo + 1 > (1 << PRECISION):
            raise ValueError("The support is too large to assign a nonzero probability to each element.")
        self.cdf, self.approximate_inverse_cdf = cdf, approximate_inverse_cdf
        self.min_symbol, self.max_symbol = lo, hi
        self._dev = None

    def is_concrete(self):
        return True      # (decided per call: with parameters it acts as a family, see _single.model_args)

    def _device_model(self):
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(leaky_cdf_table(self.cdf, self.min_symbol, self.max_symbol), self.min_symbol,
                                               PRECISION)
        return self._dev

    def cdf_rows(self, params) -> np.ndarray:
        """one tabulated row per symbol position for the family form"""
        arrays = [_as_float_params(p, "model parameter") for p in params]
        if any(len(a) != len(arrays[0]) for a in arrays):
            raise ValueError("Model parameters have unequal lengths.")
        n = self.max_symbol - self.min_symbol + 1
        if len(arrays[0]) == 0:
            return np.zeros((0, n + 1), np.uint32)
        return np.stack([leaky_cdf_table(self.cdf, self.min_symbol, self.max_symbol, tuple(float(a[t]) for a in arrays))
                         for t in range(len(arrays[0]))])


class ScipyModel(CustomModel):
    """constriction.stream.model.ScipyModel(scipy_model, min_symbol_inclusive, max_symbol_inclusive)
    (src/pybindings/stream/model.rs:320-349): CustomModel(scipy_model.cdf, scipy_model.ppf, min, max)."""

    def __init__(self, scipy_model, min_symbol_inclusive, max_symbol_inclusive):
        super().__init__(scipy_model.cdf, scipy_model.ppf, min_symbol_inclusive, max_symbol_inclusive)
