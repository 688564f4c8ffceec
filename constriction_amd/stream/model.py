"""Entropy models of `constriction.stream.model` that the stream-coder hot path uses.

Mirrors src/pybindings/stream/model.rs: every model fixes PRECISION = 24 bits, Symbol = i32, Probability = u32
(src/pybindings/stream/model/internals.rs:26-39).  A model is either *concrete* (all parameters given to the
constructor; usable for i.i.d. symbols) or a *family* (parameters passed per symbol to encode/decode).
The cumulative tables are built by the HIP library: on the GPU in bit-exact f64 (Gaussian: cst_model_create_gaussian;
Laplace / Cauchy / Binomial: cst_family_cdf_rows, one thread per table entry), in host C++ where the algorithm is a
sequential search (Categorical(perfect=True): cst_categorical_perfect_cdf), and here in numpy only for the "fast"
categorical quantisation (two vector operations in the caller's dtype) and for user-supplied Python CDFs
(CustomModel / ScipyModel: the CDF is a Python callable in the reference too).  They live in HBM as a `cst_model`.

What pins which family (tests/golden/*.json hold the reference's own vectors):
  QuantizedGaussian, Categorical(perfect=False / lazy=True), CustomModel, ScipyModel   -- golden vectors
  Uniform                                                                             -- integer arithmetic only (uniform.rs)
  Categorical(perfect=True), Bernoulli                                                -- categorical.rs:56-177 over libm::log1p;
        no reference vector exists; compared with the oracle's separate restatement (tests/test_model_families_cpu.py)
  QuantizedLaplace, QuantizedCauchy, Binomial                                         -- the CDFs live in the un-vendored
        `probability` crate and no reference vector pins them; device tables are compared bit for bit with the oracle's
        separate restatement (tests/test_gpu_model_families.py), which DEFINES the last ulp
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


_warned = set()


def _warn_once(key, text):
    """the reference prints its deprecation warnings once per process (pybindings/stream/model.rs:505-525, 998-1010)"""
    if key not in _warned:
        _warned.add(key)
        print(text)


def perfect_quantized_cdf(probabilities: np.ndarray, precision: int = PRECISION) -> np.ndarray:
    """`perfectly_quantized_probabilities` + cumulation (src/stream/model/categorical.rs:56-177, contiguous.rs:301-313):
    start from weight 1 + trunc(p * (2^P - n) / sum p) per symbol, hand the remaining weight to the symbols with the
    largest win, then move single units from the cheapest seller to the best buyer while that lowers the cross entropy.
    f64 throughout (f32 inputs are widened: `F: Into<f64>`).  Runs in the library's host code
    (cst_categorical_perfect_cdf, csrc/cst_families.hip) over its own `libm::log1p`."""
    from .. import _native as N
    p = np.asarray(probabilities)
    if p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    if p.ndim != 1:
        raise ValueError("probabilities must be rank 1")
    p = np.ascontiguousarray(p, dtype=np.float64)
    cdf = np.empty(p.shape[0] + 1, dtype=np.uint32)
    rc = N.CST_ERR_MODEL if p.shape[0] < 2 else \
        N.load_library().cst_categorical_perfect_cdf(p.ctypes.data, p.shape[0], int(precision), cdf.ctypes.data)
    if rc == N.CST_ERR_MODEL:
        raise ValueError("Probability distribution not normalizable (the array of probabilities\n"
                         "might be empty, contain negative values or NaNs, or sum to infinity).")
    N.check(rc, "cst_categorical_perfect_cdf")
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
        if self.mean is not None and hi - lo + 1 > 65536:
            # a concrete model is a device-resident table (16-bit symbol indices); with per-symbol (mean, std) arrays the
            # same family works for any support up to 2^24 symbols
            raise ValueError("A concrete QuantizedGaussian is tabulated on the device: its support may hold at most 65536 "
                             "symbols.  Pass `mean` and `std` as per-symbol arrays to encode/decode for wider supports.")
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
    (src/pybindings/stream/model.rs:455-578).  `perfect=False` and `lazy=True` give the "fast" quantisation (their tables
    are identical, tests/python/test_lazy_f32.py); `perfect=True` -- still the reference's default when neither flag is
    given, with the same one-time warning -- the cross-entropy-optimal one (perfect_quantized_cdf)."""
    _n_params = 1

    def __init__(self, probabilities=None, lazy: Optional[bool] = None, perfect: Optional[bool] = None):
        if lazy and perfect:
            raise ValueError("Both arguments `lazy` and `perfect` cannot be set to `True` at the same time.\n"
                             "Lazy categorical entropy models cannot perfectly quantize probabilities.")
        if perfect is None and lazy is None:
            _warn_once("categorical", "WARNING: Neither argument `perfect` nor `lazy` were specified for `Categorical` entropy model.\n"
                       "         In this case, `perfect` currently defaults to `True` for backward compatibility, but\n"
                       "         this default will change to `perfect=False` in constriction version 0.5.\n"
                       "         To suppress this warning, explicitly set:\n"
                       "         - `perfect=False`: recommended for most new use cases; or\n"
                       "         - `perfect=True`: if you need backward compatibility with constriction <= 0.3.5.")
            perfect = True
        self.perfect = bool(perfect)
        self.probabilities = None
        self._dev = None
        if probabilities is not None:
            p = np.asarray(probabilities)
            if p.dtype not in (np.float32, np.float64):
                raise TypeError("probabilities must have dtype float32 or float64")
            if p.ndim != 1:
                raise ValueError("probabilities must be a rank-1 array")
            self.probabilities = np.ascontiguousarray(p)
            self.cdf = self._quantize(self.probabilities)

    def _quantize(self, row):
        return perfect_quantized_cdf(row, PRECISION) if self.perfect else fast_quantized_cdf(row, PRECISION)

    def is_concrete(self):
        return self.probabilities is not None

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(self.cdf, 0, PRECISION)
        return self._dev

    min_symbol = 0

    def family_rows(self, params) -> np.ndarray:
        """One quantised cdf row per symbol for the family form (a rank-2 array of probabilities)."""
        if len(params) != 1:
            raise ValueError("Wrong number of model parameters: Categorical expects one rank-2 array of probabilities.")
        m = np.asarray(params[0])
        if m.ndim != 2:
            raise ValueError("expected a rank-2 array of probabilities (one row per symbol)")
        if m.dtype not in (np.float32, np.float64):
            raise TypeError("probabilities must have dtype float32 or float64")
        return np.stack([self._quantize(row) for row in m]) if len(m) else np.zeros((0, m.shape[1] + 1), np.uint32)

    @staticmethod
    def cdf_rows(prob_matrix) -> np.ndarray:
        """fast-quantised rows (kept for callers of the round-1 interface)"""
        return Categorical(perfect=False).family_rows((prob_matrix,))


class Bernoulli(Model):
    """constriction.stream.model.Bernoulli(p=None, perfect=None) (src/pybindings/stream/model.rs:968-1055): the categorical
    model over {0, 1} with probabilities [1 - p, p], in f64; `perfect` as for `Categorical` (default True, one-time warning)."""
    _n_params = 1
    min_symbol = 0

    def __init__(self, p=None, perfect: Optional[bool] = None):
        if perfect is None:
            _warn_once("bernoulli", "WARNING: Argument `perfect` was not specified for `Bernoulli` distribution.\n"
                       "         It currently defaults to `perfect=True` for backward compatibility, but this default\n"
                       "         will change to `perfect=False` in constriction version 0.5. To suppress this warning,\n"
                       "         explicitly set `perfect=False` (recommended for most new use cases) or explicitly set\n"
                       "         `perfect=True` (if you need backward compatibility with constriction <= 0.3.5).")
            perfect = True
        self.perfect = bool(perfect)
        self.p = None if p is None else float(p)
        self._dev = None
        if self.p is not None:
            self.cdf = self._row(self.p)

    def _row(self, p):
        try:
            row = np.array([1.0 - p, p], dtype=np.float64)
            return perfect_quantized_cdf(row, PRECISION) if self.perfect else fast_quantized_cdf(row, PRECISION)
        except ValueError:
            raise ValueError("`p` must be >= 0.0 and <= 1.0.") from None

    def is_concrete(self):
        return self.p is not None

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(self.cdf, 0, PRECISION)
        return self._dev

    def family_rows(self, params):
        if len(params) != 1:
            raise ValueError("Wrong number of model parameters: Bernoulli expects one array `p`.")
        ps = _as_float_params(params[0], "p")
        return np.stack([self._row(float(x)) for x in ps]) if len(ps) else np.zeros((0, 3), np.uint32)


class Uniform(Model):
    """constriction.stream.model.Uniform(size=None) (src/pybindings/stream/model.rs:562-600, src/stream/model/uniform.rs):
    every symbol of {0, ..., size-1} has probability floor(2^24 / size), the last one takes the remainder too.  Integer
    arithmetic only.  As a table: size <= 65536 (concrete) / <= 4096 (per-symbol sizes)."""
    _n_params = 1
    min_symbol = 0

    def __init__(self, size=None):
        self.size = None if size is None else int(size)
        self._dev = None
        if self.size is not None:
            self.cdf = self._row(self.size, 65536)

    @staticmethod
    def _row(size, limit, width=None):
        if size < 2:
            raise ValueError("`size` must be at least 2.")          # uniform.rs:114 (assert!)
        if size > (1 << PRECISION):
            raise ValueError("`size` must be smaller than 2**24.")
        if size > limit:
            raise ValueError(f"Uniform models over more than {limit} symbols are not supported by this backend in this form.")
        per_bin = (1 << PRECISION) // size
        width = size if width is None else width
        row = np.full(width + 1, 1 << PRECISION, dtype=np.uint32)       # (padding rows repeat 2^24: never selected)
        row[:size] = np.arange(size, dtype=np.uint64) * per_bin
        return row

    def is_concrete(self):
        return self.size is not None

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(self.cdf, 0, PRECISION)
        return self._dev

    def family_rows(self, params):
        if len(params) != 1:
            raise ValueError("Wrong number of model parameters: Uniform expects one array `size`.")
        sizes = np.asarray(params[0])
        if sizes.dtype != np.int32 or sizes.ndim != 1:
            raise TypeError("`size` must be a rank-1 numpy array with dtype int32")
        if len(sizes) == 0:
            return np.zeros((0, 3), np.uint32)
        width = int(sizes.max())
        return np.stack([self._row(int(z), 4096, width) for z in sizes])


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
        if hi - lo + 1 > 65536:
            raise ValueError("This backend tabulates CustomModel / ScipyModel: the support may hold at most 65536 symbols.")
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


class _LeakyFamily(Model):
    """A continuous two-parameter family under the LeakyQuantizer<f64,i32,u32,24> (quantize.rs:284-308, 525-568), tabulated
    on the device (cst_family_cdf_rows).  Subclasses name the family."""
    _n_params = 2
    _names = ("a", "b")
    _family = None

    def __init__(self, min_symbol_inclusive, max_symbol_inclusive, a=None, b=None):
        lo, hi = int(min_symbol_inclusive), int(max_symbol_inclusive)
        if not hi > lo:
            raise ValueError("The support must contain at least two symbols.")
        if hi - lo + 1 > 65536:
            raise ValueError("This backend tabulates this model family: the support may hold at most 65536 symbols.")
        if (a is None) != (b is None):
            raise ValueError(f"Either none or both of `{self._names[0]}` and `{self._names[1]}` must be specified.")
        self.min_symbol, self.max_symbol = lo, hi
        self.a = None if a is None else float(a)
        self.b = None if b is None else float(b)
        if self.b is not None:
            self._check(np.array([self.a]), np.array([self.b]))
        self._dev = None

    def _check(self, a, b):
        if not np.all(b > 0.0):
            raise ValueError(f"Invalid model parameter: `{self._names[1]}` must be positive.")

    def is_concrete(self):
        return self.a is not None

    def cdf_table(self) -> np.ndarray:
        """the concrete model's cdf[n + 1] (computed on the device)"""
        from .. import batched
        return batched.family_cdf_rows(self._family, self.min_symbol, self.max_symbol, [self.a], [self.b])[0]

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(self.cdf_table(), self.min_symbol, PRECISION)
        return self._dev

    def family_rows(self, params):
        if len(params) != 2:
            raise ValueError(f"Wrong number of model parameters: expected ({self._names[0]}, {self._names[1]}).")
        a, b = _as_float_params(params[0], self._names[0]), _as_float_params(params[1], self._names[1])
        if len(a) != len(b):
            raise ValueError("Model parameters have unequal lengths.")
        n = self.max_symbol - self.min_symbol + 1
        if len(a) == 0:
            return np.zeros((0, n + 1), np.uint32)
        self._check(a, b)
        from .. import batched
        return batched.family_cdf_rows(self._family, self.min_symbol, self.max_symbol, a, b)


class QuantizedLaplace(_LeakyFamily):
    """constriction.stream.model.QuantizedLaplace(min_symbol_inclusive, max_symbol_inclusive, mean=None, scale=None)
    (src/pybindings/stream/model.rs:736-800).  CDF of the `probability` crate's Laplace (no reference vector; pinned to
    the oracle's restatement): x <= mean: exp((x - mean) / scale) / 2, else 1 - exp(-(x - mean) / scale) / 2."""
    _names = ("mean", "scale")
    _family = 1     # CST_FAMILY_LAPLACE

    def __init__(self, min_symbol_inclusive, max_symbol_inclusive, mean=None, scale=None):
        super().__init__(min_symbol_inclusive, max_symbol_inclusive, mean, scale)


class QuantizedCauchy(_LeakyFamily):
    """constriction.stream.model.QuantizedCauchy(min_symbol_inclusive, max_symbol_inclusive, loc=None, scale=None)
    (src/pybindings/stream/model.rs:836-900).  CDF atan((x - loc) / scale) / pi + 1/2 (no reference vector; pinned to the
    oracle's restatement)."""
    _names = ("loc", "scale")
    _family = 2     # CST_FAMILY_CAUCHY

    def __init__(self, min_symbol_inclusive, max_symbol_inclusive, loc=None, scale=None):
        super().__init__(min_symbol_inclusive, max_symbol_inclusive, loc, scale)


class Binomial(Model):
    """constriction.stream.model.Binomial(n=None, p=None) (src/pybindings/stream/model.rs:903-966): LeakyQuantizer over
    {0, ..., n} applied to the Binomial(n, p) CDF, which the `probability` crate evaluates as a regularised incomplete beta
    function (Algorithm AS 63 in the `special` crate); tabulated on the device the same way (cst_family_cdf_rows; no
    reference vector, pinned to the oracle's restatement).  Forms: Binomial(n, p), Binomial(n) with `p` per symbol,
    Binomial() with `n` and `p` per symbol."""
    _n_params = 2
    min_symbol = 0

    def __init__(self, n=None, p=None):
        if n is None and p is not None:
            raise ValueError("Either none or both of `n` and `p` must be specified, or only `n`.")
        self.n = None if n is None else int(n)
        self.p = None if p is None else float(p)
        if self.n is not None and self.n > 65535:
            raise ValueError("This backend tabulates this model family: `n` may be at most 65535.")
        self._dev = None

    @staticmethod
    def _rows(ns, ps):
        """one row per (n, p) pair, padded to the largest n with 2^24"""
        ns, ps = np.asarray(ns, dtype=np.int32), np.asarray(ps, dtype=np.float64)
        if np.any(ns < 1):
            raise ValueError("`n` must be at least 1.")
        if not np.all((ps >= 0.0) & (ps <= 1.0)):
            raise ValueError("`p` must be >= 0.0 and <= 1.0.")
        from .. import batched
        return batched.family_cdf_rows(3, 0, int(ns.max()), ps, None, n_per_row=ns)     # CST_FAMILY_BINOMIAL

    def is_concrete(self):
        return self.n is not None and self.p is not None

    def _device_model(self):
        if not self.is_concrete():
            return super()._device_model()
        if self._dev is None:
            from .. import batched
            self._dev = batched.Model.from_cdf(self._rows([self.n], [self.p])[0], 0, PRECISION)
        return self._dev

    def family_rows(self, params):
        if self.n is None:
            if len(params) != 2:
                raise ValueError("Wrong number of model parameters: Binomial() expects (n, p).")
            ns = np.asarray(params[0])
            if ns.dtype != np.int32 or ns.ndim != 1:
                raise TypeError("`n` must be a rank-1 numpy array with dtype int32")
            ps = _as_float_params(params[1], "p")
        else:
            if len(params) != 1:
                raise ValueError("Wrong number of model parameters: Binomial(n) expects (p,).")
            ps = _as_float_params(params[0], "p")
            ns = np.full(len(ps), self.n, dtype=np.int32)
        if len(ns) != len(ps):
            raise ValueError("Model parameters have unequal lengths.")
        if len(ns) == 0:
            return np.zeros((0, 3), np.uint32)
        if int(ns.max()) > 4096:
            raise ValueError("per-symbol Binomial models are tabulated: `n` may be at most 4096 in this form.")
        return self._rows(ns, ps)
