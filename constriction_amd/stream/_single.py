"""Plumbing shared by the single-coder drop-in classes: every call is the n_streams == 1 case of the batched
C ABI (include/constriction_amd.h) with CST_FLAG_RAW_STATE, so that one coder object can be continued across
calls exactly like the reference's `AnsCoder` / `RangeEncoder` objects.  Host <-> device copies go through
torch tensors (device memory management only); all coding happens in the HIP library."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as N
from . import model as M

CFG = (32, 64, M.PRECISION)   # DefaultAnsCoder / DefaultRangeEncoder with PRECISION = 24 (internals.rs:26-39)


def cfg():
    return N.CoderConfig(*CFG)


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def symbols_array(symbols):
    """Returns (int32 array, is_scalar)."""
    if isinstance(symbols, (int, np.integer)):
        return np.array([int(symbols)], dtype=np.int32), True
    a = np.asarray(symbols)
    if a.dtype != np.int32:
        raise TypeError("symbols must be an integer or a numpy array with dtype int32")
    if a.ndim != 1:
        raise ValueError("symbols must be a rank-1 array")
    return np.ascontiguousarray(a), False


def raise_for_status(status: int):
    if status == N.STREAM_OK:
        return
    if status == N.STREAM_IMPOSSIBLE_SYMBOL:
        # src/pybindings/stream/mod.rs:82-89
        raise KeyError("Tried to encode symbol that has zero probability under the used entropy model.")
    if status == N.STREAM_CAPACITY:
        raise MemoryError("internal error: output slab too small")
    if status == N.STREAM_INVALID_DATA:
        # src/pybindings/stream/queue.rs:678-683
        raise AssertionError("Tried to decode from compressed data that is invalid for the employed entropy model.")
    raise RuntimeError(f"stream status {status}")


def model_args(model, params, n_expected=None):
    """Classifies an (model, *params) call.  Returns one of
       ("table", device_model)                         concrete model, iid symbols
       ("gaussian", lo, hi, means, stds)               QuantizedGaussian family with per-symbol parameters
       ("rows", cdf_rows, min_symbol)                  one tabulated cdf row per symbol position (Categorical family with
                                                       a probability matrix; CustomModel / ScipyModel with parameters)
    """
    if not isinstance(model, M.Model):
        raise TypeError("model must be a constriction_amd.stream.model.Model")
    if isinstance(model, M.CustomModel) and len(params) > 0:
        return ("rows", model.cdf_rows(params), model.min_symbol)
    if len(params) == 0:
        if not model.is_concrete():
            raise ValueError("This model family needs its parameters to be passed to `encode`/`decode`.")
        return ("table", model._device_model())
    if model.is_concrete():
        raise ValueError("Model parameters were specified but the model is already fully parameterized.")
    if isinstance(model, M.QuantizedGaussian):
        if len(params) != 2:
            raise ValueError("Wrong number of model parameters: QuantizedGaussian expects (means, stds).")
        means, stds = M._as_float_params(params[0], "means"), M._as_float_params(params[1], "stds")
        if len(means) != len(stds):
            raise ValueError("Model parameters have unequal lengths.")
        return ("gaussian", model.min_symbol, model.max_symbol, means, stds)
    if hasattr(model, "family_rows"):      # tabulated families: one cdf row per symbol position
        return ("rows", model.family_rows(params), model.min_symbol)
    raise TypeError("unsupported model family")


class Scalars:
    """The per-call scalars of one coder in ONE 32-byte device buffer -- coder state (u64), word count in / out (u32 each),
    stream status (i32) -- so that a call costs one small upload and one small download instead of one per value."""
    STATE, N, N_OUT, STATUS = 0, 8, 12, 16          # byte offsets

    def __init__(self, state: int = 0, n: int = 0):
        host = np.zeros(4, dtype=np.int64)
        host[0] = np.uint64(state).astype(np.int64)
        host[1] = int(n)                             # n in the low word, n_out = 0 in the high one
        self.t = torch.from_numpy(host).cuda()

    def p(self, offset):
        return C.c_void_p(self.t.data_ptr() + offset)

    def read(self):
        """(state, n, n_out, status) after the calls on the current stream (the copy synchronises)"""
        h = self.t.cpu().numpy()
        w = h.view(np.uint32)
        return int(h.view(np.uint64)[0]), int(w[2]), int(w[3]), int(h.view(np.int32)[4])
