"""`constriction.stream.stack.AnsCoder`, computed on the MI355X.

Mirror of src/pybindings/stream/stack.rs (same method names, argument meaning, return dtypes and error types).
The coder object keeps the reference's representation -- `bulk` words plus a 64-bit `state`
(src/stream/stack.rs:119-133) -- on the host; every encode/decode call ships the symbols (and the tail of the
bulk) to the GPU and runs the n_streams == 1 case of the batched kernels with CST_FLAG_RAW_STATE.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as N
from . import _single as S
from . import model as M

_W, _S = 32, 64


class AnsCoder:
    def __init__(self, compressed=None, seal=False):
        # src/pybindings/stream/stack.rs:217-241
        if compressed is None and seal:
            raise ValueError("Need compressed data to seal.")
        self._bulk = np.zeros(0, dtype=np.uint32)
        self._state = 0
        if compressed is not None:
            words = np.asarray(compressed)
            if words.dtype != np.uint32 or words.ndim != 1:
                raise TypeError("compressed must be a rank-1 numpy array with dtype uint32")
            bulk = [int(w) for w in words]
            if seal:
                # from_binary (src/stream/stack.rs:341-360)
                state = 1
                while state < (1 << (_S - _W)) and bulk:
                    state = (state << _W) | bulk.pop()
            else:
                # from_compressed + read_initial_state (src/stream/stack.rs:299-318, 440-462)
                state = 0
                if bulk:
                    first = bulk.pop()
                    if first == 0:
                        raise ValueError("Invalid compressed data: ANS compressed data never ends in a zero word.")
                    state = first
                    while bulk:
                        state = (state << _W) | bulk.pop()
                        if state >= (1 << (_S - _W)):
                            break
            self._bulk = np.array(bulk, dtype=np.uint32)
            self._state = state

    # ------------------------------------------------------------------ introspection
    def _state_words(self):
        """bit_array_to_chunks_truncated(state).rev(): least significant word first (src/lib.rs:719-731)."""
        out, st = [], self._state
        while st:
            out.append(st & 0xFFFFFFFF)
            st >>= _W
        return out

    def pos(self):
        return (len(self._bulk), self._state)          # src/stream/stack.rs:1130-1139

    def seek(self, position, state):
        position = int(position)
        if position > len(self._bulk):
            raise ValueError("Tried to seek past end of stream. Note: in an ANS coder,\n"
                             "both decoding and seeking *consume* compressed data. The Python API of\n"
                             "`constriction`'s ANS coder currently does not support seeking backward.")
        self._bulk = self._bulk[:position].copy()      # Vec::seek = truncate (src/backends.rs:537-555)
        self._state = int(state)

    def clear(self):
        self._bulk = np.zeros(0, dtype=np.uint32)
        self._state = 0

    def num_words(self):
        return len(self._bulk) + len(self._state_words())

    def num_bits(self):
        return _W * self.num_words()

    def num_valid_bits(self):                          # src/stream/stack.rs:623-630
        return _W * len(self._bulk) + max(self._state.bit_length(), 1) - 1

    def is_empty(self):
        return len(self._bulk) == 0 and self._state == 0

    def get_compressed(self, unseal=False):
        if unseal:
            # get_binary = CoderGuard<SEALED = true> (src/stream/stack.rs:548-555, 1164-1177): the most significant
            # word of the state must be exactly 1 (the seal); the remaining state words follow the bulk
            tail = self._state_words()
            if not tail or tail[-1] != 1:
                raise AssertionError("Cannot unseal compressed data because it doesn't fit into integer number of words. "
                                     "Did you create the encoder with `seal=True` and restore its original state?")
            return np.concatenate([self._bulk, np.array(tail[:-1], dtype=np.uint32)]).astype(np.uint32)
        return np.concatenate([self._bulk, np.array(self._state_words(), dtype=np.uint32)]).astype(np.uint32)

    def clone(self):
        c = AnsCoder()
        c._bulk, c._state = self._bulk.copy(), self._state
        return c

    # ------------------------------------------------------------------ coding
    def encode_reverse(self, symbols, model, *optional_model_params):
        sym, is_scalar = S.symbols_array(symbols)
        if is_scalar and optional_model_params:
            raise ValueError("To encode a single symbol, use a concrete model, i.e., pass the\n"
                             "model parameters directly to the constructor of the model and not to the\n"
                             "`encode` method of the entropy coder.")
        kind = S.model_args(model, optional_model_params)
        n = len(sym)
        if kind[0] == "gaussian" and len(kind[3]) != n or kind[0] == "rows" and len(kind[1]) != n:
            raise ValueError("`symbols` argument has wrong length.")
        if n == 0:
            return
        L = N.lib()
        cap = L.cst_ans_max_words(n, S.cfg())
        d_words = torch.empty(cap, dtype=torch.int32, device="cuda")
        sc = S.Scalars(self._state)
        d_n, d_state, d_status = sc.p(sc.N), sc.p(sc.STATE), sc.p(sc.STATUS)
        sp = S.stream_ptr()
        if kind[0] == "table":
            d_sym = S.dev(sym)
            st = L.cst_ans_encode_batch(kind[1]._h, S.cfg(), S.ptr(d_sym), 1, n, N.LAYOUT_STREAM_MAJOR, S.ptr(d_words), cap,
                                        d_n, d_state, d_status, N.FLAG_RAW_STATE, sp)
        elif kind[0] == "gaussian":
            _, lo, hi, means, stds = kind
            d_sym, d_mu, d_sd = S.dev(sym), S.dev(means), S.dev(stds)
            st = L.cst_ans_encode_gaussian_batch(S.cfg(), lo, hi, S.ptr(d_sym), S.ptr(d_mu), S.ptr(d_sd), 1, n,
                                                 N.LAYOUT_STREAM_MAJOR, S.ptr(d_words), cap, d_n, d_state,
                                                 d_status, N.FLAG_RAW_STATE, sp)
        else:
            rows = kind[1]
            idx = sym.astype(np.int64) - kind[2]
            ok = (idx >= 0) & (idx < rows.shape[1] - 1)
            safe = np.where(ok, idx, 0)
            ar = np.arange(n)
            left = rows[ar, safe].astype(np.uint32)
            prob = np.where(ok, rows[ar, safe + 1].astype(np.int64) - left.astype(np.int64), 0).astype(np.uint32)
            d_left, d_prob = S.dev(left.view(np.int32)), S.dev(prob.view(np.int32))
            st = L.cst_ans_encode_cp_batch(S.cfg(), S.ptr(d_left), S.ptr(d_prob), 1, n, N.LAYOUT_STREAM_MAJOR, S.ptr(d_words),
                                           cap, d_n, d_state, d_status, N.FLAG_RAW_STATE, sp)
        N.check(st, "ans encode")
        state, k, _, status = sc.read()
        S.raise_for_status(status)
        if k:
            self._bulk = np.concatenate([self._bulk, d_words[:k].cpu().numpy().view(np.uint32)])
        self._state = state

    def decode(self, model, *optional_amt_or_model_params):
        params = optional_amt_or_model_params
        scalar = False
        if len(params) == 0:
            scalar, amt, params = True, 1, ()
            kind = S.model_args(model, ())
        elif len(params) == 1 and isinstance(params[0], (int, np.integer)) and not isinstance(params[0], bool):
            amt = int(params[0])
            kind = S.model_args(model, ())
        else:
            kind = S.model_args(model, params)
            amt = len(kind[3]) if kind[0] == "gaussian" else len(kind[1])
        if amt == 0:
            return np.zeros(0, dtype=np.int32)
        L = N.lib()
        # at most one word is consumed per symbol (src/stream/stack.rs:1089-1097)
        tail = min(len(self._bulk), amt)
        words = self._bulk[len(self._bulk) - tail:]
        d_words = S.dev(words.view(np.int32)) if tail else torch.zeros(4, dtype=torch.int32, device="cuda")
        sc = S.Scalars(self._state, tail)
        d_n, d_n_out, d_state, d_status = sc.p(sc.N), sc.p(sc.N_OUT), sc.p(sc.STATE), sc.p(sc.STATUS)
        d_sym = torch.empty(amt, dtype=torch.int32, device="cuda")
        sp = S.stream_ptr()
        if kind[0] == "table":
            st = L.cst_ans_decode_batch(kind[1]._h, S.cfg(), S.ptr(d_words), None, max(tail, 1), d_words.numel(), d_n, S.ptr(d_sym), 1, amt,
                                        N.LAYOUT_STREAM_MAJOR, d_state, d_n_out, d_status,
                                        N.FLAG_RAW_STATE, sp)
        elif kind[0] == "gaussian":
            _, lo, hi, means, stds = kind
            d_mu, d_sd = S.dev(means), S.dev(stds)
            st = L.cst_ans_decode_gaussian_batch(S.cfg(), lo, hi, S.ptr(d_words), None, max(tail, 1), d_words.numel(), d_n, S.ptr(d_mu),
                                                 S.ptr(d_sd), S.ptr(d_sym), 1, amt, N.LAYOUT_STREAM_MAJOR, d_state,
                                                 d_n_out, d_status, N.FLAG_RAW_STATE, sp)
        else:
            rows = kind[1]
            d_rows = S.dev(rows.view(np.int32))
            st = L.cst_ans_decode_rows_batch(S.cfg(), S.ptr(d_words), None, max(tail, 1), d_words.numel(), d_n, S.ptr(d_rows),
                                             rows.shape[1] - 1, kind[2], S.ptr(d_sym), 1, amt, N.LAYOUT_STREAM_MAJOR,
                                             d_state, d_n_out, d_status, N.FLAG_RAW_STATE, sp)
        N.check(st, "ans decode")
        state, _, n_left, status = sc.read()
        S.raise_for_status(status)
        consumed = tail - n_left
        if consumed:
            self._bulk = self._bulk[: len(self._bulk) - consumed].copy()
        self._state = state
        out = d_sym.cpu().numpy()
        return int(out[0]) if scalar else out
