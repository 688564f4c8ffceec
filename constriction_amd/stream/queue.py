"""`constriction.stream.queue.RangeEncoder` / `RangeDecoder`, computed on the MI355X.

Mirror of src/pybindings/stream/queue.rs.  The coder objects keep the reference's representation on the host:
encoder = bulk words + RangeCoderState{lower, range} + EncoderSituation (src/stream/queue.rs:60-71, 107-142);
decoder = the compressed words + a read position + {lower, range, point} (queue.rs:727-737).  Every
encode/decode call is the n_streams == 1 case of the batched kernels with CST_FLAG_RAW_STATE.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as N
from . import _single as S

_W, _S = 32, 64
_MAX = (1 << _S) - 1
_MASK = _MAX


class _Scalars:
    """cst_range_state + word count + stream status of one call in ONE 64-byte device buffer (one upload, one download)"""
    N_OFF, STATUS_OFF = 48, 52

    def __init__(self, lower, rng, point=0, inv_n=0, inv_first=0, position=0, n=0):
        rs = N.RangeState(lower, rng, point, inv_n, inv_first, position)
        buf = np.zeros(64, dtype=np.uint8)
        raw = np.frombuffer(bytes(rs), dtype=np.uint8)
        assert len(raw) <= self.N_OFF
        buf[: len(raw)] = raw
        buf[self.N_OFF: self.N_OFF + 4] = np.frombuffer(np.uint32(n).tobytes(), dtype=np.uint8)
        self.t = torch.from_numpy(buf).cuda()
        self.state, self.n, self.status = (C.c_void_p(self.t.data_ptr() + off) for off in (0, self.N_OFF, self.STATUS_OFF))

    def read(self):
        """(cst_range_state, n_words, status) after the calls on the current stream (the copy synchronises)"""
        h = self.t.cpu().numpy()
        return (N.RangeState.from_buffer_copy(h[: self.N_OFF].tobytes()), int(h[self.N_OFF: self.N_OFF + 4].view(np.uint32)[0]),
                int(h[self.STATUS_OFF: self.STATUS_OFF + 4].view(np.int32)[0]))


class RangeEncoder:
    def __init__(self):
        self.clear()

    def clear(self):
        self._bulk = np.zeros(0, dtype=np.uint32)
        self._lower, self._range = 0, _MAX          # RangeCoderState::default (queue.rs:96-104)
        self._inv_n, self._inv_first = 0, 0         # EncoderSituation::Normal

    def _seal_words(self):
        """iter_seal / seal_words (src/stream/queue.rs:458-523)."""
        if self._range == _MAX:
            return []
        out = []
        point = (self._lower + ((1 << (_S - _W)) - 1)) & _MASK
        if self._inv_n:
            if point >= self._lower:
                first, cons = self._inv_first, 0xFFFFFFFF
            else:
                first, cons = (self._inv_first + 1) & 0xFFFFFFFF, 0
            out.append(first)
            out.extend([cons] * (self._inv_n - 1))
        point_word = point >> (_S - _W)
        upper_word = ((self._lower + self._range) & _MASK) >> (_S - _W)
        out.append(point_word)
        if upper_word == point_word:
            out.append(0)
        return out

    def pos(self):
        return (len(self._bulk) + self._inv_n, (self._lower, self._range))    # queue.rs:188-195

    def num_words(self):
        return len(self._bulk) + len(self._seal_words())

    def num_bits(self):
        return _W * self.num_words()

    def is_empty(self):
        return self._range == _MAX and len(self._bulk) == 0

    def get_compressed(self):
        return np.concatenate([self._bulk, np.array(self._seal_words(), dtype=np.uint32)]).astype(np.uint32)

    def get_decoder(self):
        return RangeDecoder(self.get_compressed())

    def clone(self):
        c = RangeEncoder()
        c._bulk = self._bulk.copy()
        c._lower, c._range, c._inv_n, c._inv_first = self._lower, self._range, self._inv_n, self._inv_first
        return c

    def encode(self, symbols, model, *optional_model_params):
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
        cap = L.cst_range_max_words(n, S.cfg()) + self._inv_n + 2
        d_words = torch.empty(cap, dtype=torch.int32, device="cuda")
        sc = _Scalars(self._lower, self._range, 0, self._inv_n, self._inv_first, 0)
        d_n, d_rs, d_status = sc.n, sc.state, sc.status
        sp = S.stream_ptr()
        if kind[0] == "table":
            d_sym = S.dev(sym)
            st = L.cst_range_encode_batch(kind[1]._h, S.cfg(), S.ptr(d_sym), 1, n, N.LAYOUT_STREAM_MAJOR, S.ptr(d_words), cap,
                                          d_n, d_rs, d_status, N.FLAG_RAW_STATE, sp)
        elif kind[0] == "gaussian":
            _, lo, hi, means, stds = kind
            d_sym, d_mu, d_sd = S.dev(sym), S.dev(means), S.dev(stds)
            st = L.cst_range_encode_gaussian_batch(S.cfg(), lo, hi, S.ptr(d_sym), S.ptr(d_mu), S.ptr(d_sd), 1, n,
                                                   N.LAYOUT_STREAM_MAJOR, S.ptr(d_words), cap, d_n, d_rs,
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
            st = L.cst_range_encode_cp_batch(S.cfg(), S.ptr(d_left), S.ptr(d_prob), 1, n, N.LAYOUT_STREAM_MAJOR, S.ptr(d_words),
                                             cap, d_n, d_rs, d_status, N.FLAG_RAW_STATE, sp)
        N.check(st, "range encode")
        rs, k, status = sc.read()
        S.raise_for_status(status)
        if k:
            self._bulk = np.concatenate([self._bulk, d_words[:k].cpu().numpy().view(np.uint32)])
        self._lower, self._range, self._inv_n, self._inv_first = rs.lower, rs.range, rs.inverted_n, rs.inverted_first


class RangeDecoder:
    def __init__(self, compressed):
        words = np.asarray(compressed)
        if words.dtype != np.uint32 or words.ndim != 1:
            raise TypeError("compressed must be a rank-1 numpy array with dtype uint32")
        self._words = np.ascontiguousarray(words).copy()
        self._lower, self._range = 0, _MAX
        self._pos = 0
        self._point = self._read_point()

    def _read_point(self):
        """RangeDecoder::read_point (src/stream/queue.rs:847-868)."""
        point, num_read = 0, 0
        while self._pos < len(self._words):
            point = ((point << _W) | int(self._words[self._pos])) & _MASK
            self._pos += 1
            num_read += 1
            if num_read == _S // _W:
                break
        if 0 < num_read < _S // _W:
            point = (point << (_S - num_read * _W)) & _MASK
        return point

    def maybe_exhausted(self):
        """src/stream/queue.rs:872-883."""
        max_difference = (((1 << (_S - _W)) << 1) - 1) & _MASK
        return self._pos >= len(self._words) and (self._range == _MAX or ((self._point - self._lower) & _MASK) < max_difference)

    def pos(self):
        # position of the next word to read, minus the words already folded into `point` (queue.rs:897-915)
        return (max(self._pos - _S // _W, 0) if self._pos >= _S // _W else 0, (self._lower, self._range))

    def seek(self, position, state):
        position = int(position)
        if position > len(self._words):
            raise ValueError("Tried to seek past end of stream.")
        self._pos = position
        self._point = self._read_point()
        self._lower, self._range = int(state[0]), int(state[1])

    def clone(self):
        c = RangeDecoder(self._words)
        c._lower, c._range, c._pos, c._point = self._lower, self._range, self._pos, self._point
        return c

    def decode(self, model, *optional_amt_or_model_params):
        params = optional_amt_or_model_params
        scalar = False
        if len(params) == 0:
            scalar, amt = True, 1
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
        # at most one word is consumed per symbol (src/stream/queue.rs:1010-1027)
        window = self._words[self._pos: self._pos + amt]
        nwin = len(window)
        d_words = S.dev(window.view(np.int32)) if nwin else torch.zeros(4, dtype=torch.int32, device="cuda")
        sc = _Scalars(self._lower, self._range, self._point, 0, 0, 0, nwin)
        d_n, d_rs, d_status = sc.n, sc.state, sc.status
        d_sym = torch.empty(amt, dtype=torch.int32, device="cuda")
        sp = S.stream_ptr()
        if kind[0] == "table":
            st = L.cst_range_decode_batch(kind[1]._h, S.cfg(), S.ptr(d_words), None, max(nwin, 1), d_words.numel(), d_n, S.ptr(d_sym), 1, amt,
                                          N.LAYOUT_STREAM_MAJOR, d_rs, d_status, N.FLAG_RAW_STATE, sp)
        elif kind[0] == "gaussian":
            _, lo, hi, means, stds = kind
            d_mu, d_sd = S.dev(means), S.dev(stds)
            st = L.cst_range_decode_gaussian_batch(S.cfg(), lo, hi, S.ptr(d_words), None, max(nwin, 1), d_words.numel(), d_n, S.ptr(d_mu),
                                                   S.ptr(d_sd), S.ptr(d_sym), 1, amt, N.LAYOUT_STREAM_MAJOR, d_rs,
                                                   d_status, N.FLAG_RAW_STATE, sp)
        else:
            rows = kind[1]
            d_rows = S.dev(rows.view(np.int32))
            st = L.cst_range_decode_rows_batch(S.cfg(), S.ptr(d_words), None, max(nwin, 1), d_words.numel(), d_n, S.ptr(d_rows),
                                               rows.shape[1] - 1, kind[2], S.ptr(d_sym), 1, amt, N.LAYOUT_STREAM_MAJOR, d_rs,
                                               d_status, N.FLAG_RAW_STATE, sp)
        N.check(st, "range decode")
        rs, _, status = sc.read()
        S.raise_for_status(status)
        self._lower, self._range, self._point = rs.lower, rs.range, rs.point
        self._pos += int(rs.position)
        out = d_sym.cpu().numpy()
        return int(out[0]) if scalar else out
