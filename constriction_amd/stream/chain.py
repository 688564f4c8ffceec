"""Drop-in mirror of `constriction.stream.chain.ChainCoder` (src/pybindings/stream/chain.rs:262-517; the coder itself:
src/stream/chain.rs).  The constructors and terminators move a handful of words and are host code, as in the reference
(ChainCoderHeads::new chain.rs:270-303, from_remainders :430-456, into_remainders :406-423, into_compressed :475-496,
into_binary :516-541); the symbol loops -- model evaluation included -- run on the MI355X through
`cst_chain_{encode,decode}_*_batch` (include/constriction_amd.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as N
from . import _single as S

_W, _S, _P = 32, 64, 24          # DefaultChainCoder = ChainCoder<u32, u64, Vec<u32>, Vec<u32>, 24> (chain.rs:306)
_MASK = (1 << _W) - 1


class _Heads(C.Structure):
    """cst_chain_heads"""
    _fields_ = [("remainders_head", C.c_uint64), ("compressed_head", C.c_uint32), ("reserved", C.c_uint32)]


def _new_heads(source: list, push_one: bool, what: str) -> int:
    """ChainCoderHeads::new (chain.rs:270-303): pops the remainders head off `source`"""
    threshold = 1 << (_S - _W - _P)
    if push_one:
        head = 1
    else:
        if not source or source[-1] == 0:
            raise ValueError(what)
        head = source.pop()
    while head < threshold:
        if not source:
            raise ValueError(what)
        head = (head << _W) | source.pop()
    return head


class ChainCoder:
    def __init__(self, data, is_remainders=False, seal=False):
        data = np.asarray(data)
        if data.dtype != np.uint32 or data.ndim != 1:
            raise TypeError("data must be a rank-1 numpy array with dtype uint32")
        words = data.tolist()
        if is_remainders:
            if seal:
                raise AssertionError("Cannot seal remainders data.")
            msg = "Too little data provided, or provided data ends in zero word and `is_remainders==True`."
            if not words or words[-1] == 0:
                raise ValueError(msg)
            head = words.pop()
            self._rem_head = _new_heads(words, False, msg)
            self._comp_head = head
            self._remainders = np.array(words, dtype=np.uint32)
            self._compressed = np.zeros(0, dtype=np.uint32)
        else:
            msg = "Too little data provided." if seal else \
                "Too little data provided, or provided data ends in zero word and `seal==False`."
            self._rem_head = _new_heads(words, bool(seal), msg)
            self._comp_head = 1
            self._compressed = np.array(words, dtype=np.uint32)
            self._remainders = np.zeros(0, dtype=np.uint32)

    # ------------------------------------------------------------------ terminators
    def get_remainders(self):
        """(compressed, remainders) after decoding: into_remainders (chain.rs:406-423)"""
        rem, head = self._remainders.tolist(), self._rem_head
        while head != 0:
            rem.append(head & _MASK)
            head >>= _W
        rem.append(self._comp_head)
        return self._compressed.copy(), np.array(rem, dtype=np.uint32)

    def get_data(self, unseal=False):
        """(remainders, compressed) after re-encoding: into_compressed (chain.rs:475-496) / into_binary (:516-541)"""
        whole = self._comp_head == 1
        if unseal:
            whole = whole and (self._rem_head.bit_length() - 1) % _W == 0
        if not whole:
            raise AssertionError("Fractional number of words in compressed or remainders data.")
        comp, head = self._compressed.tolist(), self._rem_head
        while head > (1 if unseal else 0):
            comp.append(head & _MASK)
            head >>= _W
        return self._remainders.copy(), np.array(comp, dtype=np.uint32)

    def clone(self):
        c = ChainCoder.__new__(ChainCoder)
        c._rem_head, c._comp_head = self._rem_head, self._comp_head
        c._compressed, c._remainders = self._compressed.copy(), self._remainders.copy()
        return c

    # ------------------------------------------------------------------ coding
    def _call(self, pop: np.ndarray, amt: int, launch):
        """One stream through a chain entry point.  `launch(d_pop, d_n_pop, d_push, push_cap, d_n_push, d_heads, d_status)`
        returns the cst_status.  Returns (words popped, pushed words)."""
        tail = min(len(pop), amt)                               # at most one word per symbol moves either way
        d_pop = S.dev(pop[len(pop) - tail:].view(np.int32)) if tail else torch.zeros(4, dtype=torch.int32, device="cuda")
        d_push = torch.empty(max(amt, 1), dtype=torch.int32, device="cuda")
        # [heads 16 B][n_pop u32][n_push u32][status i32][pad]: one upload, one download
        host = np.zeros(8, dtype=np.uint32)
        host[0], host[1] = self._rem_head & _MASK, self._rem_head >> 32
        host[2] = self._comp_head
        host[4] = tail
        buf = torch.from_numpy(host.view(np.int32)).cuda()
        base = buf.data_ptr()
        st = launch(S.ptr(d_pop), C.c_void_p(base + 16), S.ptr(d_push), max(amt, 1), C.c_void_p(base + 20), C.c_void_p(base),
                    C.c_void_p(base + 24))
        N.check(st, "chain coder")
        h = buf.cpu().numpy().view(np.uint32)
        status = int(h.view(np.int32)[6])
        if status == N.STREAM_OUT_OF_DATA:
            return None, None, status
        S.raise_for_status(status)
        self._rem_head = int(h[0]) | (int(h[1]) << 32)
        self._comp_head = int(h[2])
        left, pushed = int(h[4]), int(h[5])
        return tail - left, d_push[:pushed].cpu().numpy().view(np.uint32), status

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
        d_sym = torch.empty(amt, dtype=torch.int32, device="cuda")
        sp = S.stream_ptr()
        keep = []

        def launch(d_pop, d_n_pop, d_push, cap, d_n_push, d_heads, d_status):
            if kind[0] == "gaussian":
                _, lo, hi, means, stds = kind
                d_mu, d_sd = S.dev(means), S.dev(stds)
                keep.extend((d_mu, d_sd))
                return L.cst_chain_decode_gaussian_batch(S.cfg(), lo, hi, d_pop, None, 0, d_n_pop, S.ptr(d_mu), S.ptr(d_sd), S.ptr(d_sym),
                                                         1, amt, N.LAYOUT_STREAM_MAJOR, d_push, cap, d_n_push, d_heads, d_status, sp)
            if kind[0] == "table":
                rows, lo, stride = kind[1].cdf().astype(np.uint32)[None, :], kind[1].min_symbol, 0
            else:
                rows, lo = kind[1], kind[2]
                stride = rows.shape[1]
            d_rows = S.dev(np.ascontiguousarray(rows).view(np.int32))
            keep.append(d_rows)
            return L.cst_chain_decode_rows_batch(S.cfg(), d_pop, None, 0, d_n_pop, S.ptr(d_rows), stride, rows.shape[1] - 1, lo,
                                                 S.ptr(d_sym), 1, amt, N.LAYOUT_STREAM_MAJOR, d_push, cap, d_n_push, d_heads, d_status, sp)

        consumed, pushed, status = self._call(self._compressed, amt, launch)
        if status == N.STREAM_OUT_OF_DATA:
            # DecoderFrontendError::OutOfCompressedData (chain.rs:854-866); the reference's binding panics on it
            raise AssertionError("Out of compressed data.")
        if consumed:
            self._compressed = self._compressed[: len(self._compressed) - consumed].copy()
        if len(pushed):
            self._remainders = np.concatenate([self._remainders, pushed])
        out = d_sym.cpu().numpy()
        return int(out[0]) if scalar else out

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
        sp = S.stream_ptr()
        keep = []

        def launch(d_pop, d_n_pop, d_push, cap, d_n_push, d_heads, d_status):
            if kind[0] == "gaussian":
                _, lo, hi, means, stds = kind
                d_s, d_mu, d_sd = S.dev(sym), S.dev(means), S.dev(stds)
                keep.extend((d_s, d_mu, d_sd))
                return L.cst_chain_encode_gaussian_batch(S.cfg(), lo, hi, S.ptr(d_s), S.ptr(d_mu), S.ptr(d_sd), 1, n, N.LAYOUT_STREAM_MAJOR,
                                                         d_pop, None, 0, d_n_pop, d_push, cap, d_n_push, d_heads, d_status, sp)
            if kind[0] == "table":
                # (c, p) of every symbol: a gather from the model's cdf, on the device
                d_cdf = S.dev(kind[1].cdf().astype(np.int64))
                idx = S.dev(sym).to(torch.int64) - int(kind[1].min_symbol)
                ok = (idx >= 0) & (idx < d_cdf.numel() - 1)
                safe = torch.where(ok, idx, torch.zeros_like(idx))
                d_left = d_cdf[safe].to(torch.int32)
                d_prob = torch.where(ok, d_cdf[safe + 1] - d_cdf[safe], torch.zeros_like(idx)).to(torch.int32)
                keep.extend((d_left, d_prob))
                return L.cst_chain_encode_cp_batch(S.cfg(), S.ptr(d_left), S.ptr(d_prob), 1, n, N.LAYOUT_STREAM_MAJOR, d_pop, None, 0,
                                                   d_n_pop, d_push, cap, d_n_push, d_heads, d_status, sp)
            else:
                rows = kind[1]
                idx = sym.astype(np.int64) - kind[2]
                ok = (idx >= 0) & (idx < rows.shape[1] - 1)
                safe = np.where(ok, idx, 0)
                ar = np.arange(n)
                left = rows[ar, safe].astype(np.uint32)
                prob = np.where(ok, rows[ar, safe + 1].astype(np.int64) - left.astype(np.int64), 0).astype(np.uint32)
            d_left, d_prob = S.dev(left.view(np.int32)), S.dev(prob.view(np.int32))
            keep.extend((d_left, d_prob))
            return L.cst_chain_encode_cp_batch(S.cfg(), S.ptr(d_left), S.ptr(d_prob), 1, n, N.LAYOUT_STREAM_MAJOR, d_pop, None, 0, d_n_pop,
                                               d_push, cap, d_n_push, d_heads, d_status, sp)

        consumed, pushed, status = self._call(self._remainders, n, launch)
        if status == N.STREAM_OUT_OF_DATA:
            # EncoderFrontendError::OutOfRemainders -> AssertionError (src/pybindings/stream/chain.rs:519-530)
            raise AssertionError("Out of remainders.")
        if consumed:
            self._remainders = self._remainders[: len(self._remainders) - consumed].copy()
        if len(pushed):
            self._compressed = np.concatenate([self._compressed, pushed])
