"""Batched, device-resident front end: thousands of independent coders per call.

This is the extension the reference has no counterpart for (it codes one stream per `AnsCoder`
object, src/stream/stack.rs); every stream's compressed words are bit-identical to what one
reference coder would produce for that stream alone.  PyTorch is used for device memory and
streams only; all computation happens in the HIP library behind include/constriction_amd.h.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _native as N

PRESETS = {"default": (32, 64, 24), "lookup": (32, 64, 12), "small": (16, 32, 12)}


_SYMBOL_BYTES = {torch.int8: 1, torch.int16: 2, torch.int32: 4}


def _cfg(W, S, P):
    return N.CoderConfig(W, S, P)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_cuda(t: torch.Tensor, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name} must live in device memory (HBM)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}")
    return t.contiguous()


class Model:
    """Device image of an entropy model with contiguous support (include/constriction_amd.h, cst_model)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.load_library().cst_model_destroy(h)
            except Exception:
                pass

    @classmethod
    def from_cdf(cls, cdf, min_symbol: int, precision: int) -> "Model":
        """Any tabulated model: cdf[n+1] with cdf[0]=0 < ... < cdf[n]=2^P."""
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        h = C.c_void_p()
        N.check(N.lib().cst_model_create_table(precision, int(min_symbol), len(cdf) - 1, cdf.ctypes.data, C.byref(h)),
                "cst_model_create_table")
        return cls(h)

    @classmethod
    def from_cdf_noncontiguous(cls, symbols, cdf, precision: int) -> "Model":
        """A tabulated model over arbitrary distinct int32 symbols (symbols[i] has left cumulative cdf[i]):
        NonContiguousCategorical{Encoder,Decoder}Model / NonContiguousLookupDecoderModel of the reference."""
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        symbols = np.ascontiguousarray(symbols, dtype=np.int32)
        if len(symbols) != len(cdf) - 1:
            raise ValueError("one symbol per table entry")
        h = C.c_void_p()
        N.check(N.lib().cst_model_create_table_noncontiguous(precision, len(symbols), symbols.ctypes.data, cdf.ctypes.data, C.byref(h)),
                "cst_model_create_table_noncontiguous")
        m = cls(h)
        m.noncontiguous = True
        return m

    def symbols_to_indices(self, symbols: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        symbols = _require_cuda(symbols, torch.int32, "symbols")
        out = torch.empty_like(symbols) if out is None else out
        N.check(N.lib().cst_symbols_to_indices(self._h, _ptr(symbols), symbols.numel(), _ptr(out), _stream_ptr()), "cst_symbols_to_indices")
        return out

    def indices_to_symbols(self, indices: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = torch.empty_like(indices) if out is None else out
        N.check(N.lib().cst_indices_to_symbols(self._h, _ptr(indices), indices.numel(), _ptr(out), _stream_ptr()), "cst_indices_to_symbols")
        return out

    noncontiguous = False

    @classmethod
    def quantized_gaussian(cls, min_symbol: int, max_symbol: int, mean: float, std: float, precision: int) -> "Model":
        """LeakyQuantizer(min..=max) x Gaussian(mean, std), tabulated on the GPU in bit-exact f64."""
        h = C.c_void_p()
        N.check(N.lib().cst_model_create_gaussian(precision, int(min_symbol), int(max_symbol), float(mean), float(std),
                                                  _stream_ptr(), C.byref(h)), "cst_model_create_gaussian")
        return cls(h)

    @classmethod
    def quantized_gaussian_per_stream(cls, min_symbol, max_symbol, means: torch.Tensor, stds: torch.Tensor,
                                      precision: int) -> "Model":
        means = _require_cuda(means, torch.float64, "means")
        stds = _require_cuda(stds, torch.float64, "stds")
        if means.shape != stds.shape or means.dim() != 1:
            raise ValueError("means and stds must be 1-d and of equal length (one entry per stream)")
        h = C.c_void_p()
        N.check(N.lib().cst_model_create_gaussian_per_stream(precision, int(min_symbol), int(max_symbol), _ptr(means),
                                                             _ptr(stds), means.numel(), _stream_ptr(), C.byref(h)),
                "cst_model_create_gaussian_per_stream")
        return cls(h)

    @property
    def precision(self):
        return N.load_library().cst_model_precision(self._h)

    @property
    def min_symbol(self):
        return N.load_library().cst_model_min_symbol(self._h)

    @property
    def n_symbols(self):
        return N.load_library().cst_model_n_symbols(self._h)

    @property
    def n_tables(self):
        return N.load_library().cst_model_n_tables(self._h)

    def cdfs_device(self, first: int = 0, count: Optional[int] = None) -> torch.Tensor:
        """The cdfs of tables [first, first + count) as an int32-typed tensor [count, n_symbols + 1] in HBM (values < 2^31
        for every precision <= 24 ... 30), copied on the current stream."""
        count = self.n_tables - first if count is None else count
        out = torch.empty((count, self.n_symbols + 1), dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        N.check(N.lib().cst_model_copy_cdfs(self._h, int(first), int(count), _ptr(out), _stream_ptr()), "cst_model_copy_cdfs")
        return out

    def cdf(self, index: int = 0) -> np.ndarray:
        out = np.zeros(self.n_symbols + 1, dtype=np.uint32)
        N.check(N.lib().cst_model_get_cdf(self._h, index, out.ctypes.data, _stream_ptr()), "cst_model_get_cdf")
        return out


def family_cdf_rows(family: int, min_symbol: int, max_symbol: int, a, b=None, n_per_row=None, precision: int = 24,
                    to_numpy: bool = True):
    """LeakyQuantizer(min..=max) x Laplace / Cauchy / Binomial tabulated on the GPU, one cdf row per parameter pair
    (cst_family_cdf_rows; family = 1 Laplace(mean, scale), 2 Cauchy(loc, scale), 3 Binomial(p) over 0..=max or, with
    `n_per_row`, over 0..=n_per_row[r]).  Returns uint32 rows [n_rows, max - min + 2] (numpy, or a torch.int32 view of
    them in HBM with to_numpy=False); raises ValueError where the reference panics (a probability of zero)."""
    dev = torch.device("cuda", torch.cuda.current_device())

    def dev_f64(x):
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            return _require_cuda(x, torch.float64, "model parameter")
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)

    da, db = dev_f64(a), dev_f64(b)
    dn = None if n_per_row is None else torch.from_numpy(np.ascontiguousarray(n_per_row, dtype=np.int32)).to(dev)
    n_rows = da.numel()
    width = int(max_symbol) - int(min_symbol) + 2
    rows = torch.empty((n_rows, width), dtype=torch.int32, device=dev)
    bad = torch.empty(n_rows, dtype=torch.int32, device=dev)
    N.check(N.lib().cst_family_cdf_rows(int(family), int(precision), int(min_symbol), int(max_symbol), _ptr(da), _ptr(db), _ptr(dn),
                                        n_rows, _ptr(rows), _ptr(bad), _stream_ptr()), "cst_family_cdf_rows")
    if n_rows and bool(bad.any().item()):
        raise ValueError("Invalid model: a symbol of the support gets probability zero under the leaky quantizer "
                         "(quantize.rs:560-566).")
    return rows.cpu().numpy().view(np.uint32) if to_numpy else rows


def _to_indices(model: "Model", symbols: torch.Tensor) -> torch.Tensor:
    """Every table-model entry point codes INDICES into the model's alphabet; for a non-contiguous alphabet
    (Model.from_cdf_noncontiguous) the symbols are translated first (a symbol outside the alphabet becomes index n, which the
    coders report as CST_STREAM_IMPOSSIBLE_SYMBOL), for a contiguous one the kernels subtract min_symbol themselves."""
    return model.symbols_to_indices(symbols) if model.noncontiguous else symbols


def _to_symbols(model: "Model", decoded: torch.Tensor) -> torch.Tensor:
    if model.noncontiguous:
        model.indices_to_symbols(decoded, out=decoded)
    return decoded


@dataclass
class EncodedBatch:
    """Per-stream compressed words in fixed-stride slabs (stream s: words[s, :n_words[s]]).

    jump: the jump points the encode call noted for its own words (Checkpoints / RangeCheckpoints, `jump_points=` of the encode
    calls; None = none) -- side information the matching decode call uses to decode every stream on several lanes.  They describe
    the words the encoder wrote: replacing `words` or `n_words` drops them, and whoever edits those tensors IN PLACE sets
    `jump = None` (the plain decoders read the words alone)."""
    words: torch.Tensor      # uint32 as int32 storage? -> torch.int32 view of uint32 words [n_streams, stride]
    n_words: torch.Tensor    # int32 [n_streams] (values are counts)
    status: torch.Tensor     # int32 [n_streams]
    config: tuple
    jump: Optional[object] = None

    def __setattr__(self, name, value):
        if name in ("words", "n_words") and "jump" in self.__dict__:
            object.__setattr__(self, "jump", None)
        object.__setattr__(self, name, value)

    @property
    def stride(self):
        return self.words.shape[1]

    def total_words(self) -> int:
        return int(self.n_words.to(torch.int64).sum().item())

    @property
    def packed16(self) -> bool:
        """(16,32) words two per 32-bit slot, as the reference's Vec<u16> (CST_FLAG_PACKED_W16): `words` is an int16 tensor and
        every count, stride and offset is in 16-bit words"""
        return self.words.dtype == torch.int16

    def stream(self, s: int) -> np.ndarray:
        n = int(self.n_words[s].item())
        return self.words[s, :n].cpu().numpy().view(np.uint16 if self.packed16 else np.uint32)

    def to_numpy(self):
        return (self.words.cpu().numpy().view(np.uint16 if self.packed16 else np.uint32), self.n_words.cpu().numpy().view(np.uint32),
                self.status.cpu().numpy())


# Provenance of compressed words (round 5).  The (32,64), P <= 12 decoder has two ways of reading its words: 16-byte chunks per
# lane (fastest on words that sit in the GPU's caches: what an encode call on the same HIP stream has just left there) and
# whole 64-byte segments by lane quads (CST_FLAG_COLD_WORDS, cst_ans_dq.hip: 11 - 17 % faster on words that come from HBM -- by
# DMA from the host, from a peer, from a file, or simply written long ago).  The C ABI takes a flag; this layer KNOWS where its
# words come from: ans_encode stamps the EncodedBatch it fills, any later encode on that stream takes the stamp's currency away,
# and words that arrive as plain tensors (packed + offsets, scatter results, container files) never had one.  Batches whose words
# alone exceed the chip's 256-MiB memory-side cache are cold whatever their history.
_INFINITY_CACHE_BYTES = 256 << 20
_launch_serial = [0]
_last_encode = {}        # HIP stream -> serial of the last encode launched on it


def _stamp_fresh(out) -> None:
    _launch_serial[0] += 1
    out._fresh = (torch.cuda.current_stream().cuda_stream, _launch_serial[0])
    _last_encode[out._fresh[0]] = _launch_serial[0]


def _words_are_cold(encoded) -> bool:
    fresh = getattr(encoded, "_fresh", None)
    if fresh is None or fresh[0] != torch.cuda.current_stream().cuda_stream or _last_encode.get(fresh[0]) != fresh[1]:
        return True
    return int(encoded.n_words.numel()) * int(encoded.words.shape[1]) * encoded.words.element_size() > 3 * _INFINITY_CACHE_BYTES   # (slabs are ~1/3 used)


def last_kernel() -> str:
    """the kernel family this thread's last batched coder call launched (cst_last_kernel_name)"""
    return N.lib().cst_last_kernel_name().decode()


def max_words(n_per_stream: int, config=(32, 64, 12)) -> int:
    return N.load_library().cst_ans_max_words(n_per_stream, _cfg(*config))


_TUNED_STRIDES = {}


def tuned_stride(symbols: torch.Tensor, model: "Model", config=(32, 64, 12), layout="stream_major", span=640, step=32,
                 coder="ans", report=None) -> int:
    """The slab stride (words per stream, >= max_words) at which THIS batch shape codes fastest on this device, by measurement.

    How far apart the slabs lie is the caller's choice (`stride_words` of the C ABI), and the P <= 12 ANS decoder is sensitive to
    it: at 65 536 x 4096 it takes 0.25 ms at some strides and 0.35 ms at others, reproducibly, with nothing in the stride's
    residues to go by ("Slab stride": the vector-memory unit takes longer over a load whose 64 lanes
    address 64 lines `stride` apart).  So: encode and decode the given batch at every multiple of `step` words in
    [max_words, max_words + span] (one encode, three decodes each, HIP events), keep the stride with the smallest
    encode + decode time, remember it per (device, shape, preset, layout).  Costs about 30 ms and a few hundred MB of scratch
    the first time; small batches (< 2^26 symbols) get max_words at once.  coder="range": the same for range_encode /
    range_decode.  report: a list that receives (stride, encode_ms, decode_ms) per candidate.  With the environment variable
    CST_STRIDE_CACHE=<file.json> the choice is kept across processes (one file per kind of device)."""
    symbols = _require_cuda(symbols, torch.int32, "symbols")
    n_streams, n_per, _ = _layout_shape(symbols, layout)
    if coder not in ("ans", "range"):
        raise ValueError("coder must be 'ans' or 'range'")
    enc_fn, dec_fn, base = (ans_encode, ans_decode, max_words(n_per, config)) if coder == "ans" else \
                           (range_encode, range_decode, range_max_words(n_per, config))
    # (what picks the kernel: the batch shape, the preset, the layout, the coder, one table or one per stream, the alphabet size)
    kind = ("shared" if model.n_tables == 1 else "per_stream", model.n_symbols)
    key = (symbols.device.index, n_streams, n_per, tuple(config), layout, coder, *kind)
    if key in _TUNED_STRIDES and report is None:
        return _TUNED_STRIDES[key]
    # CST_STRIDE_CACHE=<file.json>: strides measured by earlier processes on this kind of device (a service does not measure at
    # every start; a profiler run of a command that has run before sees the command's own launches only)
    cache_path = os.environ.get("CST_STRIDE_CACHE")
    cache_key = "|".join(map(str, (n_streams, n_per, *config, layout, coder, *kind)))      # (one file per kind of device)
    if cache_path and report is None and os.path.exists(cache_path):
        try:
            with open(cache_path) as f:
                cached = int(json.load(f).get(cache_key, 0))
        except (OSError, ValueError, TypeError, AttributeError):
            cached = 0
        if cached >= base:
            _TUNED_STRIDES[key] = cached
            return cached
    best = base
    if n_streams * n_per >= (1 << 26):
        first = (base + step - 1) // step * step
        cands = [base] + [c for c in range(first, base + span + 1, step) if c != base]
        flat = torch.empty(n_streams * max(cands), dtype=torch.int32, device=symbols.device)
        n_words = torch.empty(n_streams, dtype=torch.int32, device=symbols.device)
        status = torch.empty(n_streams, dtype=torch.int32, device=symbols.device)
        decoded = torch.empty_like(symbols)

        def ms(fn, reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            ev[0].record()
            for k in range(reps):
                fn()
                ev[k + 1].record()
            torch.cuda.synchronize()
            return min(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))

        best_ms = None
        for c in cands:
            enc = EncodedBatch(flat[: n_streams * c].view(n_streams, c), n_words, status, tuple(config))
            enc_fn(symbols, model, config, layout, out=enc)
            te, td = ms(lambda: enc_fn(symbols, model, config, layout, out=enc), 2), ms(lambda: dec_fn(enc, model, n_per, layout, out=decoded), 3)
            if report is not None:
                report.append((c, te, td))
            if best_ms is None or te + td < best_ms:
                best, best_ms = c, te + td
        del flat, decoded
    _TUNED_STRIDES[key] = best
    if cache_path and report is None:
        try:
            known = {}
            if os.path.exists(cache_path):
                with open(cache_path) as f:
                    known = json.load(f)
            known[cache_key] = best
            with open(cache_path, "w") as f:
                json.dump(known, f, indent=1, sort_keys=True)
        except (OSError, ValueError, TypeError):
            pass                                    # (a cache that cannot be written is only a cache)
    return best


def _layout_shape(symbols: torch.Tensor, layout: str):
    if symbols.dim() != 2:
        raise ValueError("symbols must be 2-d")
    if layout == "stream_major":
        return symbols.shape[0], symbols.shape[1], N.LAYOUT_STREAM_MAJOR
    if layout == "symbol_major":
        return symbols.shape[1], symbols.shape[0], N.LAYOUT_SYMBOL_MAJOR
    raise ValueError("layout must be 'stream_major' or 'symbol_major'")


def _new_batch(n_streams, stride, dev, config, packed16=False) -> EncodedBatch:
    return EncodedBatch(torch.empty((n_streams, stride), dtype=torch.int16 if packed16 else torch.int32, device=dev),
                        torch.empty(n_streams, dtype=torch.int32, device=dev),
                        torch.empty(n_streams, dtype=torch.int32, device=dev), tuple(config))


def _jump_interval(jump_points, n_per: int, auto) -> int:
    """`jump_points` of an encode call -> symbols between jump points (0 = none).  "auto": the library's own choice for this batch
    (`auto()` asks cst_jump_points_auto); an integer k: k jump points per stream (must divide the row length)."""
    if isinstance(jump_points, str):
        if jump_points != "auto":
            raise ValueError("jump_points must be 'auto' or a number of jump points per stream")
        return int(auto())
    k = int(jump_points or 0)
    if k == 0:
        return 0
    if k < 0 or n_per % k != 0:
        raise ValueError("jump_points must divide the number of symbols per stream")
    return n_per // k


def _jump_table(out: EncodedBatch, kind, interval: int, n_streams: int, n_per: int, dev):
    """the jump table of `out` for this shape: the batch's own one if it has the right form (coding again into the same buffers
    allocates nothing), else a new one"""
    n_chunks = (n_per + interval - 1) // interval
    ck = out.jump
    if isinstance(ck, kind) and ck.interval == interval and tuple(ck.pos.shape) == (n_streams, n_chunks) and ck.pos.device == dev:
        return ck
    if kind is Checkpoints:
        return Checkpoints(interval, torch.zeros((n_streams, n_chunks), dtype=torch.int32, device=dev),
                           torch.zeros((n_streams, n_chunks), dtype=torch.int64, device=dev))
    return RangeCheckpoints(interval, *(torch.zeros((n_streams, n_chunks), dtype=dt, device=dev) for dt in (torch.int32, torch.int64, torch.int64)))


def ans_encode(symbols: torch.Tensor, model: Model, config=(32, 64, 12), layout="stream_major",
               stride=None, out: Optional[EncodedBatch] = None, packed16: bool = False, jump_points="auto") -> EncodedBatch:
    """One AnsCoder per stream: encode_iid_symbols_reverse + into_compressed (stack.rs:835-849, 891-895).
    jump_points: the encoder can also note `AnsCoder.pos()` (stack.rs:1107-1139) in front of every k-th part of a stream -- the
    words are unchanged, the batch carries the points as `.jump`, and ans_decode then decodes every part on a lane of its own (two
    or more waves per SIMD where a 65 536-stream batch has one).  "auto" (default): where that pays and costs the encoder nothing
    (cst_jump_points_auto: int8 / int16 matrices, tables per stream, P > 12, fewer streams than the chip has lanes; none for
    int32 symbols with one table of P <= 12 at 65 536 streams or more); an integer k: k points per stream (k divides the rows);
    0: none.
    stride: words per slab (default max_words), or "tuned" for the stride measured fastest for this shape (tuned_stride).
    On packed 16-bit words and symbol-major matrices "auto" notes none; an explicit k does (the generic checkpointing encoder supplies the
    table -- slower -- and ans_decode decodes the chunks through the plain decoder with raw states: random access, not speed).
    packed16 (the (16,32) preset only): the words two per 32-bit slot, as the reference's Vec<u16> (CST_FLAG_PACKED_W16) -- the
    batch's `words` is then an int16 tensor; ans_decode and compact recognise it.
    out: an EncodedBatch of an earlier call with the same shapes, to code into the same buffers (its jump table is reused,
    replaced or dropped as this call's jump_points say)."""
    if isinstance(stride, str):
        if stride != "tuned":
            raise ValueError("stride must be a number of words or 'tuned'")
        stride = tuned_stride(symbols, model, config, layout) if out is None else None
    given = symbols
    narrow = _SYMBOL_BYTES.get(symbols.dtype, 4) if symbols.dtype in _SYMBOL_BYTES else 4
    if narrow != 4:
        # int8 / int16 symbol matrices (the reference's Symbol is generic, quantize.rs:229-255; cst_ans_encode_batch_sym): rows of whole
        # 128-byte lines at (32, 64, P <= 12) are read by the encoder loops themselves (round 5: a quarter / half of the symbol bytes in
        # HBM and on the link); every other shape is widened on the device next to the int32 call
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only (map the symbols to indices first)")
        symbols = _require_cuda(symbols, symbols.dtype, "symbols")
    else:
        symbols = _to_indices(model, _require_cuda(symbols, torch.int32, "symbols"))
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    if out is None:
        out = _new_batch(n_streams, stride or max_words(n_per, config), symbols.device, config, packed16)
    L = N.lib()
    # (packed 16-bit words: "auto" notes none -- the packed decoder's LDS image leaves one workgroup per CU, more lanes would only
    #  queue; an explicit k is honoured for what the reference's Pos / Seek is FOR, random access, see below)
    interval = _jump_interval(0 if out.packed16 and isinstance(jump_points, str) else jump_points, n_per, lambda: L.cst_jump_points_auto(
        model._h, _cfg(*config), N.CODER_ANS, narrow, _ptr(symbols), n_streams, n_per, lay, _ptr(out.words), out.words.shape[1]))
    packed_jump = None
    if interval and out.packed16:
        # AnsCoder::pos() does not depend on how the words are stored (stack.rs:1107-1139: words in the bulk + state).  Chunks of whole
        # tiles: the packed encoder notes them itself.  Other intervals: the WORDS come from the packed encoder below, the TABLE from
        # the checkpointing encoder of the unpacked preset run beside it into a scratch slab (same recurrence, same counts)
        if narrow != 4:
            raise ValueError("packed16 with jump points: int32 symbols")
        packed_jump = _jump_table(out, Checkpoints, interval, n_streams, n_per, symbols.device)
        if interval % 32 == 0 and layout == "stream_major" and 8 <= config[2] <= 12 and not model.noncontiguous:
            # chunks of whole tiles: the packed encoder notes the points on its way (cst_ans_encode_batch_ckpt_packed16)
            N.check(L.cst_ans_encode_batch_ckpt_packed16(model._h, _cfg(*config), _ptr(symbols), n_streams, n_per, _ptr(out.words), out.words.shape[1],
                                                         _ptr(out.n_words), interval, _ptr(packed_jump.pos), _ptr(packed_jump.state), _ptr(out.status),
                                                         _stream_ptr()), "cst_ans_encode_batch_ckpt_packed16")
            out.jump = packed_jump
            _stamp_fresh(out)
            return out
        tmp = _new_batch(n_streams, max_words(n_per, config), symbols.device, config, False)
        ans_encode_checkpointed(given, model, interval, config, layout, out=(tmp, packed_jump))
        del tmp
        interval = 0
    if interval:
        ck = _jump_table(out, Checkpoints, interval, n_streams, n_per, symbols.device)
        ans_encode_checkpointed(given, model, interval, config, layout, out=(out, ck))
        out.jump = ck
        _stamp_fresh(out)
        return out
    out.jump = None              # (an earlier call's jump points describe an earlier call's words)
    flags = N.FLAG_PACKED_W16 if out.packed16 else N.FLAG_NONE
    if narrow != 4:
        scratch = _ckpt_scratch("widen", symbols.device, L.cst_symbols_scratch_bytes(n_streams, n_per, narrow))
        N.check(L.cst_ans_encode_batch_sym(model._h, _cfg(*config), _ptr(symbols), narrow, n_streams, n_per, lay, _ptr(out.words),
                                           out.words.shape[1], _ptr(out.n_words), None, _ptr(out.status), flags, _ptr(scratch),
                                           _stream_ptr()), "cst_ans_encode_batch_sym")
    else:
        N.check(L.cst_ans_encode_batch(model._h, _cfg(*config), _ptr(symbols), n_streams, n_per, lay, _ptr(out.words),
                                       out.words.shape[1], _ptr(out.n_words), None, _ptr(out.status), flags,
                                       _stream_ptr()), "cst_ans_encode_batch")
    out.jump = packed_jump
    _stamp_fresh(out)
    return out


def ans_roundtrip_launcher(symbols: torch.Tensor, model: Model, encoded: EncodedBatch, decoded: torch.Tensor, layout="stream_major"):
    """Returns step(): one cst_ans_encode_batch + one cst_ans_decode_batch over fixed buffers, with every argument converted
    once.  For loops that launch the same pair many times (bench.py's timed steps, a server coding batch after batch into the
    same buffers): ~5 us of host time per launch instead of ~40 us through ans_encode / ans_decode, so that a busy host
    core does not show up as gaps between 0.3-ms kernels.  The launches go to the stream that is current when step() runs."""
    symbols = _to_indices(model, _require_cuda(symbols, torch.int32, "symbols"))
    if model.noncontiguous:
        raise ValueError("ans_roundtrip_launcher: contiguous alphabets only")
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    cfg = _cfg(*encoded.config)
    lib = N.lib()
    enc_fn, dec_fn = lib.cst_ans_encode_batch, lib.cst_ans_decode_batch
    words, n_words, status = encoded.words, encoded.n_words, encoded.status
    dstatus = torch.empty(n_streams, dtype=torch.int32, device=symbols.device)
    enc_args = (model._h, cfg, _ptr(symbols), n_streams, n_per, lay, _ptr(words), words.shape[1], _ptr(n_words), None, _ptr(status), N.FLAG_NONE)
    dec_args = (model._h, cfg, _ptr(words), None, words.shape[1], words.numel(), _ptr(n_words), _ptr(decoded), n_streams, n_per, lay, None, None,
                _ptr(dstatus), N.FLAG_NONE)
    keep = (symbols, words, n_words, status, decoded, dstatus, model)      # (the pointers above stay valid as long as step does)

    def step():
        sp = _stream_ptr()
        rc = enc_fn(*enc_args, sp) or dec_fn(*dec_args, sp)
        if rc:
            N.check(rc, "ans_roundtrip_launcher")
    step.keep = keep
    step.decode_status = dstatus
    return step


def ans_decode(encoded, model: Model, n_per_stream: int, layout="stream_major", offsets: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, config=None, cold: Optional[bool] = None, dtype=torch.int32):
    """One AnsCoder per stream: from_compressed + decode_iid_symbols (stack.rs:299-318, mod.rs:1016-1031).
    dtype (or the dtype of `out`): torch.int32, or int16 / int8 for a narrow symbol matrix (cst_ans_decode_batch_sym: written by the
    decoder loops themselves where the rows are whole 128-byte lines at (32, 64, P <= 12), narrowed on the device next to the int32
    call otherwise; the model's support must fit the type).

    `encoded` is an EncodedBatch, or (words, n_words) with `offsets` for the packed layout.  `cold` (CST_FLAG_COLD_WORDS, a hint
    that never changes results): are the words NOT expected in the GPU's caches?  Default None = decided by provenance: hot only
    if `encoded` is the EncodedBatch that the last ans_encode on this HIP stream filled (and small enough to have stayed in the
    caches); words that came from the host, a peer or a file -- plain tensors, packed + offsets -- are cold."""
    jump = encoded.jump if isinstance(encoded, EncodedBatch) else None
    if isinstance(jump, Checkpoints) and offsets is None and \
            n_per_stream == jump.interval * jump.pos.shape[1] and jump.pos.shape[0] == encoded.n_words.numel():
        # the batch carries jump points for exactly this decode (ans_encode, jump_points): every part of a stream on a lane of its
        # own.  (A prefix of the streams, n_per_stream < what was encoded, is a plain decode: the table's rows have another stride.)
        dec, part_status = ans_decode_checkpointed(encoded, jump, model, n_per_stream, out=out, dtype=dtype if out is None else out.dtype,
                                                   layout=layout)
        return dec, _status_per_stream(part_status)
    if isinstance(encoded, EncodedBatch):
        words, n_words, config = encoded.words, encoded.n_words, config or encoded.config
        stride = words.shape[1]
    else:
        words, n_words = encoded
        stride = words.shape[1] if words.dim() == 2 else 0
        config = config or (32, 64, 12)
    if cold is None:
        cold = _words_are_cold(encoded) if isinstance(encoded, EncodedBatch) else True
    n_streams = n_words.numel()
    dev = words.device
    if out is None:
        shape = (n_streams, n_per_stream) if layout == "stream_major" else (n_per_stream, n_streams)
        out = torch.empty(shape, dtype=dtype, device=dev)
    lay = N.LAYOUT_STREAM_MAJOR if layout == "stream_major" else N.LAYOUT_SYMBOL_MAJOR
    status = torch.empty(n_streams, dtype=torch.int32, device=dev)
    narrow = _SYMBOL_BYTES.get(out.dtype)
    if narrow is None:
        raise TypeError("decoded symbols are int32, int16 or int8")
    flags = (N.FLAG_COLD_WORDS if cold else N.FLAG_NONE) | (N.FLAG_PACKED_W16 if words.dtype == torch.int16 else N.FLAG_NONE)
    if narrow != 4:
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only")
        L = N.lib()
        scratch = _ckpt_scratch("narrow", dev, L.cst_symbols_scratch_bytes(n_streams, n_per_stream, narrow))
        N.check(L.cst_ans_decode_batch_sym(model._h, _cfg(*config), _ptr(words), _ptr(offsets), stride, words.numel(), _ptr(n_words),
                                           _ptr(out), narrow, n_streams, n_per_stream, lay, None, None, _ptr(status),
                                           flags, _ptr(scratch), _stream_ptr()), "cst_ans_decode_batch_sym")
        return out, status
    N.check(N.lib().cst_ans_decode_batch(model._h, _cfg(*config), _ptr(words), _ptr(offsets), stride, words.numel(), _ptr(n_words),
                                         _ptr(out), n_streams, n_per_stream, lay, None, None, _ptr(status),
                                         flags, _stream_ptr()), "cst_ans_decode_batch")
    return _to_symbols(model, out), status


_scratch = {}


# ---------------------------------------------------------------------------------------------------------------------
# streams of different lengths: thousands of small coders with one model in one launch (the reference's "compressed index"
# pattern, tests/issue52.rs: one AnsCoder per document)
# ---------------------------------------------------------------------------------------------------------------------

@dataclass
class RaggedBatch:
    """Compressed words of streams of different lengths: stream s owns words[word_offsets[s] : word_offsets[s] + n_words[s]]."""
    words: torch.Tensor          # int32, flat
    word_offsets: torch.Tensor   # int64 [n_streams + 1]: the slabs (a slab is at least as long as its stream can get)
    n_words: torch.Tensor        # int32 [n_streams]
    status: torch.Tensor         # int32 [n_streams]
    config: tuple
    order: Optional[torch.Tensor] = None   # int32 [n_streams]: the schedule the encoder ran with (the decoders reuse it)
    jump: Optional["RaggedJump"] = None    # jump points the encoder noted (ans_encode_ragged, jump_every): ans_decode_ragged decodes the chunks side by side

    def stream(self, s: int) -> np.ndarray:
        """get_compressed() of stream s (uint32, host)."""
        lo, n = int(self.word_offsets[s].item()), int(self.n_words[s].item())
        return self.words[lo: lo + n].cpu().numpy().view(np.uint32)


@dataclass
class RaggedJump:
    """AnsCoder.pos() in front of every `interval` symbols of every stream of a RaggedBatch (stack.rs:1107-1139): chunk j of stream s is
    entry chunk_offsets[s] + j of pos (words in the bulk) / state (the coder state there)."""
    interval: int
    chunk_offsets: torch.Tensor   # int64 [n_streams + 1]: exclusive prefix sum of ceil(length / interval)
    pos: torch.Tensor             # int32 [>= total chunks]
    state: torch.Tensor           # int64 [>= total chunks]


RAGGED_JUMP_EVERY = 256            # jump_every="auto": symbols between the jump points of a ragged batch


def ragged(sequences, device="cuda"):
    """list of 1-d integer arrays -> (flat int32 device tensor, int64 offsets [n + 1]) as the ragged entry points take them."""
    lengths = np.fromiter((len(x) for x in sequences), dtype=np.int64, count=len(sequences))
    offsets = np.zeros(len(sequences) + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    flat = np.concatenate([np.asarray(x, dtype=np.int32).reshape(-1) for x in sequences]) if len(sequences) else np.zeros(0, np.int32)
    return torch.from_numpy(flat).to(device), torch.from_numpy(offsets).to(device)


RAGGED_BALANCE_FROM = 250_000      # streams from which `order="auto"` sorts: below, the launch is bound by its longest stream anyway


def ragged_order(keys: torch.Tensor) -> torch.Tensor:
    """The schedule of the `*_ragged_ordered` entry points for streams whose cost grows with `keys` (lengths for the encoder,
    word counts for the decoders): stream indices, longest first, so that the 64 streams of a wave are about equally long."""
    return torch.sort(keys.to(torch.int32), descending=True).indices.to(torch.int32)


def _ragged_order(order, n_streams, keys):
    """`order` argument of the ragged calls: None (slot i = stream i), "auto" (sorted from RAGGED_BALANCE_FROM streams on),
    "sorted", or an int32 permutation on the device."""
    if order is None or (isinstance(order, str) and order == "auto" and n_streams < RAGGED_BALANCE_FROM):
        return None
    if isinstance(order, str):
        if order not in ("auto", "sorted"):
            raise ValueError('order: None, "auto", "sorted" or an int32 tensor of stream indices')
        return ragged_order(keys)
    order = _require_cuda(order, torch.int32, "order")
    if order.numel() != n_streams:
        raise ValueError("order must hold one stream index per stream")
    return order


def ans_encode_ragged(symbols: torch.Tensor, sym_offsets: torch.Tensor, model: Model, config=(32, 64, 24), order="auto",
                      jump_every="auto") -> RaggedBatch:
    """One AnsCoder per stream, streams of different lengths (`symbols` flat, stream s = symbols[sym_offsets[s]:sym_offsets[s+1]]):
    encode_iid_symbols_reverse + into_compressed per stream (stack.rs:835-849, 891-895) in ONE launch.  `order`: see
    _ragged_order (big batches of very different lengths run 1.5 - 3x faster with their streams sorted by length; the
    results do not depend on it).
    jump_every: the encoder also notes AnsCoder.pos() in front of every jump_every symbols of every stream (a multiple of 8; the words
    are unchanged) and the batch carries the table as `.jump`: ans_decode_ragged then decodes all chunks side by side -- a launch
    lasts as long as its longest CHAIN, and a 2000-symbol document among short ones is 2000 dependent steps without jump points.
    "auto" (default): every RAGGED_JUMP_EVERY symbols if the streams average more than a quarter of that; 0: none."""
    symbols = _to_indices(model, _require_cuda(symbols, torch.int32, "symbols"))
    sym_offsets = _require_cuda(sym_offsets, torch.int64, "sym_offsets")
    n_streams = sym_offsets.numel() - 1
    if symbols.dim() != 1 or n_streams < 0:
        raise ValueError("symbols must be flat and sym_offsets hold n_streams + 1 entries")
    W, S, P = config
    lengths = sym_offsets[1:] - sym_offsets[:-1]
    # a stream of n symbols fills at most min(n, ceil(n P / W)) + S / W words (cst_ans_max_words without its rounding to 64 bytes:
    # thousands of short streams; 16-byte slabs keep the chunk stores aligned)
    bound = torch.minimum(lengths, (lengths * P + (W - 1)) // W) + S // W
    slabs = (bound + 3) // 4 * 4
    word_offsets = torch.zeros(n_streams + 1, dtype=torch.int64, device=symbols.device)
    torch.cumsum(slabs, 0, out=word_offsets[1:])
    total = int(word_offsets[-1].item()) if n_streams else 0
    dev = symbols.device
    order = _ragged_order(order, n_streams, lengths)
    out = RaggedBatch(torch.empty(max(total, 4), dtype=torch.int32, device=dev), word_offsets,
                      torch.empty(n_streams, dtype=torch.int32, device=dev), torch.empty(n_streams, dtype=torch.int32, device=dev), tuple(config),
                      order)
    if isinstance(jump_every, str):
        if jump_every != "auto":
            raise ValueError("jump_every: 'auto', 0 or a multiple of 8")
        jump_every = RAGGED_JUMP_EVERY if n_streams > 0 and symbols.numel() * 4 > n_streams * RAGGED_JUMP_EVERY else 0
    jump_every = int(jump_every or 0)
    if jump_every < 0 or jump_every % 8 != 0:
        raise ValueError("jump_every: 'auto', 0 or a multiple of 8")
    if jump_every and n_streams > 0:
        chunk_offsets = torch.zeros(n_streams + 1, dtype=torch.int64, device=dev)
        torch.cumsum((lengths + (jump_every - 1)) // jump_every, 0, out=chunk_offsets[1:])
        # (no read-back of the total: sum(ceil(len / I)) <= total / I + n_streams bounds it, entries behind the last chunk stay unused)
        n_chunks = symbols.numel() // jump_every + n_streams
        jump = RaggedJump(jump_every, chunk_offsets, torch.empty(max(n_chunks, 1), dtype=torch.int32, device=dev),
                          torch.empty(max(n_chunks, 1), dtype=torch.int64, device=dev))
        N.check(N.lib().cst_ans_encode_ragged_jump(model._h, _cfg(*config), _ptr(symbols), _ptr(sym_offsets), n_streams,
                                                   _ptr(order) if order is not None else None, _ptr(out.words), _ptr(word_offsets), 0,
                                                   _ptr(out.n_words), jump_every, _ptr(chunk_offsets), _ptr(jump.pos), _ptr(jump.state),
                                                   _ptr(out.status), _stream_ptr()), "cst_ans_encode_ragged_jump")
        out.jump = jump
        return out
    N.check(N.lib().cst_ans_encode_ragged_ordered(model._h, _cfg(*config), _ptr(symbols), _ptr(sym_offsets), n_streams,
                                                  _ptr(order) if order is not None else None, _ptr(out.words), _ptr(word_offsets), 0,
                                                  _ptr(out.n_words), _ptr(out.status), _stream_ptr()), "cst_ans_encode_ragged_ordered")
    return out


def ans_decode_ragged(encoded: RaggedBatch, model: Model, sym_offsets: torch.Tensor, out: Optional[torch.Tensor] = None, order="auto"):
    """from_compressed + decode_iid_symbols per stream (stack.rs:299-318, 1070-1100): stream s yields
    sym_offsets[s + 1] - sym_offsets[s] symbols at out[sym_offsets[s]:].  Returns (symbols flat, status per stream).
    `order`: the schedule (see _ragged_order); "auto" reuses the encoder's, or sorts by word count from RAGGED_BALANCE_FROM
    streams on."""
    sym_offsets = _require_cuda(sym_offsets, torch.int64, "sym_offsets")
    n_streams = sym_offsets.numel() - 1
    # a batch with jump points decodes its chunks side by side (they are all at most `interval` symbols long: no schedule needed) --
    # unless the caller hands in a schedule of their own, which is then honoured on the whole streams
    take_jump = encoded.jump is not None and (order is None or (isinstance(order, str) and order == "auto"))
    if isinstance(order, str) and order == "auto" and encoded.order is not None and encoded.order.numel() == n_streams:
        order = encoded.order
    order = _ragged_order(order, n_streams, encoded.n_words)
    dev = encoded.words.device
    total = int(sym_offsets[-1].item()) if n_streams > 0 else 0
    if out is None:
        out = torch.empty(total, dtype=torch.int32, device=dev)
    status = torch.empty(max(n_streams, 0), dtype=torch.int32, device=dev)
    jump = encoded.jump
    if take_jump and n_streams > 0 and jump.chunk_offsets.numel() == n_streams + 1:
        # the batch carries jump points: every chunk of every stream on a lane of its own (the table is checked against the lengths on
        # the device: a stream it does not describe reports INVALID_DATA)
        L = N.lib()
        n_chunks = int(jump.pos.numel())              # (an upper bound of the chunks is enough: entries behind the last chunk are empty streams)
        scratch = _ckpt_scratch("ragged_jump", dev, L.cst_ragged_jump_scratch_bytes(n_chunks))
        N.check(L.cst_ans_decode_ragged_jump(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(encoded.word_offsets), 0, encoded.words.numel(),
                                             _ptr(encoded.n_words), _ptr(out), _ptr(sym_offsets), n_streams, jump.interval, _ptr(jump.chunk_offsets),
                                             n_chunks, _ptr(jump.pos), _ptr(jump.state), _ptr(scratch), _ptr(status), _stream_ptr()),
                "cst_ans_decode_ragged_jump")
        return _to_symbols(model, out), status
    N.check(N.lib().cst_ans_decode_ragged_ordered(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(encoded.word_offsets), 0,
                                                  encoded.words.numel(), _ptr(encoded.n_words), _ptr(out), _ptr(sym_offsets), n_streams,
                                                  _ptr(order) if order is not None else None, _ptr(status), _stream_ptr()),
            "cst_ans_decode_ragged_ordered")
    return _to_symbols(model, out), status


def ans_decode_until(encoded: RaggedBatch, model: Model, eof_symbol: int, max_symbols: Optional[int] = None):
    """Streams whose length is not stored: every stream is decoded until `eof_symbol` (tests/issue52.rs:63-80).  Two launches
    -- count, prefix sum, decode.  Returns (symbols flat, sym_offsets [n + 1], status); the terminator is the last symbol of
    every stream; status CAPACITY (2): no terminator among the first max_symbols symbols -- such a stream (corrupt data, a wrong
    eof_symbol) decodes to NO symbols: its length is 0 in sym_offsets, so that a batch of bad documents cannot ask for
    n_streams x max_symbols symbols of output.  max_symbols defaults to 2^20."""
    max_symbols = (1 << 20) if max_symbols is None else int(max_symbols)
    if model.noncontiguous:
        raise ValueError("ans_decode_until: contiguous alphabets only")
    n_streams = encoded.n_words.numel()
    dev = encoded.words.device
    lengths = torch.zeros(n_streams, dtype=torch.int64, device=dev)
    status = torch.zeros(n_streams, dtype=torch.int32, device=dev)
    order = encoded.order if encoded.order is not None and encoded.order.numel() == n_streams else _ragged_order("auto", n_streams, encoded.n_words)
    N.check(N.lib().cst_ans_count_until_ordered(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(encoded.word_offsets), 0,
                                                encoded.words.numel(), _ptr(encoded.n_words), n_streams,
                                                _ptr(order) if order is not None else None, int(eof_symbol), int(max_symbols),
                                                _ptr(lengths), _ptr(status), _stream_ptr()), "cst_ans_count_until_ordered")
    lengths = torch.where(status != 0, torch.zeros_like(lengths), lengths)      # (a stream without a terminator is not decoded)
    sym_offsets = torch.zeros(n_streams + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lengths, 0, out=sym_offsets[1:])
    symbols, status2 = ans_decode_ragged(encoded, model, sym_offsets, order=order)
    return symbols, sym_offsets, torch.where(status != 0, status, status2)


def _compact_scratch(device, n_streams):
    # (one scratch per device AND stream: two compactions on different streams must not share ticket / status words)
    need = N.load_library().cst_compact_scratch_bytes(n_streams)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _scratch.get(key)
    if t is None or t.numel() < need:
        t = torch.empty(max(need, 4096), dtype=torch.uint8, device=device)
        _scratch[key] = t
    return t


def compact(encoded: EncodedBatch, capacity: Optional[int] = None, out=None):
    """Packs the slabs: returns (packed uint32 words as int32 tensor, offsets int64[n_streams+1]); offsets[-1] = total words.

    One asynchronous kernel (single-pass scan fused with the gather), no host synchronisation: `packed` has `capacity`
    words (default: the upper bound n_streams * stride, so that nothing can overflow) of which ONLY the first offsets[-1]
    are valid -- slice with `packed[:int(offsets[-1])]` (that reads the total back from the device; callers that stay on
    the device hand `packed` and `offsets` on as they are).  `out=(packed, offsets)` reuses buffers of an earlier call.
    With a caller-chosen `capacity` or `out` the words may not fit: streams that would overflow are left out by the kernel,
    and this function then reads offsets[-1] back (one synchronisation, only in this case) and raises."""
    n_streams = encoded.n_words.numel()
    dev = encoded.words.device
    if out is not None:
        packed, offsets = out
    else:
        capacity = capacity if capacity is not None else n_streams * encoded.words.shape[1]
        packed = torch.empty(max(capacity, 1), dtype=encoded.words.dtype, device=dev)
        offsets = torch.empty(n_streams + 1, dtype=torch.int64, device=dev)
    fn = N.lib().cst_compact_words16 if encoded.packed16 else N.lib().cst_compact_words       # (packed 16-bit words: everything in halfwords)
    N.check(fn(_ptr(encoded.words), encoded.words.shape[1], _ptr(encoded.n_words), n_streams,
               _ptr(offsets), _ptr(packed), packed.numel(), _ptr(_compact_scratch(dev, n_streams)),
               _stream_ptr()), "cst_compact_words")
    if (out is not None or capacity is not None) and int(offsets[-1].item()) > packed.numel():
        raise ValueError(f"compact: {int(offsets[-1].item())} words do not fit the packed buffer of {packed.numel()} words")
    return packed, offsets


def reverse_words(encoded, offsets: Optional[torch.Tensor] = None, out=None):
    """Every stream's words in the opposite order (its own inverse): from the reference's default order -- the LAST word of a
    stream is the first an ANS decoder reads -- to the order `AnsCoder::from_reversed_compressed` / `Cursor::into_reversed`
    (src/stream/stack.rs:734-748, src/backends.rs:1424-1448) use, first-read word first, and back.

    `encoded`: an `EncodedBatch` of (32, 64) slabs -> a new `EncodedBatch` (or `out=encoded` for in place); or a tuple
    `(packed int32 words, n_words)` with `offsets` int64[n_streams + 1] -> a packed tensor of the same layout."""
    if isinstance(encoded, EncodedBatch):
        if encoded.packed16:
            raise ValueError("reverse_words: 32-bit words only")
        n_streams = encoded.n_words.numel()
        res = out if out is not None else EncodedBatch(torch.empty_like(encoded.words), encoded.n_words, encoded.status, encoded.config)
        N.check(N.lib().cst_words_reverse(_ptr(encoded.words), None, encoded.words.shape[1], _ptr(encoded.n_words), n_streams,
                                          _ptr(res.words), None, res.words.shape[1], _stream_ptr()), "cst_words_reverse")
        return res
    packed, n_words = encoded
    if offsets is None:
        raise ValueError("reverse_words: packed words need their offsets")
    res = out if out is not None else torch.empty_like(packed)
    N.check(N.lib().cst_words_reverse(_ptr(packed), _ptr(offsets), 0, _ptr(n_words), n_words.numel(), _ptr(res), _ptr(offsets), 0,
                                      _stream_ptr()), "cst_words_reverse")
    return res


def range_max_words(n_per_stream: int, config=(32, 64, 12)) -> int:
    return N.load_library().cst_range_max_words(n_per_stream, _cfg(*config))


def range_encode(symbols: torch.Tensor, model: Model, config=(32, 64, 12), layout="stream_major",
                 stride=None, out: Optional[EncodedBatch] = None, jump_points="auto") -> EncodedBatch:
    """One RangeEncoder per stream: encode_iid_symbols + get_compressed (queue.rs:612-705, 458-523).
    stride: words per slab (default range_max_words), or "tuned" (tuned_stride(..., coder="range")).
    jump_points: as for ans_encode -- `RangeEncoder.pos()` (queue.rs:172-196) in front of every k-th part of a stream, carried as
    `.jump` for range_decode; "auto" (default) = cst_jump_points_auto (the range decoder's division chain is latency-bound at one
    wave per SIMD: two lanes per stream at 65 536 streams), an integer k, or 0."""
    if isinstance(stride, str):
        if stride != "tuned":
            raise ValueError("stride must be a number of words or 'tuned'")
        stride = tuned_stride(symbols, model, config, layout, coder="range") if out is None else None
    narrow = _SYMBOL_BYTES.get(symbols.dtype, 4) if symbols.dtype in _SYMBOL_BYTES else 4
    if narrow != 4:
        # int8 / int16 matrices (the reference's Symbol is generic, queue.rs:612; cst_range_encode_batch[_ckpt]_sym): int8 rows of whole
        # 32-symbol tiles at (32, 64) are read by the encoder loop itself (round 6), every other shape is widened next to the int32 call
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only (map the symbols to indices first)")
        given = symbols = _require_cuda(symbols, symbols.dtype, "symbols")
    else:
        given = _require_cuda(symbols, torch.int32, "symbols")
        symbols = _to_indices(model, given)
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    if out is None:
        out = _new_batch(n_streams, stride or range_max_words(n_per, config), symbols.device, config)
    L = N.lib()
    interval = 0 if model.n_tables != 1 else _jump_interval(jump_points, n_per, lambda: L.cst_jump_points_auto(
        model._h, _cfg(*config), N.CODER_RANGE, narrow, _ptr(symbols), n_streams, n_per, lay, _ptr(out.words), out.words.shape[1]))
    if interval:
        ck = _jump_table(out, RangeCheckpoints, interval, n_streams, n_per, symbols.device)
        range_encode_checkpointed(given, model, interval, config, layout, out=(out, ck))
        out.jump = ck
        return out
    out.jump = None
    if narrow != 4:
        scratch = _ckpt_scratch("range_sym", symbols.device, L.cst_range_sym_scratch_bytes(n_streams, n_per, 0, narrow))
        N.check(L.cst_range_encode_batch_sym(model._h, _cfg(*config), _ptr(symbols), narrow, n_streams, n_per, lay, _ptr(out.words),
                                             out.words.shape[1], _ptr(out.n_words), None, _ptr(out.status), N.FLAG_NONE, _ptr(scratch),
                                             _stream_ptr()), "cst_range_encode_batch_sym")
        return out
    N.check(L.cst_range_encode_batch(model._h, _cfg(*config), _ptr(symbols), n_streams, n_per, lay, _ptr(out.words),
                                     out.words.shape[1], _ptr(out.n_words), None, _ptr(out.status), N.FLAG_NONE,
                                     _stream_ptr()), "cst_range_encode_batch")
    return out


def range_decode(encoded, model: Model, n_per_stream: int, layout="stream_major", offsets: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, config=None, dtype=torch.int32):
    """One RangeDecoder per stream: from_compressed + decode_iid_symbols (queue.rs:847-868, 968-1033).
    dtype (or the dtype of `out`): int32, or int16 / int8 -- narrowed on the device behind the decoder (cst_symbols_narrow)."""
    if isinstance(encoded, EncodedBatch):
        words, n_words, config = encoded.words, encoded.n_words, config or encoded.config
        stride = words.shape[1]
    else:
        words, n_words = encoded
        stride = words.shape[1] if words.dim() == 2 else 0
        config = config or (32, 64, 12)
    n_streams = n_words.numel()
    dev = words.device
    dtype = out.dtype if out is not None else dtype
    if dtype not in _SYMBOL_BYTES:
        raise TypeError("decoded symbols are int32, int16 or int8")
    narrow = _SYMBOL_BYTES[dtype]
    if narrow != 4:
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only")
        info = torch.iinfo(dtype)
        if model.min_symbol < info.min or model.min_symbol + model.n_symbols - 1 > info.max:
            raise ValueError(f"the model's support does not fit {dtype}")
    jump = encoded.jump if isinstance(encoded, EncodedBatch) else None
    if isinstance(jump, RangeCheckpoints) and offsets is None and layout == "stream_major" and \
            n_per_stream == jump.interval * jump.pos.shape[1] and jump.pos.shape[0] == n_streams:
        # the batch carries jump points for exactly this decode: every part of a stream on a lane of its own
        dec, part_status = range_decode_checkpointed(encoded, jump, model, n_per_stream, out=out, dtype=dtype)
        return dec, _status_per_stream(part_status)
    if out is None:
        shape = (n_streams, n_per_stream) if layout == "stream_major" else (n_per_stream, n_streams)
        out = torch.empty(shape, dtype=dtype, device=dev)
    lay = N.LAYOUT_STREAM_MAJOR if layout == "stream_major" else N.LAYOUT_SYMBOL_MAJOR
    status = torch.empty(n_streams, dtype=torch.int32, device=dev)
    L = N.lib()
    if narrow != 4:
        # (cst_range_decode_batch_sym: int8 stream-major matrices are written by the sub-lane decoder itself, one lane per stream;
        #  every other shape is narrowed on the device behind the int32 decoder)
        scratch = _ckpt_scratch("range_sym", dev, L.cst_range_sym_scratch_bytes(n_streams, n_per_stream, 0, narrow))
        N.check(L.cst_range_decode_batch_sym(model._h, _cfg(*config), _ptr(words), _ptr(offsets), stride, words.numel(), _ptr(n_words),
                                             _ptr(out), narrow, n_streams, n_per_stream, lay, None, _ptr(status), N.FLAG_NONE, _ptr(scratch),
                                             _stream_ptr()), "cst_range_decode_batch_sym")
        return out, status
    N.check(L.cst_range_decode_batch(model._h, _cfg(*config), _ptr(words), _ptr(offsets), stride, words.numel(), _ptr(n_words),
                                     _ptr(out), n_streams, n_per_stream, lay, None, _ptr(status), N.FLAG_NONE,
                                     _stream_ptr()), "cst_range_decode_batch")
    return _to_symbols(model, out), status


# ---------------------------------------------------------------------------------------------------------------------
# per-symbol quantized Gaussians: the reference's flagship call for many streams at once
#     coder.encode_reverse(symbols, QuantizedGaussian(lo, hi), means, stds) / coder.decode(family, means, stds)
# (src/pybindings/stream/stack.rs:567-588, 733-751): every symbol has its own (mean, std); `means` / `stds` are float64
# tensors of the shape and layout of the symbol matrix (float32 callers widen first, src/pybindings/mod.rs:211-216).
# ---------------------------------------------------------------------------------------------------------------------

def _gaussian_args(symbols_shape, means, stds):
    # float32 parameter matrices are widened, exactly as the reference's Python API does (PyReadonlyFloatArray::cast_f64,
    # src/pybindings/mod.rs:187-214; its own doc example passes float32 arrays): the models are those of the widened values
    if means.dtype == torch.float32:
        means = means.to(torch.float64)
    if stds.dtype == torch.float32:
        stds = stds.to(torch.float64)
    means = _require_cuda(means, torch.float64, "means")
    stds = _require_cuda(stds, torch.float64, "stds")
    if tuple(means.shape) != tuple(symbols_shape) or tuple(stds.shape) != tuple(symbols_shape):
        raise ValueError("means and stds must have the shape of the symbol matrix")
    return means, stds


def _encode_gaussian(fn_name, max_words_fn, symbols, min_symbol, max_symbol, means, stds, config, layout, stride, out):
    symbols = _require_cuda(symbols, torch.int32, "symbols")
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    means, stds = _gaussian_args(symbols.shape, means, stds)
    if out is None:
        stride = stride or max_words_fn(n_per, config)
        dev = symbols.device
        out = EncodedBatch(torch.empty((n_streams, stride), dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev), tuple(config))
    N.check(getattr(N.lib(), fn_name)(_cfg(*config), int(min_symbol), int(max_symbol), _ptr(symbols), _ptr(means), _ptr(stds),
                                      n_streams, n_per, lay, _ptr(out.words), out.words.shape[1], _ptr(out.n_words), None,
                                      _ptr(out.status), N.FLAG_NONE, _stream_ptr()), fn_name)
    return out


def ans_encode_gaussian(symbols, min_symbol, max_symbol, means, stds, config=(32, 64, 24), layout="stream_major",
                        stride: Optional[int] = None, out: Optional[EncodedBatch] = None, jump_points="auto") -> EncodedBatch:
    """One AnsCoder per stream: encode_reverse(symbols[s], QuantizedGaussian(min, max), means[s], stds[s]) + get_compressed.
    jump_points: as for ans_encode ("auto" = cst_jump_points_auto_gaussian: batches the fused encoder takes, of at most one wave of
    streams per SIMD, get the points that give ans_decode_gaussian two; an integer k; 0)."""
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    interval = _jump_interval(jump_points, n_per, lambda: N.lib().cst_jump_points_auto_gaussian(_cfg(*config), N.CODER_ANS, n_streams, n_per, lay))
    if interval:
        if layout != "stream_major":
            raise ValueError("jump_points: stream-major batches only")
        if out is None:
            out = _new_batch(n_streams, stride or max_words(n_per, config), symbols.device, config)
        ck = _jump_table(out, Checkpoints, interval, n_streams, n_per, symbols.device)
        ans_encode_gaussian_checkpointed(symbols, min_symbol, max_symbol, means, stds, interval, config, out=(out, ck))
        out.jump = ck
        return out
    out = _encode_gaussian("cst_ans_encode_gaussian_batch", max_words, symbols, min_symbol, max_symbol, means, stds, config,
                           layout, stride, out)
    out.jump = None
    return out


def range_encode_gaussian(symbols, min_symbol, max_symbol, means, stds, config=(32, 64, 24), layout="stream_major",
                          stride: Optional[int] = None, out: Optional[EncodedBatch] = None, jump_points="auto") -> EncodedBatch:
    """One RangeEncoder per stream: encode(symbols[s], QuantizedGaussian(min, max), means[s], stds[s]) + get_compressed
    (src/pybindings/stream/queue.rs:343-410).  jump_points: as for ans_encode_gaussian (RangeEncoder.pos(), queue.rs:172-196)."""
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    interval = _jump_interval(jump_points, n_per, lambda: N.lib().cst_jump_points_auto_gaussian(_cfg(*config), N.CODER_RANGE, n_streams, n_per, lay))
    if interval:
        if layout != "stream_major":
            raise ValueError("jump_points: stream-major batches only")
        symbols = _require_cuda(symbols, torch.int32, "symbols")
        means, stds = _gaussian_args(symbols.shape, means, stds)
        if out is None:
            out = _new_batch(n_streams, stride or range_max_words(n_per, config), symbols.device, config)
        ck = _jump_table(out, RangeCheckpoints, interval, n_streams, n_per, symbols.device)
        N.check(N.lib().cst_range_encode_gaussian_batch_ckpt(_cfg(*config), int(min_symbol), int(max_symbol), _ptr(symbols), _ptr(means), _ptr(stds),
                                                             n_streams, n_per, lay, _ptr(out.words), out.words.shape[1], _ptr(out.n_words), interval,
                                                             _ptr(ck.pos), _ptr(ck.lower), _ptr(ck.range), _ptr(out.status), _stream_ptr()),
                "cst_range_encode_gaussian_batch_ckpt")
        out.jump = ck
        return out
    out = _encode_gaussian("cst_range_encode_gaussian_batch", range_max_words, symbols, min_symbol, max_symbol, means, stds,
                           config, layout, stride, out)
    out.jump = None
    return out


def _decode_gaussian(fn_name, ans, encoded, min_symbol, max_symbol, means, stds, layout, offsets, out, config):
    if isinstance(encoded, EncodedBatch):
        words, n_words, config = encoded.words, encoded.n_words, config or encoded.config
        stride = words.shape[1]
    else:
        words, n_words = encoded
        stride = words.shape[1] if words.dim() == 2 else 0
        config = config or (32, 64, 24)
    if means.dim() != 2:
        raise ValueError("means must be 2-d")
    n_streams, n_per, lay = _layout_shape(means, layout)
    if n_streams != n_words.numel():
        raise ValueError("means / stds do not match the number of streams")
    means, stds = _gaussian_args(means.shape, means, stds)
    dev = words.device
    if out is None:
        out = torch.empty(tuple(means.shape), dtype=torch.int32, device=dev)
    status = torch.empty(n_streams, dtype=torch.int32, device=dev)
    args = [_cfg(*config), int(min_symbol), int(max_symbol), _ptr(words), _ptr(offsets), stride, words.numel(), _ptr(n_words), _ptr(means), _ptr(stds),
            _ptr(out), n_streams, n_per, lay, None]
    if ans:
        args.append(None)          # d_n_words_out
    N.check(getattr(N.lib(), fn_name)(*args, _ptr(status), N.FLAG_NONE, _stream_ptr()), fn_name)
    return out, status


# ---------------------------------------------------------------------------------------------------------------------
# chain coders (src/stream/chain.rs), many at once.  A chain has two word stacks and two heads; a call pops one stack and
# pushes the other (decode: pops `compressed`, pushes `remainders`; encode: the reverse).
# ---------------------------------------------------------------------------------------------------------------------

@dataclass
class ChainBatch:
    """words / n_words: the stack that the next call pops, stream s = words[s, :n_words[s]], consumed from the end;
    heads: int64 [n_streams, 2] = cst_chain_heads (remainders head; compressed head in the low word).  Calls update
    n_words and heads in place and return what they pushed."""
    words: torch.Tensor
    n_words: torch.Tensor
    heads: torch.Tensor
    config: tuple = (32, 64, 24)


def _chain_call(fn_name, chains: ChainBatch, n_streams, n_per, call):
    dev = chains.words.device
    pushed = torch.empty((n_streams, max(n_per, 1)), dtype=torch.int32, device=dev)
    n_pushed = torch.empty(n_streams, dtype=torch.int32, device=dev)
    status = torch.empty(n_streams, dtype=torch.int32, device=dev)
    N.check(call(pushed, n_pushed, status), fn_name)
    return pushed, n_pushed, status


def chain_decode_gaussian(chains: ChainBatch, min_symbol, max_symbol, means, stds, layout="stream_major"):
    """ChainCoder.decode(QuantizedGaussian(min, max), means[s], stds[s]) for every chain.
    Returns (symbols, pushed remainders words [n_streams, n_per], their counts, status)."""
    n_streams, n_per, lay = _layout_shape(means, layout)
    means, stds = _gaussian_args(means.shape, means, stds)
    out = torch.empty(tuple(means.shape), dtype=torch.int32, device=means.device)
    fn = "cst_chain_decode_gaussian_batch"
    pushed, n_pushed, status = _chain_call(fn, chains, n_streams, n_per, lambda pw, pn, st: getattr(N.lib(), fn)(
        _cfg(*chains.config), int(min_symbol), int(max_symbol), _ptr(chains.words), None, chains.words.shape[1], _ptr(chains.n_words),
        _ptr(means), _ptr(stds), _ptr(out), n_streams, n_per, lay, _ptr(pw), pw.shape[1], _ptr(pn), _ptr(chains.heads), _ptr(st),
        _stream_ptr()))
    return out, pushed, n_pushed, status


def chain_encode_gaussian(chains: ChainBatch, symbols, min_symbol, max_symbol, means, stds, layout="stream_major"):
    """ChainCoder.encode_reverse(symbols[s], QuantizedGaussian(min, max), means[s], stds[s]) for every chain; `chains` holds
    the REMAINDERS stacks.  Returns (pushed compressed words [n_streams, n_per], their counts, status)."""
    symbols = _require_cuda(symbols, torch.int32, "symbols")
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    means, stds = _gaussian_args(symbols.shape, means, stds)
    fn = "cst_chain_encode_gaussian_batch"
    return _chain_call(fn, chains, n_streams, n_per, lambda pw, pn, st: getattr(N.lib(), fn)(
        _cfg(*chains.config), int(min_symbol), int(max_symbol), _ptr(symbols), _ptr(means), _ptr(stds), n_streams, n_per, lay,
        _ptr(chains.words), None, chains.words.shape[1], _ptr(chains.n_words), _ptr(pw), pw.shape[1], _ptr(pn), _ptr(chains.heads),
        _ptr(st), _stream_ptr()))


def release_scratch():
    """Hands the scratch memory the per-symbol calls keep in the device's memory pool back to the system."""
    N.check(N.lib().cst_release_scratch(), "cst_release_scratch")


def ans_decode_gaussian(encoded, min_symbol, max_symbol, means, stds, layout="stream_major", offsets=None, out=None, config=None):
    """One AnsCoder per stream: AnsCoder(words[s]).decode(QuantizedGaussian(min, max), means[s], stds[s]).  A batch that carries
    jump points for exactly this shape (ans_encode_gaussian, jump_points) decodes every part of a stream on a lane of its own."""
    jump = encoded.jump if isinstance(encoded, EncodedBatch) else None
    if isinstance(jump, Checkpoints) and offsets is None and layout == "stream_major" and means.dim() == 2 and \
            tuple(means.shape) == (jump.pos.shape[0], jump.interval * jump.pos.shape[1]) and jump.pos.shape[0] == encoded.n_words.numel():
        dec, part_status = ans_decode_gaussian_checkpointed(encoded, jump, min_symbol, max_symbol, means, stds, out=out)
        return dec, _status_per_stream(part_status)
    return _decode_gaussian("cst_ans_decode_gaussian_batch", True, encoded, min_symbol, max_symbol, means, stds, layout, offsets, out, config)


def range_decode_gaussian(encoded, min_symbol, max_symbol, means, stds, layout="stream_major", offsets=None, out=None, config=None):
    """One RangeDecoder per stream: RangeDecoder(words[s]).decode(QuantizedGaussian(min, max), means[s], stds[s]).  A batch that carries
    jump points for exactly this shape (range_encode_gaussian, jump_points) decodes every part of a stream on a lane of its own."""
    jump = encoded.jump if isinstance(encoded, EncodedBatch) else None
    if isinstance(jump, RangeCheckpoints) and offsets is None and layout == "stream_major" and means.dim() == 2 and \
            tuple(means.shape) == (jump.pos.shape[0], jump.interval * jump.pos.shape[1]) and jump.pos.shape[0] == encoded.n_words.numel():
        n_streams, n_per = means.shape
        means, stds = _gaussian_args(means.shape, means, stds)
        dev = encoded.words.device
        if out is None:
            out = torch.empty((n_streams, n_per), dtype=torch.int32, device=dev)
        part_status = torch.empty(tuple(jump.pos.shape), dtype=torch.int32, device=dev)
        L = N.lib()
        scratch = _ckpt_scratch("range_gaussian_ckpt", dev, L.cst_range_gaussian_ckpt_scratch_bytes(n_streams, n_per, jump.interval))
        N.check(L.cst_range_decode_gaussian_batch_ckpt(_cfg(*encoded.config), int(min_symbol), int(max_symbol), _ptr(encoded.words), None,
                                                       encoded.words.shape[1], encoded.words.numel(), _ptr(encoded.n_words), jump.interval,
                                                       _ptr(jump.pos), _ptr(jump.lower), _ptr(jump.range), _ptr(means), _ptr(stds), _ptr(out),
                                                       n_streams, n_per, _ptr(scratch), _ptr(part_status), _stream_ptr()),
                "cst_range_decode_gaussian_batch_ckpt")
        return out, _status_per_stream(part_status)
    return _decode_gaussian("cst_range_decode_gaussian_batch", False, encoded, min_symbol, max_symbol, means, stds, layout, offsets, out, config)


# ---------------------------------------------------------------------------------------------------------------------
# checkpointed streams (the reference's Pos / Seek jump tables, src/stream/stack.rs:1107-1139): one long stream decodes
# on as many lanes as it has chunks
# ---------------------------------------------------------------------------------------------------------------------

@dataclass
class Checkpoints:
    interval: int
    pos: torch.Tensor        # int32 [n_streams, n_chunks]: words in the bulk in front of chunk j (AnsCoder.pos()[0])
    state: torch.Tensor      # int64 [n_streams, n_chunks]: coder state there (AnsCoder.pos()[1])


def ans_encode_checkpointed(symbols: torch.Tensor, model: Model, interval: int, config=(32, 64, 24), layout="stream_major",
                            stride: Optional[int] = None, out=None):
    """ans_encode + a checkpoint in front of every `interval` symbols.  Returns (EncodedBatch, Checkpoints); the words are
    those of ans_encode.  Models with one table per stream are taken too (stream-major): see DESIGN.md 4.12 (sub-lanes).
    out: (EncodedBatch, Checkpoints) of an earlier call with the same shapes, to code into the same buffers."""
    narrow = _SYMBOL_BYTES.get(symbols.dtype, 4) if symbols.dtype in _SYMBOL_BYTES else 4
    if narrow != 4:        # int8 / int16 matrices (round 5: cst_ans_encode_batch_ckpt_sym; int8 lines are read by the encoder loops themselves)
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only (map the symbols to indices first)")
        symbols = _require_cuda(symbols, symbols.dtype, "symbols")
    else:
        symbols = _to_indices(model, _require_cuda(symbols, torch.int32, "symbols"))
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    dev = symbols.device
    n_chunks = (n_per + interval - 1) // interval
    if out is not None:
        out, ck = out
        stride = out.words.shape[1]
        if ck.interval != int(interval) or tuple(ck.pos.shape) != (n_streams, n_chunks):
            raise ValueError("out: checkpoints of another shape")
    else:
        stride = stride or max_words(n_per, config)
        out = EncodedBatch(torch.empty((n_streams, stride), dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev), tuple(config))
        ck = Checkpoints(int(interval), torch.zeros((n_streams, n_chunks), dtype=torch.int32, device=dev),
                         torch.zeros((n_streams, n_chunks), dtype=torch.int64, device=dev))
    if narrow != 4:
        L = N.lib()
        scratch = _ckpt_scratch("ckpt_widen", dev, L.cst_ckpt_sym_scratch_bytes(n_streams, n_per, int(interval), narrow))
        N.check(L.cst_ans_encode_batch_ckpt_sym(model._h, _cfg(*config), _ptr(symbols), narrow, n_streams, n_per, lay, _ptr(out.words), stride,
                                                _ptr(out.n_words), int(interval), _ptr(ck.pos), _ptr(ck.state), _ptr(out.status), _ptr(scratch),
                                                _stream_ptr()), "cst_ans_encode_batch_ckpt_sym")
        return out, ck
    N.check(N.lib().cst_ans_encode_batch_ckpt(model._h, _cfg(*config), _ptr(symbols), n_streams, n_per, lay, _ptr(out.words), stride,
                                              _ptr(out.n_words), int(interval), _ptr(ck.pos), _ptr(ck.state), _ptr(out.status),
                                              _stream_ptr()), "cst_ans_encode_batch_ckpt")
    return out, ck


def _ckpt_scratch(kind, dev, nbytes):
    """a scratch buffer per purpose, device AND HIP stream (two calls on different streams must not share one: the kernels of
    both may be in flight), grown on demand.  A buffer is only ever used on the stream it was allocated on -- the stream in its
    key -- so the caching allocator's stream-ordered reuse holds when a larger one replaces it."""
    key = (kind, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _scratch[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return buf


def _status_per_stream(part_status: torch.Tensor) -> torch.Tensor:
    """[n_streams, n_chunks] -> [n_streams]: a stream's status is the worst of its chunks' (cst_ckpt_status_per_stream)"""
    n_streams, n_chunks = part_status.shape
    out = torch.empty(n_streams, dtype=torch.int32, device=part_status.device)
    N.check(N.lib().cst_ckpt_status_per_stream(_ptr(part_status), n_streams, n_chunks, _ptr(out), _stream_ptr()), "cst_ckpt_status_per_stream")
    return out


def _check_jump_shape(checkpoints, n_per_stream: int) -> None:
    """The C calls index the jump tables as [n_streams][n_per_stream / interval]: a table with another number of points per stream
    (a prefix decode of a batch encoded with more) would be read with the wrong row stride -- refuse it."""
    if tuple(checkpoints.pos.shape[1:]) != (n_per_stream // checkpoints.interval,) or n_per_stream % checkpoints.interval != 0:
        raise ValueError(f"jump points every {checkpoints.interval} symbols, {checkpoints.pos.shape[1]} per stream, do not describe streams of "
                         f"{n_per_stream} symbols (decode them whole, or without the jump points)")
    for t in vars(checkpoints).values():
        if isinstance(t, torch.Tensor) and tuple(t.shape) != tuple(checkpoints.pos.shape):
            raise ValueError("jump points: tensors of different shapes")


def _decode_jump_composed(encoded: EncodedBatch, ck: Checkpoints, model: Model, n_per_stream: int, out, status, layout: str):
    """AnsCoder.seek + decode of every chunk for the two batch forms cst_ans_decode_batch_ckpt does not take, composed of the plain
    batched decode (raw states, counts = the jump points' word counts):
      packed 16-bit words (stream-major): the n_streams * k chunks are streams of their own at their stream's slab (offsets in
        16-bit words) and rows of the matrix [n_streams * k][interval] -- ONE launch;
      symbol-major: chunk j of all streams is the symbol-major matrix [interval][n_streams] behind row j * interval -- k launches.
    A jump point with more words than its stream holds is caller data gone wrong: CST_STREAM_INVALID_DATA, nothing is read."""
    n_streams, k = ck.pos.shape
    interval = ck.interval
    dev = encoded.words.device
    L = N.lib()
    words = encoded.words
    stride = words.shape[1]
    flags = N.FLAG_RAW_STATE | (N.FLAG_PACKED_W16 if encoded.packed16 else N.FLAG_NONE)
    bad = (ck.pos.view(torch.int32) < 0) | (ck.pos > encoded.n_words.view(-1, 1)) | (ck.pos > stride)
    pos = torch.where(bad, torch.zeros_like(ck.pos), ck.pos)
    if layout == "stream_major":
        v_off = (torch.arange(n_streams, device=dev, dtype=torch.int64) * stride).view(-1, 1).expand(n_streams, k).contiguous()
        v_n, v_state = pos.contiguous().view(-1), ck.state.clone().view(-1)     # (CST_FLAG_RAW_STATE: the state array is in AND out)
        part = torch.empty(n_streams * k, dtype=torch.int32, device=dev)
        N.check(L.cst_ans_decode_batch(model._h, _cfg(*encoded.config), _ptr(words), _ptr(v_off), 0, words.numel(), _ptr(v_n), _ptr(out),
                                       n_streams * k, interval, N.LAYOUT_STREAM_MAJOR, _ptr(v_state), None, _ptr(part), flags, _stream_ptr()),
                "cst_ans_decode_batch")
        status.copy_(part.view(n_streams, k))
    else:
        part = torch.empty(n_streams, dtype=torch.int32, device=dev)
        for j in range(k):
            n_j, st_j = pos[:, j].contiguous(), ck.state[:, j].clone()
            N.check(L.cst_ans_decode_batch(model._h, _cfg(*encoded.config), _ptr(words), None, stride, words.numel(), _ptr(n_j),
                                           _ptr(out[j * interval:(j + 1) * interval]), n_streams, interval, N.LAYOUT_SYMBOL_MAJOR, _ptr(st_j), None,
                                           _ptr(part), flags, _stream_ptr()), "cst_ans_decode_batch")
            status[:, j] = part
    status.masked_fill_(bad, N.STREAM_INVALID_DATA)
    return _to_symbols(model, out), status


def ans_decode_checkpointed(encoded, checkpoints: Checkpoints, model: Model, n_per_stream: int, out=None, status=None,
                            dtype=torch.int32, offsets: Optional[torch.Tensor] = None, config=None, layout="stream_major"):
    """Decodes every chunk on its own lane (AnsCoder.seek(pos, state) + `interval` symbols per chunk).
    Returns (symbols [n_streams, n_per_stream], status [n_streams, n_chunks]).  dtype (or the dtype of `out`): int32, or int16 / int8 for
    a narrow symbol matrix (cst_ans_decode_batch_ckpt_sym: int8 chunks of whole 128-symbol lines are written by the decoder loops).
    `encoded`: an EncodedBatch (slabs), or the PACKED words of compact() / container.load() / a gather with `offsets` (int64
    [n_streams + 1]) and `config` -- jump points count words from the start of their stream, wherever the stream lies."""
    if not isinstance(encoded, EncodedBatch):
        if offsets is None:
            raise ValueError("packed words need their offsets")
        words = encoded[0] if isinstance(encoded, (tuple, list)) else encoded
        encoded = EncodedBatch(words.view(1, -1), checkpoints.pos[:, 0], checkpoints.pos[:, 0], tuple(config or (32, 64, 12)))
    n_streams = checkpoints.pos.shape[0]
    stride_arg = 0 if offsets is not None else encoded.words.shape[1]
    dev = encoded.words.device
    n_chunks = checkpoints.pos.shape[1]
    _check_jump_shape(checkpoints, n_per_stream)
    if layout not in ("stream_major", "symbol_major"):
        raise ValueError("layout")
    if out is None:
        out = torch.empty((n_streams, n_per_stream) if layout == "stream_major" else (n_per_stream, n_streams), dtype=dtype, device=dev)
    if status is None:
        status = torch.empty((n_streams, n_chunks), dtype=torch.int32, device=dev)
    L = N.lib()
    narrow = _SYMBOL_BYTES.get(out.dtype)
    if narrow is None:
        raise TypeError("decoded symbols are int32, int16 or int8")
    if encoded.packed16 or layout == "symbol_major":
        if offsets is not None or narrow != 4:
            raise ValueError("jump points of packed 16-bit or symbol-major batches: slabs and int32 symbols")
        return _decode_jump_composed(encoded, checkpoints, model, n_per_stream, out, status, layout)
    if narrow != 4:
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only")
        scratch = _ckpt_scratch("ans_ckpt_sym", dev,
                                L.cst_ckpt_sym_scratch_bytes(n_streams, n_per_stream, checkpoints.interval, narrow))
        N.check(L.cst_ans_decode_batch_ckpt_sym(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(offsets), stride_arg,
                                                encoded.words.numel(), checkpoints.interval, _ptr(checkpoints.pos), _ptr(checkpoints.state), _ptr(out),
                                                narrow, n_streams, n_per_stream, _ptr(scratch), _ptr(status), _stream_ptr()),
                "cst_ans_decode_batch_ckpt_sym")
        return out, status
    scratch = _ckpt_scratch("ans_ckpt", dev, L.cst_ckpt_scratch_bytes(n_streams, n_per_stream, checkpoints.interval))
    N.check(L.cst_ans_decode_batch_ckpt(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(offsets), stride_arg,
                                        encoded.words.numel(), checkpoints.interval, _ptr(checkpoints.pos), _ptr(checkpoints.state), _ptr(out), n_streams,
                                        n_per_stream, _ptr(scratch), _ptr(status), _stream_ptr()), "cst_ans_decode_batch_ckpt")
    return _to_symbols(model, out), status


@dataclass
class RangeCheckpoints:
    """RangeEncoder.pos() in front of every chunk (queue.rs:182-196): words emitted so far incl. held-back ones, and the state"""
    interval: int
    pos: torch.Tensor        # int32 [n_streams, n_chunks]
    lower: torch.Tensor      # int64 [n_streams, n_chunks]  (uint64 values)
    range: torch.Tensor      # int64 [n_streams, n_chunks]


def range_encode_checkpointed(symbols: torch.Tensor, model: Model, interval: int, config=(32, 64, 12), layout="stream_major",
                              stride: Optional[int] = None, out=None):
    """range_encode + a jump point in front of every `interval` symbols.  Returns (EncodedBatch, RangeCheckpoints); the words
    are those of range_encode."""
    narrow = _SYMBOL_BYTES.get(symbols.dtype, 4) if symbols.dtype in _SYMBOL_BYTES else 4
    if narrow != 4:        # int8 / int16 matrices (round 6: cst_range_encode_batch_ckpt_sym; int8 tiles are read by the encoder loop itself)
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only (map the symbols to indices first)")
        symbols = _require_cuda(symbols, symbols.dtype, "symbols")
    else:
        symbols = _to_indices(model, _require_cuda(symbols, torch.int32, "symbols"))
    n_streams, n_per, lay = _layout_shape(symbols, layout)
    dev = symbols.device
    n_chunks = (n_per + interval - 1) // interval
    if out is not None:
        out, ck = out
        stride = out.words.shape[1]
        if ck.interval != int(interval) or tuple(ck.pos.shape) != (n_streams, n_chunks):
            raise ValueError("out: checkpoints of another shape")
    else:
        stride = stride or range_max_words(n_per, config)
        out = EncodedBatch(torch.empty((n_streams, stride), dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev), tuple(config))
        ck = RangeCheckpoints(int(interval), *(torch.zeros((n_streams, n_chunks), dtype=dt, device=dev)
                                               for dt in (torch.int32, torch.int64, torch.int64)))
    if narrow != 4:
        L = N.lib()
        scratch = _ckpt_scratch("range_sym", dev, L.cst_range_sym_scratch_bytes(n_streams, n_per, int(interval), narrow))
        N.check(L.cst_range_encode_batch_ckpt_sym(model._h, _cfg(*config), _ptr(symbols), narrow, n_streams, n_per, lay, _ptr(out.words), stride,
                                                  _ptr(out.n_words), int(interval), _ptr(ck.pos), _ptr(ck.lower), _ptr(ck.range),
                                                  _ptr(out.status), _ptr(scratch), _stream_ptr()), "cst_range_encode_batch_ckpt_sym")
        return out, ck
    N.check(N.lib().cst_range_encode_batch_ckpt(model._h, _cfg(*config), _ptr(symbols), n_streams, n_per, lay, _ptr(out.words), stride,
                                                _ptr(out.n_words), int(interval), _ptr(ck.pos), _ptr(ck.lower), _ptr(ck.range),
                                                _ptr(out.status), _stream_ptr()), "cst_range_encode_batch_ckpt")
    return out, ck


def range_decode_checkpointed(encoded, checkpoints: RangeCheckpoints, model: Model, n_per_stream: int, out=None, status=None,
                              offsets: Optional[torch.Tensor] = None, config=None, dtype=torch.int32):
    """Every chunk on its own lane: RangeDecoder.seek(pos, (lower, range)) + `interval` symbols per chunk.
    Returns (symbols [n_streams, n_per_stream], status [n_streams, n_chunks]).  `encoded`: an EncodedBatch, or (packed words, n_words)
    with `offsets` (int64 [n_streams + 1]) and `config` for the words of compact() / container.load() / a gather."""
    stride_arg = None
    if not isinstance(encoded, EncodedBatch):
        if offsets is None:
            raise ValueError("packed words need their offsets")
        words, n_words = encoded
        encoded = EncodedBatch(words.view(1, -1), n_words, n_words, tuple(config or (32, 64, 12)))
        stride_arg = 0
    n_streams = encoded.n_words.numel()
    dev = encoded.words.device
    n_chunks = checkpoints.pos.shape[1]
    _check_jump_shape(checkpoints, n_per_stream)
    if checkpoints.pos.shape[0] != n_streams:
        raise ValueError("jump points of another batch: one row per stream")
    if out is None:
        out = torch.empty((n_streams, n_per_stream), dtype=dtype, device=dev)
    if status is None:
        status = torch.empty((n_streams, n_chunks), dtype=torch.int32, device=dev)
    L = N.lib()
    narrow = _SYMBOL_BYTES.get(out.dtype)
    if narrow is None:
        raise TypeError("decoded symbols are int32, int16 or int8")
    if narrow != 4:        # (round 6: int8 matrices are written by the sub-lane decoder itself, cst_range_decode_batch_ckpt_sym)
        if model.noncontiguous:
            raise ValueError("narrow symbol matrices: contiguous alphabets only")
        scratch = _ckpt_scratch("range_sym", dev, L.cst_range_sym_scratch_bytes(n_streams, n_per_stream, checkpoints.interval, narrow))
        N.check(L.cst_range_decode_batch_ckpt_sym(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(offsets),
                                                  encoded.words.shape[1] if stride_arg is None else stride_arg, encoded.words.numel(), _ptr(encoded.n_words),
                                                  checkpoints.interval, _ptr(checkpoints.pos), _ptr(checkpoints.lower), _ptr(checkpoints.range), _ptr(out),
                                                  narrow, n_streams, n_per_stream, _ptr(scratch), _ptr(status), _stream_ptr()),
                "cst_range_decode_batch_ckpt_sym")
        return out, status
    scratch = _ckpt_scratch("range_ckpt", dev, L.cst_range_ckpt_scratch_bytes(n_streams, n_per_stream, checkpoints.interval))
    N.check(L.cst_range_decode_batch_ckpt(model._h, _cfg(*encoded.config), _ptr(encoded.words), _ptr(offsets),
                                          encoded.words.shape[1] if stride_arg is None else stride_arg, encoded.words.numel(), _ptr(encoded.n_words), checkpoints.interval, _ptr(checkpoints.pos),
                                          _ptr(checkpoints.lower), _ptr(checkpoints.range), _ptr(out), n_streams, n_per_stream,
                                          _ptr(scratch), _ptr(status), _stream_ptr()), "cst_range_decode_batch_ckpt")
    return _to_symbols(model, out), status


def ans_encode_gaussian_checkpointed(symbols, min_symbol, max_symbol, means, stds, interval: int, config=(32, 64, 24),
                                     stride: Optional[int] = None, out=None):
    """ans_encode_gaussian + a jump point in front of every `interval` symbols (a multiple of 16 that divides the row length;
    batches of at least 16 384 streams, stream-major).  Returns (EncodedBatch, Checkpoints); the words are ans_encode_gaussian's."""
    symbols = _require_cuda(symbols, torch.int32, "symbols")
    n_streams, n_per, lay = _layout_shape(symbols, "stream_major")
    means, stds = _gaussian_args(symbols.shape, means, stds)
    dev = symbols.device
    n_chunks = (n_per + interval - 1) // interval
    if out is not None:
        out, ck = out
    else:
        stride = stride or max_words(n_per, config)
        out = EncodedBatch(torch.empty((n_streams, stride), dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev),
                           torch.empty(n_streams, dtype=torch.int32, device=dev), tuple(config))
        ck = Checkpoints(int(interval), torch.zeros((n_streams, n_chunks), dtype=torch.int32, device=dev),
                         torch.zeros((n_streams, n_chunks), dtype=torch.int64, device=dev))
    N.check(N.lib().cst_ans_encode_gaussian_batch_ckpt(_cfg(*config), int(min_symbol), int(max_symbol), _ptr(symbols), _ptr(means), _ptr(stds),
                                                       n_streams, n_per, lay, _ptr(out.words), out.words.shape[1], _ptr(out.n_words),
                                                       int(interval), _ptr(ck.pos), _ptr(ck.state), _ptr(out.status), _stream_ptr()),
            "cst_ans_encode_gaussian_batch_ckpt")
    return out, ck


def ans_decode_gaussian_checkpointed(encoded: EncodedBatch, checkpoints: Checkpoints, min_symbol, max_symbol, means, stds, out=None, status=None):
    """Every chunk on its own lane: AnsCoder.seek(pos, state) + decode(QuantizedGaussian(min, max), means[chunk], stds[chunk]).
    Returns (symbols [n_streams, n_per_stream], status [n_streams, n_chunks])."""
    if means.dim() != 2:
        raise ValueError("means must be 2-d")
    n_streams, n_per = means.shape
    means, stds = _gaussian_args(means.shape, means, stds)
    dev = encoded.words.device
    n_chunks = checkpoints.pos.shape[1]
    _check_jump_shape(checkpoints, n_per)
    if checkpoints.pos.shape[0] != n_streams:
        raise ValueError("jump points of another batch: one row per stream")
    if out is None:
        out = torch.empty((n_streams, n_per), dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty((n_streams, n_chunks), dtype=torch.int32, device=dev)
    L = N.lib()
    scratch = _ckpt_scratch("ans_ckpt", dev, L.cst_ckpt_scratch_bytes(n_streams, n_per, checkpoints.interval))
    N.check(L.cst_ans_decode_gaussian_batch_ckpt(_cfg(*encoded.config), int(min_symbol), int(max_symbol), _ptr(encoded.words), None,
                                                 encoded.words.shape[1], encoded.words.numel(), checkpoints.interval, _ptr(checkpoints.pos),
                                                 _ptr(checkpoints.state), _ptr(means), _ptr(stds), _ptr(out), n_streams, n_per, _ptr(scratch),
                                                 _ptr(status), _stream_ptr()), "cst_ans_decode_gaussian_batch_ckpt")
    return out, status
