"""Wire format of a packed batch: the counterpart of the reference's `compressed.tofile(...)` / `np.fromfile(...)`
(src/pybindings/stream/stack.rs:149-166, 378-396) for MANY streams.

A file holds what `batched.compact` produces -- the concatenation of every stream's `get_compressed()` words and the
per-stream offsets -- in little-endian byte order whatever the host's (the reference's doc examples byteswap on big-endian
hosts for the same reason):

    bytes  0.. 7   magic  b"CSTPACK1"
    bytes  8..15   n_streams          (u64)
    bytes 16..23   total_words        (u64)  = offsets[n_streams]
    bytes 24..27   word_bits, 28..31 state_bits, 32..35 precision   (u32 each: the coder preset the words belong to)
    bytes 36..39   n_chunks: jump points per stream (u32; 0 = none: the field was "reserved (0)" before round 5)
    then           offsets[n_streams + 1]   (u64)
    then           words[total_words]       (u32; 16-bit presets keep one word per u32 as everywhere in this library)
    then, if n_chunks > 0, the batch's JUMP TABLE (the reference's `Pos` / `Seek` side information, stack.rs:1107-1139: `AnsCoder::pos()`
    in front of every chunk of `interval` symbols -- batched.ans_encode_checkpointed):
                   interval                           (u64)
                   pos[n_streams][n_chunks]           (u32: words of the stream below the jump point)
                   state[n_streams][n_chunks]         (u64: the coder state there)
    A reader that ignores the table (`load`) loses only speed: the words are the plain encoder's.

Stream s is `words[offsets[s] : offsets[s + 1]]`, bit for bit the array one reference coder would have written with
`tofile` for that stream; `load` returns numpy arrays, ready for `torch.from_numpy(...).cuda()` and the `offsets=` form
of `batched.ans_decode` / `range_decode`.  No GPU is needed to read or write a container.
"""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

MAGIC = b"CSTPACK1"
_HEADER = struct.Struct("<8sQQIIII")


def save(path, packed, offsets, config: Tuple[int, int, int], jump_points=None) -> None:
    """packed: the words of all streams back to back (uint32 / int32 array or tensor), offsets: n_streams + 1 positions.
    jump_points: a batched.Checkpoints (or anything with .interval, .pos [n_streams, n_chunks], .state) to store behind the words."""
    packed = np.ascontiguousarray(_to_numpy(packed)).view(np.uint32).ravel()
    offsets = np.ascontiguousarray(_to_numpy(offsets)).astype(np.uint64).ravel()
    if len(offsets) < 1 or int(offsets[0]) != 0 or np.any(np.diff(offsets.astype(np.int64)) < 0):
        raise ValueError("offsets must start at 0 and never decrease")
    total = int(offsets[-1])
    if total > len(packed):
        raise ValueError("offsets run past the packed words")
    n_chunks, pos, state = 0, None, None
    if jump_points is not None:
        pos = np.ascontiguousarray(_to_numpy(jump_points.pos)).view(np.uint32)
        state = np.ascontiguousarray(_to_numpy(jump_points.state)).view(np.uint64)
        if pos.ndim != 2 or pos.shape != state.shape or pos.shape[0] != len(offsets) - 1 or pos.shape[1] < 1 or int(jump_points.interval) < 1:
            raise ValueError("jump points: pos and state are [n_streams, n_chunks] arrays, interval >= 1")
        n_chunks = pos.shape[1]
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, len(offsets) - 1, total, int(config[0]), int(config[1]), int(config[2]), n_chunks))
        f.write(offsets.astype("<u8").tobytes())
        f.write(packed[:total].astype("<u4").tobytes())
        if n_chunks:
            f.write(struct.pack("<Q", int(jump_points.interval)))
            f.write(pos.astype("<u4").tobytes())
            f.write(state.astype("<u8").tobytes())


def load(path):
    """-> (words uint32[total], offsets uint64[n_streams + 1], (word_bits, state_bits, precision)), native byte order
    (a jump table in the file is checked and skipped: load_with_jump_points returns it)"""
    return load_with_jump_points(path)[:3]


def load_with_jump_points(path):
    """-> (words, offsets, config, jump) with jump = None or (interval, pos uint32[n_streams, n_chunks], state uint64[n_streams, n_chunks]):
    `batched.Checkpoints(interval, pos, state)` of device tensors is what `batched.ans_decode_checkpointed(words, ..., offsets=...)` takes"""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise ValueError("not a packed-batch container (file too short)")
        magic, n_streams, total, w, s, p, n_chunks = _HEADER.unpack(head)
        if magic != MAGIC:
            raise ValueError("not a packed-batch container (bad magic)")
        # the header is untrusted: sizes are checked against the file BEFORE anything is read or allocated
        f.seek(0, 2)
        jump_bytes = 8 + 12 * n_streams * n_chunks if n_chunks else 0
        if _HEADER.size + 8 * (n_streams + 1) + 4 * total + jump_bytes != f.tell():
            raise ValueError("truncated or inconsistent packed-batch container (sizes in the header do not match the file)")
        if (w, s) not in ((32, 64), (16, 32)) or not 1 <= p <= (24 if w == 32 else 16):
            raise ValueError(f"packed-batch container for an unsupported coder preset ({w}, {s}, {p})")
        f.seek(_HEADER.size)
        offsets = np.frombuffer(f.read(8 * (n_streams + 1)), dtype="<u8")
        words = np.frombuffer(f.read(4 * total), dtype="<u4")
        jump = None
        if n_chunks:
            (interval,) = struct.unpack("<Q", f.read(8))
            pos = np.frombuffer(f.read(4 * n_streams * n_chunks), dtype="<u4").reshape(n_streams, n_chunks)
            state = np.frombuffer(f.read(8 * n_streams * n_chunks), dtype="<u8").reshape(n_streams, n_chunks)
            if interval < 1:
                raise ValueError("truncated or inconsistent packed-batch container (jump table without an interval)")
            jump = (int(interval), pos.astype(np.uint32), state.astype(np.uint64))
    # ... and the offsets go straight into the `offsets=` form of the GPU decoders: they must start at 0, never decrease
    # and end at the number of words (what `save` enforces)
    if (len(offsets) != n_streams + 1 or len(words) != total or int(offsets[0]) != 0 or int(offsets[-1]) != total
            or np.any(offsets[1:] < offsets[:-1])):
        raise ValueError("truncated or inconsistent packed-batch container")
    # a jump point counts words from the start of ITS stream: one beyond the stream's words would make the packed decoders read the
    # neighbour's (they check a point against the whole buffer and against the stream's first point, not against the stream's end)
    if jump is not None and np.any(jump[1].astype(np.uint64) > np.diff(offsets.astype(np.uint64))[:, None]):
        raise ValueError("inconsistent packed-batch container (a jump point beyond its stream's words)")
    return words.astype(np.uint32), offsets.astype(np.uint64), (int(w), int(s), int(p)), jump


def _to_numpy(x):
    if hasattr(x, "detach"):          # a torch tensor (device or host)
        return x.detach().cpu().numpy()
    return np.asarray(x)
