"""Wire format of a packed batch: the counterpart of the reference's `compressed.tofile(...)` / `np.fromfile(...)`
(src/pybindings/stream/stack.rs:149-166, 378-396) for MANY streams.

A file holds what `batched.compact` produces -- the concatenation of every stream's `get_compressed()` words and the
per-stream offsets -- in little-endian byte order whatever the host's (the reference's doc examples byteswap on big-endian
hosts for the same reason):

    bytes  0.. 7   magic  b"CSTPACK1"
    bytes  8..15   n_streams          (u64)
    bytes 16..23   total_words        (u64)  = offsets[n_streams]
    bytes 24..27   word_bits, 28..31 state_bits, 32..35 precision   (u32 each: the coder preset the words belong to)
    bytes 36..39   reserved (0)
    then           offsets[n_streams + 1]   (u64)
    then           words[total_words]       (u32; 16-bit presets keep one word per u32 as everywhere in this library)

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


def save(path, packed, offsets, config: Tuple[int, int, int]) -> None:
    """packed: the words of all streams back to back (uint32 / int32 array or tensor), offsets: n_streams + 1 positions."""
    packed = np.ascontiguousarray(_to_numpy(packed)).view(np.uint32).ravel()
    offsets = np.ascontiguousarray(_to_numpy(offsets)).astype(np.uint64).ravel()
    if len(offsets) < 1 or int(offsets[0]) != 0 or np.any(np.diff(offsets.astype(np.int64)) < 0):
        raise ValueError("offsets must start at 0 and never decrease")
    total = int(offsets[-1])
    if total > len(packed):
        raise ValueError("offsets run past the packed words")
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, len(offsets) - 1, total, int(config[0]), int(config[1]), int(config[2]), 0))
        f.write(offsets.astype("<u8").tobytes())
        f.write(packed[:total].astype("<u4").tobytes())


def load(path):
    """-> (words uint32[total], offsets uint64[n_streams + 1], (word_bits, state_bits, precision)), native byte order"""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise ValueError("not a packed-batch container (file too short)")
        magic, n_streams, total, w, s, p, _ = _HEADER.unpack(head)
        if magic != MAGIC:
            raise ValueError("not a packed-batch container (bad magic)")
        # the header is untrusted: sizes are checked against the file BEFORE anything is read or allocated
        f.seek(0, 2)
        if _HEADER.size + 8 * (n_streams + 1) + 4 * total != f.tell():
            raise ValueError("truncated or inconsistent packed-batch container (sizes in the header do not match the file)")
        if (w, s) not in ((32, 64), (16, 32)) or not 1 <= p <= (24 if w == 32 else 16):
            raise ValueError(f"packed-batch container for an unsupported coder preset ({w}, {s}, {p})")
        f.seek(_HEADER.size)
        offsets = np.frombuffer(f.read(8 * (n_streams + 1)), dtype="<u8")
        words = np.frombuffer(f.read(4 * total), dtype="<u4")
    # ... and the offsets go straight into the `offsets=` form of the GPU decoders: they must start at 0, never decrease
    # and end at the number of words (what `save` enforces)
    if (len(offsets) != n_streams + 1 or len(words) != total or int(offsets[0]) != 0 or int(offsets[-1]) != total
            or np.any(offsets[1:] < offsets[:-1])):
        raise ValueError("truncated or inconsistent packed-batch container")
    return words.astype(np.uint32), offsets.astype(np.uint64), (int(w), int(s), int(p))


def _to_numpy(x):
    if hasattr(x, "detach"):          # a torch tensor (device or host)
        return x.detach().cpu().numpy()
    return np.asarray(x)
