"""Multi-GPU: streams shard trivially (one process per GPU, contiguous blocks of streams, shared tables
replicated); the only exchange step is the gather of the per-stream compressed words to one rank
(BASELINE config C5).  RCCL has no allgatherv, so the gather is
  (1) one small all_gather of (n_streams, total_words) per rank -> displacements, then
  (2) grouped point-to-point send/recv of the packed words and of the per-stream lengths straight into
      their final position on the destination (each peer uses its own xGMI link to the root).
Works with the "nccl" (= RCCL) backend on device tensors and with "gloo" on host tensors (CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of streams owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_packed(packed: torch.Tensor, offsets: torch.Tensor, dst: int = 0, group=None
                  ) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Gathers every rank's packed words (int32 storage of uint32 words) and per-stream offsets to `dst`.

    packed  : [total_words_local]           offsets : int64 [n_streams_local + 1] (offsets[-1] == total)
    Returns on dst (all_packed, all_offsets) in rank order with global offsets; None on the other ranks.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = offsets.device
    n_local = offsets.numel() - 1
    lengths = (offsets[1:] - offsets[:-1]).contiguous()
    meta = torch.stack([torch.tensor(n_local, dtype=torch.int64, device=dev), offsets[-1].to(torch.int64)])
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu()
    n_streams = metas[:, 0].tolist()
    n_words = metas[:, 1].tolist()

    if rank == dst:
        all_packed = torch.empty(max(sum(n_words), 1), dtype=packed.dtype, device=dev)
        all_len = torch.empty(sum(n_streams), dtype=torch.int64, device=dev)
        ops, wpos, spos = [], 0, 0
        for r in range(world):
            pw, pl = all_packed[wpos: wpos + n_words[r]], all_len[spos: spos + n_streams[r]]
            if r == rank:
                pw.copy_(packed[: n_words[r]])
                pl.copy_(lengths)
            else:
                if n_words[r]:
                    ops.append(dist.P2POp(dist.irecv, pw, r, group))
                if n_streams[r]:
                    ops.append(dist.P2POp(dist.irecv, pl, r, group))
            wpos += n_words[r]
            spos += n_streams[r]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        all_off = torch.zeros(sum(n_streams) + 1, dtype=torch.int64, device=dev)
        torch.cumsum(all_len, 0, out=all_off[1:])
        return all_packed[: sum(n_words)], all_off
    ops = []
    if n_words[rank]:
        ops.append(dist.P2POp(dist.isend, packed[: n_words[rank]].contiguous(), dst, group))
    if n_local:
        ops.append(dist.P2POp(dist.isend, lengths, dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return None


def scatter_packed(all_packed: Optional[torch.Tensor], all_offsets: Optional[torch.Tensor], n_local: int, src: int = 0,
                   group=None, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of gather_packed: `src` hands every rank the words of its block of streams
    (blocks as in shard_range over the gathered stream order).  Returns (packed_local, offsets_local)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = torch.tensor([n_local], dtype=torch.int64, device=device)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = [int(c.item()) for c in all_counts]
    starts = [sum(all_counts[:r]) for r in range(world)]
    if rank == src:
        ops, mine = [], None
        for r in range(world):
            a, b = starts[r], starts[r] + all_counts[r]
            off = all_offsets[a: b + 1]
            lens = (off[1:] - off[:-1]).contiguous()
            words = all_packed[int(off[0].item()): int(off[-1].item())].contiguous()
            if r == rank:
                mine = (words, lens)
            else:
                if all_counts[r]:
                    ops.append(dist.P2POp(dist.isend, lens, r, group))
                    if words.numel():
                        ops.append(dist.P2POp(dist.isend, words, r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        words, lens = mine
    else:
        lens = torch.empty(n_local, dtype=torch.int64, device=device)
        if n_local:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, lens, src, group)]):
                req.wait()
        total = int(lens.sum().item()) if n_local else 0
        words = torch.empty(max(total, 1), dtype=torch.int32, device=device)[:total]
        if total:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, words, src, group)]):
                req.wait()
    off = torch.zeros(n_local + 1, dtype=torch.int64, device=lens.device)
    torch.cumsum(lens, 0, out=off[1:])
    return words, off


# ---------------------------------------------------------------------------------------------------------------------
# the same gather through the C ABI (cst_gather_sizes_rccl + cst_gather_rccl: what a Rust / C caller of the library uses)
# ---------------------------------------------------------------------------------------------------------------------

class RcclComm:
    """An RCCL communicator owned by the coder library (include/constriction_amd.h, "multi-GPU").  The 128-byte unique id
    is created on rank 0 and handed to the other ranks through torch.distributed's existing process group."""

    def __init__(self, group=None):
        import ctypes as C
        import numpy as np
        from . import _native as N
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        lib = N.lib()
        idbuf = np.zeros(128, dtype=np.uint8)
        if self.rank == 0:
            N.check(lib.cst_rccl_get_unique_id(idbuf.ctypes.data), "cst_rccl_get_unique_id")
        if self.world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.from_numpy(idbuf).to(dev)
            dist.broadcast(t, src=0, group=group)
            idbuf = t.cpu().numpy().copy()
        h = C.c_void_p()
        N.check(lib.cst_rccl_comm_init(idbuf.ctypes.data, self.world, self.rank, C.byref(h)), "cst_rccl_comm_init")
        self._h = h

    def close(self):
        from . import _native as N
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.load_library().cst_rccl_comm_destroy(h)

    def gather_packed(self, packed: torch.Tensor, offsets: torch.Tensor, dst: int = 0):
        """packed: int32 storage of uint32 words (device), offsets: int64 [n_streams_local + 1] (device) as returned by
        batched.compact.  Returns (all_packed, all_offsets) on dst, None elsewhere."""
        import ctypes as C
        from . import _native as N
        lib = N.lib()
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        n_local = offsets.numel() - 1
        d_sizes = torch.zeros(2 * self.world, dtype=torch.int64, device=offsets.device)
        N.check(lib.cst_gather_sizes_rccl(self._h, self.world, self.rank, C.c_void_p(offsets.data_ptr()), n_local,
                                          C.c_void_p(d_sizes.data_ptr()), sp), "cst_gather_sizes_rccl")
        h_sizes = d_sizes.cpu().numpy().astype("uint64")                       # (synchronises the stream)
        all_packed = all_off = None
        if self.rank == dst:
            all_packed = torch.empty(max(int(h_sizes[1::2].sum()), 1), dtype=torch.int32, device=offsets.device)
            all_off = torch.empty(int(h_sizes[0::2].sum()) + 1, dtype=torch.int64, device=offsets.device)
        N.check(lib.cst_gather_rccl(self._h, self.world, self.rank, dst, C.c_void_p(packed.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                    h_sizes.ctypes.data, C.c_void_p(all_packed.data_ptr()) if all_packed is not None else None,
                                    C.c_void_p(all_off.data_ptr()) if all_off is not None else None, sp), "cst_gather_rccl")
        self.last_sizes = h_sizes                                              # (n_streams, n_words) per rank, for scatter_packed
        if self.rank == dst:
            return all_packed[: int(h_sizes[1::2].sum())], all_off
        return None

    def scatter_packed(self, all_packed, all_offsets, h_sizes=None, src: int = 0, device=None):
        """Inverse of gather_packed (cst_scatter_rccl): `src` hands every rank the words of its own streams and their offsets
        rebased to 0.  h_sizes: (n_streams, n_words) per rank as a flat uint64 array (default: those of the last gather).
        Returns (packed_local, offsets_local) on every rank."""
        import ctypes as C
        import numpy as np
        from . import _native as N
        lib = N.lib()
        h_sizes = np.ascontiguousarray(self.last_sizes if h_sizes is None else h_sizes, dtype=np.uint64)
        dev = device or (all_offsets.device if all_offsets is not None else torch.device("cuda", torch.cuda.current_device()))
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        n_local, w_local = int(h_sizes[2 * self.rank]), int(h_sizes[2 * self.rank + 1])
        packed = torch.empty(max(w_local, 1), dtype=torch.int32, device=dev)
        offsets = torch.empty(n_local + 1, dtype=torch.int64, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        N.check(lib.cst_scatter_rccl(self._h, self.world, self.rank, src, p(all_packed), p(all_offsets), h_sizes.ctypes.data,
                                     p(packed), p(offsets), sp), "cst_scatter_rccl")
        return packed[:w_local], offsets
