"""One rank of the multi-process exchange tests (tests/test_gpu_rccl_multirank.py): encodes its block of streams on the ONE
GPU of the box, compacts, gathers to `root` through the library's own communicator (cst_gather_sizes_rccl + cst_gather_rccl),
gets its words back (cst_scatter_rccl), decodes from the scattered buffer.  The transport is tests/rccl_double/fake_rccl.cpp
(CST_RCCL_LIB); everything else is the product path.  Writes a JSON verdict to `out`."""
import ctypes
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def main():
    spec = json.loads(sys.argv[1])
    rank, counts, root, n_per, P = spec["rank"], spec["counts"], spec["root"], spec["n_per"], spec["P"]
    world = len(counts)
    verdict = {"rank": rank, "ok": False}
    out = Path(spec["out"])
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        from constriction_amd import _native as N
        from constriction_amd import batched as B
        from constriction_amd import dist as D
        from oracle import oracle as O

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method="file://" + spec["rendezvous"], rank=rank, world_size=world)
        double = ctypes.CDLL(os.environ["CST_RCCL_LIB"])          # the same loaded object the library opened: its counters
        lo, hi = -20, 20
        cdf = O.GaussianModel(lo, hi, 1.5, 4.0, P, 32).cdf_table()
        model = B.Model.from_cdf(cdf, lo, P)
        first = sum(counts[:rank])
        sym_all = O.synth_symbols(11, 0, sum(counts), n_per, lo, cdf, P)
        sym = sym_all[first: first + counts[rank]]
        if counts[rank]:
            enc = B.ans_encode(torch.from_numpy(np.ascontiguousarray(sym)).cuda(), model, (32, 64, P))
            packed, offsets = B.compact(enc)
            n_words = enc.n_words
        else:                                                      # a rank without streams takes part with empty buffers
            packed = torch.zeros(1, dtype=torch.int32, device="cuda")
            offsets = torch.zeros(1, dtype=torch.int64, device="cuda")
            n_words = torch.zeros(0, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        comm = D.RcclComm()
        try:
            got = comm.gather_packed(packed, offsets, dst=root)
        except N.BackendError as e:
            verdict.update(error=str(e), group_depth=double.fake_rccl_group_depth(), group_ends=double.fake_rccl_group_ends())
            out.write_text(json.dumps(verdict))
            return
        torch.cuda.synchronize()
        verdict["sizes"] = comm.last_sizes.tolist()
        if rank == root:
            all_packed, all_off = got
            want_words, want_n, _ = O.ans_encode_batch(sym_all, lo, cdf, P)
            want_off = np.concatenate([[0], np.cumsum(want_n.astype(np.int64))])
            assert all_off.cpu().numpy().tolist() == want_off.tolist(), "global offsets differ from the oracle's word counts"
            pk = all_packed.cpu().numpy().view(np.uint32)
            assert len(pk) == want_off[-1]
            for s in range(len(want_n)):
                assert pk[want_off[s]: want_off[s + 1]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s} differs from the oracle"
            verdict["gathered_words"] = int(want_off[-1])
        else:
            assert got is None
            all_packed = all_off = None
        back_packed, back_off = comm.scatter_packed(all_packed, all_off, src=root)
        torch.cuda.synchronize()
        total = int(offsets[-1].item())
        assert torch.equal(back_off, offsets), "scattered offsets are not the rank's own"
        assert torch.equal(back_packed, packed[:total]), "scattered words are not the rank's own"
        if counts[rank]:
            dec, st = B.ans_decode((back_packed, n_words), model, n_per, offsets=back_off, config=(32, 64, P))
            torch.cuda.synchronize()
            assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym), "decode from the scattered buffer"
        verdict.update(ok=True, group_depth=double.fake_rccl_group_depth(), group_ends=double.fake_rccl_group_ends(),
                       sends=double.fake_rccl_sends())
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:                                     # noqa: BLE001 -- the verdict carries it to the test
        import traceback
        verdict["exception"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]
    out.write_text(json.dumps(verdict))


if __name__ == "__main__":
    main()
