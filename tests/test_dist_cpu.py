"""CPU-only, world_size 2 over gloo: the multi-GPU plumbing (stream sharding + variable-length gather/scatter of the
packed compressed words, constriction_amd/dist.py).  The same code runs over RCCL ("nccl") on device tensors."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_shard(rank, n_streams_total, world):
    """Deterministic fake 'compressed words' per stream (content = f(global stream id))."""
    from constriction_amd.dist import shard_range
    a, b = shard_range(n_streams_total, rank, world)
    lens = [(7 * s + 3) % 11 for s in range(a, b)]          # some streams are empty
    words = np.concatenate([np.arange(l, dtype=np.int64) + 1000 * s for s, l in zip(range(a, b), lens)] + [np.zeros(0, np.int64)])
    off = np.zeros(b - a + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    return a, b, torch.from_numpy(words.astype(np.int32)), torch.from_numpy(off)


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from constriction_amd.dist import gather_packed, scatter_packed, shard_range
        a, b, packed, off = _fake_shard(rank, n_total, world)
        res = gather_packed(packed, off, dst=0)
        ok = True
        if rank == 0:
            all_packed, all_off = res
            exp_words, exp_off = [], [0]
            for r in range(world):
                _, _, p, o = _fake_shard(r, n_total, world)
                exp_words.append(p.numpy())
                exp_off.extend((o.numpy()[1:] + exp_off[-1] - 0).tolist() if False else (exp_off[-1] + o.numpy()[1:]).tolist())
            ok = np.array_equal(all_packed.numpy(), np.concatenate(exp_words)) and all_off.numpy().tolist() == exp_off
        else:
            ok = res is None
            all_packed = all_off = None
        # and back: every rank gets exactly its own words again
        words, off2 = scatter_packed(all_packed, all_off, b - a, src=0)
        ok = ok and np.array_equal(words.numpy(), packed.numpy()) and off2.numpy().tolist() == off.numpy().tolist()
        # shard_range is a partition
        parts = [shard_range(n_total, r, world) for r in range(world)]
        ok = ok and parts[0][0] == 0 and parts[-1][1] == n_total and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [0, 1, 5, 64])
def test_gather_scatter_world2(n_total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.parametrize("world", [1, 2])
def test_bench_starts_its_own_ranks(world):
    """`python bench.py --gpus N` re-executes itself under torch.distributed.run (127.0.0.1) and runs its own sharding /
    gather / reporting path; here on host tensors over gloo (--plumbing), on the GPU box over RCCL."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(world), "--plumbing", "--backend", "gloo", "--streams", "777"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=str(root))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["plumbing"] == "ok" and out["n_gpus"] == world and out["streams_per_rank"] == 777
    want = sum((7 * s + 3) % 11 for s in range(world * 777))
    assert out["gathered_words"] == want
