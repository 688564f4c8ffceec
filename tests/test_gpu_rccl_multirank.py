"""The C-ABI exchange step with MORE THAN ONE RANK (cst_gather_sizes_rccl, cst_gather_rccl, cst_scatter_rccl:
constriction_amd/csrc/cst_rccl.hip; BASELINE config C5).  Real RCCL refuses two ranks on one device and the
box has one GPU, so 2-3 PROCESSES share that GPU and the library opens the test double of tests/rccl_double/ through
CST_RCCL_LIB; everything but the wire is the product path: encode and compaction of each rank's block of streams, the
all-gather of the sizes, grouped point-to-point transfers into their final displacements, the rebasing of the offsets, the
inverse scatter and a decode from the scattered buffer.  The root compares every gathered stream with the oracle."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _launch(tmp_path, counts, root, env_per_rank=None, n_per=120, timeout_ms=30000):
    from tests.rccl_double import build
    double = build()
    rendezvous = tmp_path / "rendezvous"
    procs = []
    for rank in range(len(counts)):
        spec = dict(rank=rank, counts=counts, root=root, n_per=n_per, P=12, rendezvous=str(rendezvous), out=str(tmp_path / f"rank{rank}.json"))
        env = dict(os.environ, CST_RCCL_LIB=str(double), FAKE_RCCL_DIR=str(tmp_path), FAKE_RCCL_TIMEOUT_MS=str(timeout_ms),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_per_rank or {}).get(rank, {}))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "rccl_double" / "worker.py"), json.dumps(spec)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("TIMEOUT\n" + p.communicate()[0])
    verdicts = []
    for rank in range(len(counts)):
        f = tmp_path / f"rank{rank}.json"
        assert f.exists(), f"rank {rank} wrote no verdict:\n{logs[rank][-3000:]}"
        verdicts.append(json.loads(f.read_text()))
    return verdicts


@pytest.mark.parametrize("counts,root", [
    ([70, 45], 0),            # two ranks, the usual root
    ([70, 45], 1),            # root != 0: the root's own block is NOT the first one
    ([33, 0, 61], 2),         # three ranks, the middle one has no streams at all, the last one is the root
    ([0, 129, 5], 1),         # the first rank is empty: the root's displacement is 0 although it is not rank 0
])
def test_gather_scatter_decode_multi_rank(tmp_path, counts, root):
    verdicts = _launch(tmp_path, counts, root)
    for v in verdicts:
        assert v.get("ok"), v.get("exception") or v
        assert v["group_depth"] == 0
        assert v["sizes"][0::2] == counts
    assert verdicts[root]["gathered_words"] == sum(verdicts[0]["sizes"][1::2]) > 0
    # a rank that is not the root posts (words, offsets) once for the gather; an empty rank posts nothing
    for rank, v in enumerate(verdicts):
        if rank != root:
            assert v["sends"] == (2 if counts[rank] else 0)


def test_a_failing_transfer_closes_the_group_on_both_sides(tmp_path):
    """rank 1's first ncclSend fails (injected): cst_gather_rccl returns CST_ERR_HIP there AFTER closing the group it opened;
    the root, whose receive never gets its message, returns an error as well (the double's receive times out), also with
    its group closed -- nobody is left inside an open group, nobody hangs."""
    verdicts = _launch(tmp_path, [40, 40], 0, {1: {"FAKE_RCCL_FAIL_SEND": "1"}}, timeout_ms=3000)
    for v in verdicts:
        assert not v.get("ok") and "exception" not in v, v
        assert "cst_gather_rccl" in v["error"] and v["group_depth"] == 0 and v["group_ends"] >= 1
