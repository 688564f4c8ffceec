"""CPU-only: the RCCL test double that lets the C-ABI exchange (cst_gather_sizes_rccl / cst_gather_rccl / cst_scatter_rccl)
run with several ranks on a one-GPU box (tests/rccl_double/fake_rccl.cpp; the multi-rank runs themselves need the GPU:
tests/test_gpu_rccl_multirank.py).  Here: it exports every nccl* symbol the library resolves, the library opens it through
CST_RCCL_LIB, and its own transport is right -- three processes, host buffers, an in-place all-gather and a grouped
exchange with a rank that sends nothing, plus the failure paths the GPU tests rely on."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def double():
    from tests.rccl_double import build
    return build()


def test_double_exports_what_the_library_resolves(double):
    wanted = re.findall(r'CST_SYM\(\w+,\s*"(\w+)"\)', (ROOT / "constriction_amd" / "csrc" / "cst_rccl.hip").read_text())
    assert len(wanted) == 8
    lib = C.CDLL(str(double))
    for name in wanted:
        assert hasattr(lib, name), name


def test_library_opens_the_library_named_by_CST_RCCL_LIB(double, tmp_path):
    """cst_rccl_get_unique_id / cst_rccl_comm_init only call into RCCL: they run without a GPU, and with the override the id
    is the double's rendezvous directory.  A named library that does not load is an error (no silent fall-back)."""
    code = ("import sys, numpy as np, ctypes as C; sys.path.insert(0, %r)\n"
            "from constriction_amd import _native as N\n"
            "lib = N.load_library(); buf = np.zeros(128, dtype=np.uint8)\n"
            "rc = lib.cst_rccl_get_unique_id(buf.ctypes.data); h = C.c_void_p()\n"
            "rc2 = lib.cst_rccl_comm_init(buf.ctypes.data, 3, 2, C.byref(h)) if rc == 0 else -99\n"
            "print(rc, rc2, bytes(buf).split(b'\\0')[0].decode())\n" % str(ROOT))
    env = dict(os.environ, CST_RCCL_LIB=str(double), FAKE_RCCL_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, rc2, path = out.stdout.split()
    assert rc == "0" and rc2 == "0" and path.startswith(str(tmp_path)) and Path(path).is_dir()
    env["CST_RCCL_LIB"] = str(tmp_path / "no_such_library.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split()[0] == "-3", out.stdout + out.stderr        # CST_ERR_NO_DEVICE


_RANK = r"""
import ctypes as C, json, sys, numpy as np
lib = C.CDLL(sys.argv[1]); rank, n, iddir, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
class Id(C.Structure): _fields_ = [("internal", C.c_char * 128)]
uid = Id(); uid.internal = iddir.encode()
comm = C.c_void_p()
lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Id, C.c_int]
for f in (lib.ncclSend, lib.ncclRecv): f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
assert lib.ncclCommInitRank(C.byref(comm), n, uid, rank) == 0
U32, U64 = 3, 5                                     # ncclUint32, ncclUint64
res = {}
sizes = np.zeros(2 * n, dtype=np.uint64); sizes[2 * rank: 2 * rank + 2] = (rank, 1000 + rank)
assert lib.ncclAllGather(sizes[2 * rank:].ctypes.data, sizes.ctypes.data, 2, U64, comm, None) == 0       # in place
res["sizes"] = sizes.tolist()
root = 1
mine = np.arange(100 * rank, dtype=np.uint32) + 7 * rank          # rank 0 sends nothing at all
assert lib.ncclGroupStart() == 0
rcs = []
if rank != root:
    if len(mine): rcs.append(lib.ncclSend(mine.ctypes.data, len(mine), U32, root, comm, None))
    if mode == "ok": rcs.append(lib.ncclSend(sizes.ctypes.data, 2 * n, U64, root, comm, None))
else:
    bufs = {r: (np.zeros(100 * r, dtype=np.uint32), np.zeros(2 * n, dtype=np.uint64)) for r in range(n) if r != root}
    for r, (a, b) in bufs.items():
        if len(a): rcs.append(lib.ncclRecv(a.ctypes.data, len(a), U32, r, comm, None))
        rcs.append(lib.ncclRecv(b.ctypes.data, 2 * n if mode != "size" else 2 * n + 1, U64, r, comm, None))
res["post"] = rcs
res["end"] = lib.ncclGroupEnd()
res["depth"] = lib.fake_rccl_group_depth()
if rank == root and res["end"] == 0:
    res["got"] = {str(r): [a.tolist(), b.tolist()] for r, (a, b) in bufs.items()}
print(json.dumps(res))
"""


def _run(double, tmp_path, mode, extra_env=None):
    iddir = tmp_path / ("id_" + mode)
    iddir.mkdir()
    procs = []
    for rank in range(3):
        env = dict(os.environ, FAKE_RCCL_HOST="1", FAKE_RCCL_TIMEOUT_MS="1500", **(extra_env or {}).get(rank, {}))
        procs.append(subprocess.Popen([sys.executable, "-c", _RANK, str(double), str(rank), "3", str(iddir), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    return [json.loads(o[0]) for o in outs]


def test_double_transport_three_processes(double, tmp_path):
    res = _run(double, tmp_path, "ok")
    sizes = [0, 1000, 1, 1001, 2, 1002]
    assert all(r["sizes"] == sizes and r["end"] == 0 and r["depth"] == 0 and all(x == 0 for x in r["post"]) for r in res)
    got = res[1]["got"]
    assert got["0"] == [[], sizes] and got["2"] == [(np.arange(200) + 14).tolist(), sizes]


def test_double_reports_a_missing_message_and_a_size_mismatch(double, tmp_path):
    """what the GPU failure test builds on: a receive whose send never comes times out with an error at ncclGroupEnd (and
    the group is closed), a receive posted with the wrong count fails, an injected send failure is returned by ncclSend"""
    res = _run(double, tmp_path, "missing")                        # ranks 0 and 2 never send their sizes
    assert res[1]["end"] != 0 and res[1]["depth"] == 0 and res[0]["end"] == 0
    res = _run(double, tmp_path, "size")
    assert res[1]["end"] != 0 and res[1]["depth"] == 0
    res = _run(double, tmp_path, "ok", {2: {"FAKE_RCCL_FAIL_SEND": "1"}})
    assert res[2]["post"][0] != 0 and res[2]["depth"] == 0 and res[1]["end"] != 0
