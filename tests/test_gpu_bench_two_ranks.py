"""Pre-flight of the first real multi-GPU run: `bench.py --gpus 2` on ONE GPU (CST_BENCH_SHARE_GPU=1: both ranks use device 0;
backend gloo, because RCCL refuses two ranks on one device; no gather).  What it exercises is the script's own N > 1 path -- the
self-launch under torch.distributed.run on 127.0.0.1, the stream shard per rank, the barrier-bracketed timed region, the MAX over
ranks, the per-rank block of the line -- so that the driver's 8-GPU run does not meet it first."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = Path(__file__).resolve().parent.parent


def test_bench_with_two_ranks_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    env = dict(os.environ, CST_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-gather", "--steps", "3", "--warmup", "1",
                          "--no-configs", "--no-cpu-baseline", "--no-end-to-end", "--slab-stride", "default"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2, res.stdout[-2000:]                      # rank 0 prints the detail record, then THE line
    assert "bench_detail" in json.loads(lines[0])
    assert len(lines[-1]) < 4096
    line = json.loads(lines[-1])
    assert "roofline" in line and "configs" not in line and "after_cache_flush" not in line
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["rccl_ranks_seen"] == 2
    assert sorted(r["rank"] for r in line["per_rank"]) == [0, 1]
    assert all(r["encode_ms"] > 0 and r["decode_ms"] > 0 and r["words"] > 0 for r in line["per_rank"])
    assert line["per_rank"][0]["words"] != line["per_rank"][1]["words"]        # the ranks code DIFFERENT streams
    assert line["bit_exact"] is True
    assert line["value"] > 0 and abs(line["value"] - 2 * 65536 * 4096 / line["ms_per_step"] / 1e3) < 1e-3 * line["value"]
