"""The bench line's format contract: the LAST stdout line of bench.py carries the contract's keys only
and stays under 4 KB however large the detail record grows.  Round 5's line was 21 KB and the driver did not parse it."""
import io
import json
import sys
from contextlib import redirect_stdout
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def _r05_line():
    return json.loads((ROOT / "profiles" / "r05_bench.json").read_text().strip().splitlines()[-1])


def test_contract_line_of_the_round5_record_is_small_and_complete():
    full = _r05_line()
    assert len(json.dumps(full)) > 16000          # the record that was not parsed
    short = bench.contract_line(full)
    text = json.dumps(short)
    assert len(text) < bench.LINE_LIMIT
    back = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "bit_exact", "encode_ms", "decode_ms", "roofline", "cpu_baseline"):
        assert key in back, key
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert 0.0 < back["roofline"]["frac"] <= 1.0 and back["roofline"]["bound"] == "hbm"
    assert back["roofline"]["achieved"] / back["roofline"]["peak"] == round(back["roofline"]["frac"], 4) or \
        abs(back["roofline"]["achieved"] / back["roofline"]["peak"] - back["roofline"]["frac"]) < 1e-3
    assert back["cpu_baseline"]["value"] > 0 and back["cpu_baseline"]["kind"] in ("port", "reference")
    assert "workload" in back["config"] and "model" not in back["config"]
    # the reports are NOT in the line
    for key in ("configs", "after_cache_flush", "end_to_end", "rate"):
        assert key not in back
    assert back["configs_reported"] == len(full["configs"])


def test_contract_line_survives_a_grown_record():
    full = _r05_line()
    full["configs"] = full["configs"] * 20
    full["per_rank"] = [{"rank": r, "encode_ms": 0.25, "decode_ms": 0.25, "words": 45_000_000} for r in range(8)]
    full["rccl_ranks_seen"] = 8
    full["configs_error"] = "x" * 300
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["roofline"]["new_prose"] = "z" * 5000
    text = json.dumps(bench.contract_line(full))
    assert len(text) < bench.LINE_LIMIT
    back = json.loads(text)
    assert back["rccl_ranks_seen"] == 8 and len(back["per_rank"]) == 8


def test_emit_prints_the_detail_first_and_the_contract_line_last(tmp_path):
    full = _r05_line()
    out = io.StringIO()
    with redirect_stdout(out):
        bench.emit(full, tmp_path / "bench_detail.json")
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 2
    detail = json.loads(lines[0])["bench_detail"]
    assert detail == full
    assert json.loads((tmp_path / "bench_detail.json").read_text())["bench_detail"] == full
    last = json.loads(lines[-1])
    assert len(lines[-1]) < bench.LINE_LIMIT and last["metric"] == full["metric"] and "roofline" in last and "cpu_baseline" in last
