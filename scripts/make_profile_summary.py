#!/usr/bin/env python3
"""Turns the rocprofv3 outputs under gpurun_out/rNN_{stats,fetch,write,l2}/ into the committed summaries under
profiles/: rNN_kernel_stats.csv (the --stats table restricted to this repo's kernels), rNN_pmc_summary.md and
profiles/traffic.json (HBM bytes per launch for bench.py's `roofline.traffic`).

FETCH_SIZE / WRITE_SIZE are in KiB (x1024).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts wide coalesced
streaming reads at half their size on gfx950; both the raw and the doubled figure are reported and `traffic` uses
fetch_corrected + write."""
import collections
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
g = ROOT / "gpurun_out"


def pmc(dirname):
    """counter averages per kernel over its FULL-SIZE dispatches only (largest grid): one kernel name serves several workloads
    of a bench run -- the headline batch, the end-to-end leg's chunks of an eighth of it, the tuner's launches -- and an average
    over all of them describes none"""
    rows = [r for r in csv.DictReader(open(g / dirname / "pmc_counter_collection.csv")) if "cst::" in r["Kernel_Name"]]
    biggest = collections.defaultdict(int)
    for r in rows:
        biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        if int(r["Grid_Size"]) == biggest[k]:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


# kernel stats
rows = list(csv.DictReader(open(g / f"{tag}_stats" / "bench_kernel_stats.csv")))
ours = [r for r in rows if "cst::" in r["Name"]]
with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in ours:
        w.writerow(r)
_lines = [json.loads(l) for l in (g / f"{tag}_stats" / "bench.json").read_text().splitlines() if l.startswith("{")]
bench = next((l["bench_detail"] for l in _lines if "bench_detail" in l), _lines[-1])      # the detail record holds the contract line's keys too

# per-dispatch durations from the kernel trace: the median is what the bench's timed steps see (the --stats average also
# holds the first launches and bench.py's three `after_cache_flush` launches per headline kernel)
import statistics
durations = collections.defaultdict(list)
by_grid = collections.defaultdict(list)
trace = g / f"{tag}_stats" / "bench_kernel_trace.csv"
if trace.exists():
    for r in csv.DictReader(open(trace)):
        if "cst::" in r["Kernel_Name"]:
            by_grid[(r["Kernel_Name"], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    top = collections.defaultdict(int)
    for (k, gs) in by_grid:
        top[k] = max(top[k], gs)
    for (k, gs), v in by_grid.items():
        if gs == top[k]:
            durations[k] = v            # full-size dispatches only
    # one row per (kernel, grid size): the --stats table averages over every workload a kernel name serves
    with open(out / f"{tag}_kernel_stats_by_grid.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid_Size_X", "Calls", "AverageUs", "MedianUs", "MinUs", "MaxUs"])
        for (k, gs), v in sorted(by_grid.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k.split("(")[0].replace("void ", ""), gs, len(v), round(sum(v) / len(v), 1), round(statistics.median(v), 1), round(min(v), 1), round(max(v), 1)])

fetch, write, l2 = pmc(f"{tag}_fetch"), pmc(f"{tag}_write"), pmc(f"{tag}_l2")
traffic = {}
lines = [f"# {tag}: rocprofv3 summary of `python bench.py --steps 20 --warmup 3 --no-cpu-baseline` (headline C2 + the `configs` block, 1 MI355X)", "",
         "Kernels by configuration: `ans_encode_pc_kernel` (coder + helper waves) and `ans_decode_kernel<32, 64, 0, true, ...8, true>` = C2 headline",
         "(`ans_encode_kernel<32, 64, 0, true, 8, ...>` = the one-wave encoder: batches the producer / consumer kernel does not take, here the tuner's odd strides); `ans_*_w16_kernel` = 16-bit words;",
         "`ans_encode_pc_kernel<true, false, true>` (the wide step) / `ans_decode_b16_kernel` = P = 24; `range_*_fast_kernel<...>` = C4 (`<1>`/`<false>`: P = 12, `<2>`/`<true>`: P = 24);",
         "`ans_*_pt_kernel` = C3 (per-stream tables); `ans_*_small_kernel` = C5 shard (131072 streams); `compact_kernel` = packing.", "",
         f"bench line: value = {bench['value']} Msym/s, encode {bench['encode_ms']} ms, decode {bench['decode_ms']} ms, "
         f"algorithmic bytes/launch = {bench['roofline']['algorithmic_bytes_per_launch']}", "",
         "Counters and medians are over a kernel's FULL-SIZE dispatches (largest grid) only; `--stats` averages are over all its launches",
         f"(`profiles/{tag}_kernel_stats_by_grid.csv` has one row per kernel and grid size).", "",
         "| kernel | calls | avg us (--stats) / median us (trace, full size) | FETCH_SIZE KiB | x2 corrected GiB | WRITE_SIZE KiB | HBM bytes/launch (corr.) | algorithmic | L2 hit |",
         "|---|---|---|---|---|---|---|---|---|"]
for r in ours:
    k = r["Name"]
    short = k.split("(")[0].replace("void cst::", "")
    fk = fetch.get(k, {}).get("FETCH_SIZE")
    wk = write.get(k, {}).get("WRITE_SIZE")
    h, m = l2.get(k, {}).get("TCC_HIT_sum"), l2.get(k, {}).get("TCC_MISS_sum")
    if fk is None or wk is None:
        continue
    hbm = (2 * fk + wk) * 1024
    # the headline kernels (C2: (32,64), P = 12, stream-major, hand-scheduled) keep their short keys for bench.py's `traffic`
    # (the plain instantiation only: <SPLIT = true, JUMP = false, WIDE = false>; the jump-point and 12 < P <= 24 forms keep their own rows)
    pc_headline = "ans_encode_pc_kernel<true, false, false>" in k or "ans_encode_pc_kernel<true, false>" in k
    headline = pc_headline or ("ans_decode_kernel<32, 64, 0, true, 1, true, 8, true" in k)
    key = ("ans_encode_pc_kernel" if pc_headline else "ans_decode_kernel") if headline else short
    traffic[key] = {"hbm_bytes_per_launch": int(hbm), "fetch_kib_raw": fk, "write_kib": wk,
                    "l2_hit_rate": None if not h else round(h / (h + m), 4)}
    alg = bench["roofline"]["algorithmic_bytes_per_launch"] if headline else ""
    med = f" / {statistics.median(durations[k]):.1f}" if durations.get(k) else ""
    lines.append(f"| `{short}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f}{med} | {fk:.0f} | {2 * fk * 1024 / 2**30:.3f} | {wk:.0f} | "
                 f"{hbm:.3e} | {alg} | {'' if not h else f'{h / (h + m):.3f}'} |")
# ---- the timed loop's own rows: the kernel trace of `bench.py --headline-only` (nothing but hot launches of the headline pair) ----
hl = g / f"{tag}_headline"
if (hl / "bench_kernel_trace.csv").exists():
    hl_lines = [json.loads(l) for l in (hl / "bench.json").read_text().splitlines() if l.startswith("{")]
    hb = next((l["bench_detail"] for l in hl_lines if "bench_detail" in l), hl_lines[-1])
    alg = hb["roofline"]["algorithmic_bytes_per_launch"]
    rows_h = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(hl / "bench_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"])):
        if "cst::" in r["Kernel_Name"]:
            rows_h[(r["Kernel_Name"], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # a kernel's launches in program order: [stride tuner / first call][~30 ms of ramp at rising clocks][warmup][K timed steps][K event-timed
    # steps][one launch that asks for the kernel's name].  The rows below are the 2 K launches of the timed and event-timed steps only.
    K = int(hb["steps"])
    for key, v in list(rows_h.items()):
        if len(v) > 2 * K + 1:
            rows_h[key] = v[-(2 * K + 1):-1]
    lines += ["", f"## The timed loop alone: `python bench.py --steps 20 --warmup 3 --headline-only --no-check` under `rocprofv3 --kernel-trace`", "",
              f"Only launches of the headline pair; the rows are the {2 * int(hb['steps'])} launches of the timed and the event-timed steps (the ramp and warmup launches in front of them run at rising clocks and are left out).  "
              f"The run's own line: encode {hb['encode_ms']} ms, decode {hb['decode_ms']} ms (HIP events), value {hb['value']} Msym/s; algorithmic bytes per launch {alg}.", "",
              "| kernel | grid | calls | avg us | median us | min | max | algorithmic / avg (GB/s) | of 8 TB/s | line's ms / avg |", "|---|---|---|---|---|---|---|---|---|---|"]
    with open(out / f"{tag}_headline_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid_Size_X", "Calls", "AverageUs", "MedianUs", "MinUs", "MaxUs", "AlgorithmicGBps", "FracOf8TBps"])
        for (k, gs), v in sorted(rows_h.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            short = k.split("(")[0].replace("void cst::", "")
            line_ms = hb["encode_ms"] if "encode" in short else hb["decode_ms"] if "decode" in short else None
            agree = "" if line_ms is None else f"{line_ms * 1e3 / avg:.3f}"
            w.writerow([short, gs, len(v), round(avg, 1), round(statistics.median(v), 1), round(min(v), 1), round(max(v), 1), round(alg / avg / 1e3, 1), round(alg / avg / 1e3 / 8000, 4)])
            lines.append(f"| `{short}` | {gs} | {len(v)} | {avg:.1f} | {statistics.median(v):.1f} | {min(v):.1f} | {max(v):.1f} | {alg / avg / 1e3:.0f} | {alg / avg / 1e3 / 8000:.3f} | {agree} |")
(out / f"{tag}_pmc_summary.md").write_text("\n".join(lines) + "\n")
(out / "traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
print("\n".join(lines))
