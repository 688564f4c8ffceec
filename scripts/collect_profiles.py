#!/usr/bin/env python3
"""Copies what scripts/final_profiles.sh left under gpurun_out/<tag>_* into the committed profiles/<tag>_* files (after
scripts/make_profile_summary.py <tag> and scripts/make_sq_counters_md.py <tag>)."""
import csv
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = Path(__file__).resolve().parent.parent
g, p = root / "gpurun_out", root / "profiles"
shutil.copy(g / f"{tag}_bench.json", p / f"{tag}_bench.json")
shutil.copy(g / f"{tag}_stats" / "bench.json", p / f"{tag}_bench_under_rocprof.json")
(p / f"{tag}_api_variants.txt").write_text(
    f"# {tag}: scripts/bench_variants.py -- kernel times (HIP events, median of 5 launches each) of API variants around the headline shape,\n"
    "# 65 536 streams, shared quantized Gaussian; round trips checked.\n" + (g / f"{tag}_api_variants.txt").read_text())
(p / f"{tag}_encstep.txt").write_text(
    f"# {tag}: scripts/microbench/encstep.hip -- the (32,64) encoder's 24-instruction step alone (no tile staging, no memory traffic),\n"
    "# one wave per SIMD, 1 and 256 workgroups; shader-clock cycles per step\n" + (g / f"{tag}_encstep.txt").read_text())


def stats(path, script):
    rows = [r for r in csv.DictReader(open(path)) if "cst::" in r["Name"]]
    return [f"| `{script}` | `{r['Name'].replace('void cst::', '').split('(')[0]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |"
            for r in rows[:8]]


lines = [f"# {tag}: per-symbol models (side benchmarks, 1 MI355X)", "",
         "`rocprofv3 --kernel-trace --stats` of `scripts/bench_per_symbol.py` (65 536 streams x 4096 symbols, every symbol its own f64 (mean, std))",
         "and `scripts/bench_dropin_single_stream.py` (ONE coder through the drop-in API: 10^3, 10^5, 10^6 symbols).  The scripts' own output (wall",
         "clock of the calls, host copies included for the drop-in):", "", "```"]
lines += (g / f"{tag}_per_symbol.txt").read_text().strip().splitlines()
lines += (g / f"{tag}_dropin_single_stream.txt").read_text().strip().splitlines()
lines += ["```", "", "| script | kernel | calls | average us | share of GPU time |", "|---|---|---|---|---|"]
lines += stats(g / f"{tag}ps_stats" / "ps_kernel_stats.csv", "bench_per_symbol")
lines += ["", f"SQ counters per symbol and wave: `profiles/{tag}_sq_counters.md` (rows f1)."]
(p / f"{tag}_per_symbol.md").write_text("\n".join(lines) + "\n")
print("profiles written:", sorted(x.name for x in p.glob(f"{tag}_*")))
