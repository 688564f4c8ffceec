#!/usr/bin/env python3
"""Turns gpurun_out/<tag>_sq_counters.txt (scripts/pmc_all.sh), <tag>_c3_counters.txt (scripts/pmc_c3.sh) and
<tag>_per_symbol_counters.txt (scripts/pmc_per_symbol.sh) into profiles/<tag>_sq_counters.md: SQ counters per symbol and wave."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
g = ROOT / "gpurun_out"
N = 4096                                              # symbols per stream in every one of these runs
CYCLES = ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT")
INSTS = ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU")


def blocks(text, header):
    """{title: (kernel, {counter: value})} for every `header`-introduced block; counters of a block's passes are merged"""
    out = {}
    title = kernel = None
    for line in text.splitlines():
        m = re.match(header, line)
        if m:
            title = m.group(1)
            out[title] = [None, {}]
            continue
        if title is None:
            continue
        m = re.match(r"\s+(SQ_[A-Z_0-9]+)\s+([0-9.]+)", line)
        if m:
            out[title][1].setdefault(m.group(1), float(m.group(2)))
        elif line.strip() and not line.startswith(" ") and "cst::" in line and out[title][0] is None:
            out[title][0] = line.strip().strip('"')
    return out


def row(label, kernel, c, waves):
    per = lambda name, cyc: (c[name] * (4 if cyc else 1) / (waves * N)) if name in c else float("nan")
    cells = [f"{per('SQ_WAVE_CYCLES', True):.0f}", f"{per('SQ_INSTS_VALU', False):.1f}", f"{per('SQ_ACTIVE_INST_VALU', True):.0f}",
             f"{per('SQ_INSTS_LDS', False):.1f}", f"{per('SQ_ACTIVE_INST_LDS', True):.0f}", f"{per('SQ_INSTS_SALU', False):.1f}",
             f"{per('SQ_WAIT_ANY', True):.0f}", f"{per('SQ_WAIT_INST_ANY', True):.0f}", f"{per('SQ_LDS_BANK_CONFLICT', True):.0f}"]
    return f"| {label} | `{kernel[:64]}` | " + " | ".join(cells) + " |"


lines = [f"# {tag}: SQ counters of the coder kernels at 65 536 streams x 4096 symbols",
         "",
         "Collected with `scripts/final_profiles.sh` (`scripts/pmc_all.sh`, `pmc_c3.sh`, `pmc_per_symbol.sh`: rocprofv3 --pmc, several passes per",
         "kernel, counters never share a run with a trace).  Per symbol and WAVE: cycle counters x 4 / (waves x 4096 symbols), instruction",
         "counters / (waves x 4096).  One wave per SIMD (1024 waves) except the fused per-symbol encoder (2048 waves of 32 streams)",
         "and the producer / consumer encoder `ans_encode_pc_kernel` (2048 waves: a CODER and a HELPER wave per 64 streams; its row is the",
         "average over both kinds -- per pair: twice the instruction counts, i.e. 23.6 VALU + 2.8 LDS per symbol, nearly all of the VALU",
         "in the coder; the helper's share of the wave cycles is mostly WAIT_ANY, parked at the per-tile barrier).",
         "",
         "| configuration | kernel | wave cycles | VALU instr | VALU cycles | LDS instr | LDS cycles | SALU instr | WAIT_ANY | WAIT_INST_ANY | LDS bank-conflict cycles |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
main = blocks((g / f"{tag}_sq_counters.txt").read_text(), r"==== (P = \d+, \w+)")
for title, (kernel, c) in main.items():
    if kernel and "SQ_WAVES" in c:
        lines.append(row(title, kernel, c, c["SQ_WAVES"]))
# C3: pmc_c3.sh prints "kernel" lines followed by "   COUNTER   value per launch ..."
c3 = {}
kernel = None
for line in (g / f"{tag}_c3_counters.txt").read_text().splitlines():
    if "pt_kernel" in line and not line.startswith(" ") and "," not in line:
        kernel = line.strip()
        c3.setdefault(kernel, {})
    else:
        m = re.match(r"\s+(SQ_[A-Z_0-9]+)\s+([0-9.]+) per launch", line)
        if m and kernel:
            c3[kernel].setdefault(m.group(1), float(m.group(2)))
for kernel, c in c3.items():
    lines.append(row("C3 (one table per stream, P = 12)", kernel, c, c.get("SQ_WAVES", 1024.0)))
ps = blocks((g / f"{tag}_per_symbol_counters.txt").read_text(), r"(void cst::[a-z_]+<\d+, \d+, \d+>)")
seen = {}
for title, (_, c) in ps.items():
    seen.setdefault(title, {}).update(c)
text = (g / f"{tag}_per_symbol_counters.txt").read_text()
# (the per-symbol file lists each kernel twice, once per pass: merge by kernel name)
merged = {}
cur = None
for line in text.splitlines():
    m = re.match(r"void (cst::[a-z_]+<\d+, \d+, \d+>)", line)
    if m:
        cur = m.group(1)
        merged.setdefault(cur, {})
        continue
    m = re.match(r"\s+(SQ_[A-Z_0-9]+)\s+([0-9.]+)", line)
    if m and cur:
        merged[cur].setdefault(m.group(1), float(m.group(2)))
for kernel, c in merged.items():
    kind = "ANS" if kernel.endswith(", 0>") else "range coder"
    lines.append(row(f"f1: per-symbol Gaussians, {kind}", kernel, c, c.get("SQ_WAVES", 1024.0)))
(ROOT / "profiles" / f"{tag}_sq_counters.md").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
