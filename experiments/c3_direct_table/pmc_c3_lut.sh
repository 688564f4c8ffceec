#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_c3_lut.sh <tag> -- SQ counters of the per-stream-table decoders at 65 536 x 4096 (C3): the
# 32-lane direct-table decoder (default) beside the k = 8 sub-lane decoder (CST_PT_LUT=0).  Two rocprofv3 --pmc passes; per wave-symbol =
# counter / (1024 x 4096) (cycle counters x 4: they count in units of four cycles).  Output: gpurun_out/<tag>_c3_lut_counters.md
set -u
tag=${1:-r06}
export TMPDIR=/tmp
R=$PWD
cat > /tmp/c3lut_run.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B, _native as N
n, k = 65536, 4096
mu_d, sigma_d = bench.c3_parameters(bench.SEED, 0, n, k, "cuda")
m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
sym = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, m3.cdfs_device(), 12)
for dt in (torch.int32, torch.int8):
    d = sym.to(dt)
    dec = torch.empty_like(d)
    for knob in ("1", "0"):
        os.environ["CST_PT_LUT"] = knob; N.reload_knobs()
        enc = B.ans_encode(d, m3, (32, 64, 12))
        for _ in range(3):
            B.ans_decode(enc, m3, k, out=dec)
        assert torch.equal(dec, d)
torch.cuda.synchronize()
PY
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
            "b SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  set -- $pass; name=$1; shift
  d=gpurun_out/${tag}_c3lut_$name; rm -rf $d; mkdir -p $d
  timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $R/$d -o pmc -- python /tmp/c3lut_run.py > /dev/null 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
done
python - <<PY > gpurun_out/${tag}_c3_lut_counters.md
import csv, glob, collections, statistics
want = ("ans_decode_pt_lut_kernel", "ans_decode_pt_sub_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${tag}_c3lut_[ab]/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(w in k for w in want): continue
        per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per.items():
        agg[k][c].append(v)
cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
        "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"]
cyc = {"SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA"}
print("# ${tag}: SQ counters of the per-stream-table decoders, C3 at 65 536 x 4096, per wave-symbol (counter / (1024 x 4096), cycle counters x 4)\n")
print("| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |")
print("|---|" + "---|" * len(cols))
for k in sorted(agg):
    row = []
    for c in cols:
        v = agg[k].get(c)
        if not v: row.append("-"); continue
        m = statistics.median(v)
        row.append(f"{m:.0f}" if c == "SQ_WAVES" else f"{m * (4 if c in cyc else 1) / (1024 * 4096):.1f}")
    print(f"| \`{k[:70]}\` | " + " | ".join(row) + " |")
PY
cat gpurun_out/${tag}_c3_lut_counters.md
