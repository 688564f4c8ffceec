#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_sublanes.sh <tag> -- SQ counters of the sub-lane decoders (round 5: k jump points per
# stream, two waves per SIMD) next to the plain decoders of the same words: C3 (k = 8, four waves per SIMD), range P = 12 and P = 24 (k = 2), at
# 65 536 x 4096.  Two rocprofv3 --pmc passes (counters never share a run with a trace) + one --kernel-trace --stats pass.
# Output: gpurun_out/<tag>_sublane_counters.md (copy to profiles/).
set -u
tag=${1:-r05}
export TMPDIR=/tmp
R=$PWD
cat > /tmp/sublanes_run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B
n, k = 65536, 4096
mu_d, sigma_d = bench.c3_parameters(bench.SEED, 0, n, k, "cuda")
m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
sym = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, m3.cdfs_device(), 12)
dec = torch.empty_like(sym)
enc, ck = B.ans_encode_checkpointed(sym, m3, k // 8, (32, 64, 12))
for _ in range(3):
    B.ans_decode(enc, m3, k, out=dec)
    B.ans_decode_checkpointed(enc, ck, m3, k, out=dec)
del sym, enc, ck, m3
for P in (12, 24):
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
    sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
    enc, ck = B.range_encode_checkpointed(sym, m, k // 2, (32, 64, P))
    for _ in range(3):
        B.range_decode(enc, m, k, out=dec)
        B.range_decode_checkpointed(enc, ck, m, k, out=dec)
torch.cuda.synchronize()
PY
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
            "b SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  set -- $pass; name=$1; shift
  d=gpurun_out/${tag}_sub_$name; mkdir -p $d
  timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $R/$d -o pmc -- python /tmp/sublanes_run.py > /dev/null 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
done
d=gpurun_out/${tag}_sub_stats; mkdir -p $d
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -o st -- python /tmp/sublanes_run.py > /dev/null 2> $d/err.log
find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
python - <<PY > gpurun_out/${tag}_sublane_counters.md
import csv, glob, collections
want = ("ans_decode_pt_kernel", "ans_decode_pt_sub_kernel", "range_decode_fast_kernel", "range_decode_sub_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${tag}_sub_[ab]/*counter_collection.csv"):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(w in k for w in want): continue
        per_dispatch[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per_dispatch.items():
        agg[k][c].append(v)
stats = {}
for f in glob.glob("gpurun_out/${tag}_sub_stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(w in r["Name"] for w in want): stats[r["Name"]] = r
norm = 65536 * 4096 / 64
print("# ${tag}: SQ counters of the sub-lane decoders next to the plain decoders of the same words (65 536 x 4096)\n")
print("scripts/pmc_sublanes.sh: rocprofv3 --pmc in two passes, medians over the launches of a kernel; per symbol and wave = counter / (1024 x 4096)")
print("wave-symbols (cycle counters x 4: they count in units of four cycles).  C3: k = 8 jump points per stream (FOUR resident waves per SIMD); range: k = 2.  Two resident")
print("waves per SIMD in the range sub-lane kernels, one in the plain ones -- WAVE_CYCLES per wave-symbol is the residence time of a wave, the time")
print("per symbol of the SIMD is WAVE_CYCLES / resident waves.\n")
cols = ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES"]
cyc = {"SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"}
print("| kernel | rocprof avg us | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |")
print("|---|---|" + "---|" * len(cols))
import statistics
for k in sorted(agg):
    cells = []
    for c in cols:
        if c not in agg[k]: cells.append("n/a"); continue
        v = statistics.median(agg[k][c])
        cells.append(f"{v:.0f}" if c == "SQ_WAVES" else f"{v * (4 if c in cyc else 1) / norm:.1f}")
    avg = next((float(r["AverageNs"]) / 1e3 for n, r in stats.items() if n == k), float("nan"))
    print(f"| \`{k[:70]}\` | {avg:.1f} | " + " | ".join(cells) + " |")
PY
cat gpurun_out/${tag}_sublane_counters.md
rm -rf gpurun_out/${tag}_sub_a gpurun_out/${tag}_sub_b gpurun_out/${tag}_sub_stats
