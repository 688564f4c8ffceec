#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_c4.sh <tag> -- SQ counters of the range-coder kernels (C4) and the P = 24 ANS kernels
set -u
tag=${1:-c4}
export TMPDIR=/tmp
R=$PWD
cat > /tmp/c4run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B
for P in (12, 24):
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
    sym = bench.synth_symbols_device(0xC0FFEE, 0, 65536, 4096, -50, cdf, P)
    enc = B.range_encode(sym, m, (32, 64, P)); dec = torch.empty_like(sym)
    for _ in range(3):
        B.range_encode(sym, m, (32, 64, P), out=enc); B.range_decode(enc, m, 4096, out=dec)
    if P == 24:
        enc2 = B.ans_encode(sym, m, (32, 64, P))
        for _ in range(3):
            B.ans_encode(sym, m, (32, 64, P), out=enc2); B.ans_decode(enc2, m, 4096, out=dec)
torch.cuda.synchronize()
PY
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  d=gpurun_out/${tag}_sq
  mkdir -p $d
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $R/$d -o pmc -- python /tmp/c4run.py > /dev/null 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python - <<PY
import csv, glob, collections
for f in glob.glob("$d/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "range_" not in k and "ans_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        print("   " + "  ".join(f"{c[3:]}={x / n[(k, c)] / (65536 * 4096 / 64):.1f}" for c, x in sorted(v.items())))
PY
done
