#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_small.sh <tag> -- SQ counters of the kernels behind the C5 shard (131072 streams)
set -u
tag=${1:-c5}
export TMPDIR=/tmp
R=$PWD
cat > /tmp/c5run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 12)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
sym = bench.synth_symbols_device(0xC0FFEE, 0, n, 4096, -50, cdf, 12)
enc = B.ans_encode(sym, m, (32, 64, 12)); dec = torch.empty_like(sym)
for _ in range(5):
    B.ans_encode(sym, m, (32, 64, 12), out=enc); B.ans_decode(enc, m, 4096, out=dec)
torch.cuda.synchronize()
print(bool(torch.equal(dec, sym)))
PY
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR"; do
  d=gpurun_out/${tag}_$(echo $grp | cut -c4-9)
  mkdir -p $d
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $R/$d -o pmc -- python /tmp/c5run.py ${2:-131072} > /dev/null 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python - <<PY
import csv, glob, collections
for f in glob.glob("$d/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "ans_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        for c, x in sorted(v.items()):
            per = x / n[(k, c)]
            print(f"   {c:28s} {per:16.0f} per launch   {per / (${2:-131072} * 4096 / 64):10.2f} per wave-symbol")
PY
done
