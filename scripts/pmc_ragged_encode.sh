#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_ragged_encode.sh [n_docs] -- SQ counters of the ragged encoder on n_docs documents of 20 .. 2000
# symbols (the bench's distribution), for CST_RAGGED_GROUP = 8, 16, 32: where do the ~340 cycles per symbol of the longest chain go?
set -u
export TMPDIR=/tmp
R=$PWD
n=${1:-2000}
cat > /tmp/ragged_run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B
P, n_sym = 24, 64
w = 0.93 ** np.arange(n_sym)
prob = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n_sym)).astype(np.int64)); prob[0] += (1 << P) - int(prob.sum())
cdf = np.concatenate([[0], np.cumsum(prob)]).astype(np.uint32)
model = B.Model.from_cdf(cdf, 0, P)
rng = np.random.default_rng(bench.SEED)
lengths = np.exp(rng.uniform(np.log(20), np.log(2000), $n)).astype(np.int64)
offsets = np.zeros($n + 1, dtype=np.int64); np.cumsum(lengths, out=offsets[1:])
gen = torch.Generator(device="cuda").manual_seed(1)
q = torch.randint(0, 1 << P, (int(offsets[-1]),), generator=gen, device="cuda", dtype=torch.int64)
flat = (torch.searchsorted(torch.from_numpy(cdf.astype(np.int64)).cuda(), q, right=True) - 1).to(torch.int32)
off_d = torch.from_numpy(offsets).cuda()
for _ in range(4):
    enc = B.ans_encode_ragged(flat, off_d, model, (32, 64, P), jump_every=0)
torch.cuda.synchronize()
print("symbols", int(offsets[-1]), "longest", int(lengths.max()))
PY
for g in 8 16 32; do
  d=gpurun_out/ragged_pmc_$g; rm -rf $d; mkdir -p $d
  CST_RAGGED_GROUP=$g timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $R/$d -o pmc -- python /tmp/ragged_run.py > $d/out.log 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$d/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ans_encode_ragged_kernel" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
last = per[sorted(per, key=int)[-1]]
print("G=$g", {k: int(v) for k, v in last.items()})
PY
  tail -1 $d/out.log
done
