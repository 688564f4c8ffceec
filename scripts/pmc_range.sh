#!/bin/bash
# usage (GPU box, repo root): [P=24] scripts/pmc_range.sh <tag> [range|ans|w16|range_dec|ans_dec|w16_dec]  -- SQ counters of the hand-scheduled encoder at the C2 / C4 shape
set -u
tag=${1:-rng}; which=${2:-range}
case $which in *_dec) pat=decode;; *) pat=encode;; esac
export TMPDIR=/tmp
R=$PWD
cat > /tmp/rrun.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B
P = int("${P:-12}")
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
sym = bench.synth_symbols_device(0xC0FFEE, 0, 65536, 4096, -50, cdf, P)
which = "$which"
f = B.range_encode if which.startswith("range") else B.ans_encode
cfg = (16, 32, P) if which.startswith("w16") else (32, 64, P)
_f = f
f = lambda *a, **k: _f(a[0], a[1], cfg, **k)
enc = f(sym, m, cfg)
if which.endswith("_dec"):
    g = B.range_decode if which.startswith("range") else B.ans_decode
    out = torch.empty_like(sym)
    for _ in range(3):
        g(enc, m, 4096, out=out)
else:
    for _ in range(3):
        f(sym, m, (32, 64, P), out=enc)
torch.cuda.synchronize()
PY
d=gpurun_out/${tag}
mkdir -p $d
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $d/sq_counters.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_STALL SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $R/$d/p$i -o pmc -- python /tmp/rrun.py > /dev/null 2> $d/err$i.log
  find $d/p$i -mindepth 2 -name "*.csv" -exec mv {} $d/p$i/ \;
  for f in $d/p$i/*counter_collection.csv; do [ -f "$f" ] && python scripts/pmc_summary.py $f | grep -A12 "$pat" ; done
done
