#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_c3.sh <tag> -- SQ counters of the per-stream-table kernels (C3)
set -u
tag=${1:-c3}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/${tag}_sq gpurun_out/${tag}_lds gpurun_out/${tag}_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o c3 -- python $R/scripts/bench_c3.py > gpurun_out/${tag}_stats/out.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/${tag}_sq -o pmc -- python $R/scripts/bench_c3.py > /dev/null 2> gpurun_out/${tag}_sq/err.log
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${tag}_lds -o pmc -- python $R/scripts/bench_c3.py > /dev/null 2> gpurun_out/${tag}_lds/err.log
for d in sq lds stats; do find gpurun_out/${tag}_$d -mindepth 2 -name "*.csv" -exec mv {} gpurun_out/${tag}_$d/ \; ; done
python - <<PY
import csv, glob, collections
for d in ("sq", "lds"):
    for f in glob.glob("gpurun_out/${tag}_%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "pt_kernel" not in k and "ps_kernel" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            print(k)
            for c, x in sorted(v.items()):
                per = x / n[(k, c)]
                print(f"   {c:28s} {per:16.0f} per launch   {per / (65536 * 4096 / 64):10.2f} per wave-symbol")
PY
grep -E "pt_kernel|ps_kernel" gpurun_out/${tag}_stats/*kernel_stats.csv | head
tail -2 gpurun_out/${tag}_stats/out.txt
