#!/bin/bash
# usage (GPU box): scripts/pmc_traffic_shapes.sh <tag> 131072x4096 131072x2048 ... -- HBM traffic counters per kernel and batch shape
set -u
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
for shape in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=gpurun_out/${tag}_${shape}_$(echo $c | cut -c1-5)
    mkdir -p $d
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$d -o pmc -- python $R/scripts/bench_shapes.py $shape > $d/out.txt 2> $d/err.log
    find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
    python - <<PY
import csv, glob, collections
for f in glob.glob("$d/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ans_" in k: agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("$shape", k, {c: round(sum(x) / len(x) / 1024, 1) if "SIZE" in c else round(sum(x) / len(x)) for c, x in v.items()}, "(SIZE in MiB)")
PY
  done
done
