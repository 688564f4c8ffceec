#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_f1_traffic.sh <tag> -- HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) of the
# per-symbol Gaussian kernels at 65 536 x 4096 (scripts/bench_per_symbol.py), averages per launch in KiB; the guide's x2
# correction for FETCH_SIZE on gfx950 is applied in the last column.
set -u
tag=${1:-f1}
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/${tag}_$c; mkdir -p $d
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$d -o pmc -- python $R/scripts/bench_per_symbol.py > $d/out.txt 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python - $d/*counter_collection.csv $c <<'PY'
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == sys.argv[2] and any(t in r['Kernel_Name'] for t in ('gaussian', 'entries')):
        agg[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k, v in agg.items():
    v = sorted(v)[len(v) // 2:]                        # (the full-size launches)
    kib = sum(v) / len(v)
    print(f"{sys.argv[2]:10s} {k:70s} {kib:12.0f} KiB  {kib * 1024 * (2 if sys.argv[2] == 'FETCH_SIZE' else 1) / 1e9:7.3f} GB")
PY
done
grep "per-symbol" gpurun_out/${tag}_FETCH_SIZE/out.txt
