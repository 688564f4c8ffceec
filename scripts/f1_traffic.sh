#!/bin/bash
# usage (GPU box, repo root): scripts/f1_traffic.sh <tag> [n_streams n_per] -- the per-symbol Gaussian kernels (f1) under rocprofv3:
# kernel durations (trace) and HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes; FETCH_SIZE x2 on gfx950 as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes), per launch, against the algorithmic bytes (16 B of parameters + 4 B of symbol per
# symbol + 4 B per word).  Output: gpurun_out/<tag>_f1_traffic.txt
set -u
tag=${1:-r06}; shift
export TMPDIR=/tmp
R=$PWD
cmd="python $R/scripts/bench_per_symbol.py $*"
for pass in trace fetch write; do mkdir -p gpurun_out/${tag}_f1_$pass; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_f1_trace -o ps -- $cmd > gpurun_out/${tag}_f1_trace/out.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_f1_fetch -o pmc -- $cmd > /dev/null 2> gpurun_out/${tag}_f1_fetch/err.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_f1_write -o pmc -- $cmd > /dev/null 2> gpurun_out/${tag}_f1_write/err.log
for pass in trace fetch write; do find gpurun_out/${tag}_f1_$pass -mindepth 2 -name "*.csv" -exec mv {} gpurun_out/${tag}_f1_$pass/ \; ; done
python - "$tag" <<'PY' > gpurun_out/${tag}_f1_traffic.txt
import csv, collections, glob, sys
tag = sys.argv[1]
def counter(pass_, name):
    out = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}_f1_{pass_}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_f1_trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(open(f"gpurun_out/{tag}_f1_trace/out.txt").read().strip())
print(f"{'kernel':90s} {'calls':>5s} {'median us':>10s} {'fetch GB (x2)':>14s} {'write GB':>9s} {'total GB':>9s}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    if "gaussian" not in k and "ckpt" not in k:
        continue
    d = sorted(dur[k]); med = d[len(d) // 2]
    fe = max(fetch.get(k, [0])) * 1024 * 2 / 1e9          # (the full-size launches: the largest value)
    wr = max(write.get(k, [0])) * 1024 / 1e9
    print(f"{k[:90]:90s} {len(d):5d} {med:10.1f} {fe:14.3f} {wr:9.3f} {fe + wr:9.3f}")
PY
cat gpurun_out/${tag}_f1_traffic.txt
