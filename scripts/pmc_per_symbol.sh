#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_per_symbol.sh <tag> -- SQ counters of the per-symbol Gaussian kernels
set -u
tag=${1:-ps}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/${tag}_sq gpurun_out/${tag}_sq2 gpurun_out/${tag}_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o ps -- python $R/scripts/bench_per_symbol.py > gpurun_out/${tag}_stats/out.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/${tag}_sq -o pmc -- python $R/scripts/bench_per_symbol.py > /dev/null 2> gpurun_out/${tag}_sq/err.log
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/${tag}_sq2 -o pmc -- python $R/scripts/bench_per_symbol.py > /dev/null 2> gpurun_out/${tag}_sq2/err.log
for d in sq sq2 stats; do find gpurun_out/${tag}_$d -mindepth 2 -name "*.csv" -exec mv {} gpurun_out/${tag}_$d/ \; ; done
for d in sq sq2; do for f in gpurun_out/${tag}_$d/*counter_collection.csv; do python scripts/pmc_summary.py $f 4096; done; done
grep -E "gaussian|entries|decode" gpurun_out/${tag}_stats/*kernel_stats.csv | head
