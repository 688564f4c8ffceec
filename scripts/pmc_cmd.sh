#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_cmd.sh <tag> <kernel-name pattern> <command ...>  -- SQ counters of the matching kernels of any command
set -u
tag=$1; pat=$2; shift 2
export TMPDIR=/tmp
R=$PWD
d=gpurun_out/${tag}
mkdir -p $d
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $R/$d/p$i -o pmc -- "$@" > /dev/null 2> $d/err$i.log
  find $d/p$i -mindepth 2 -name "*.csv" -exec mv {} $d/p$i/ \;
  for f in $d/p$i/*counter_collection.csv; do [ -f "$f" ] && python scripts/pmc_summary.py $f | grep -A12 "$pat" ; done
done
