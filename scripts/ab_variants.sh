#!/bin/bash
# usage (GPU box): scripts/ab_variants.sh "<bench_min.py args>" name1 name2 ...   -> one line per constriction_amd/lib/variants/<name>.so
args=$1; shift
for v in "$@"; do
  AB_LIB=constriction_amd/lib/variants/$v.so timeout 300 python scripts/bench_min.py $args 2>/dev/null | tail -1
done
