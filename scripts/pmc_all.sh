#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_all.sh <tag> -- SQ counter breakdown of every hand-scheduled kernel at the C2 shape,
# one block per kernel, written to gpurun_out/<tag>_sq_counters.txt (copy to profiles/ to commit)
tag=${1:-r02}
out=gpurun_out/${tag}_sq_counters.txt
: > $out
for spec in "12 ans" "12 ans_dec" "24 ans" "24 ans_dec" "12 range" "12 range_dec" "24 range" "24 range_dec" "12 w16" "12 w16_dec"; do
  set -- $spec
  echo "==== P = $1, $2" >> $out
  P=$1 scripts/pmc_range.sh ${tag}_$2_$1 $2 2>&1 | grep -v "^$" >> $out
done
python - <<PY >> $out
print("==== per symbol and wave (4 x counter / waves / 4096 symbols for the cycle counters)")
PY
tail -5 $out
