#!/bin/bash
# usage (GPU box, repo root): scripts/alt_paths.sh <tag> -- the whole GPU suite through each alternate kernel path (the dispatcher's
# A/B knobs), summary lines to gpurun_out/<tag>_alt_paths.txt.  A path that changes a RESULT fails a parity test.
tag=${1:-r06}
out=gpurun_out/${tag}_alt_paths.txt
: > $out
for env in "CST_AUTO_JUMP=0" "CST_PT_SUB_WAVES=8" "CST_SUB_ORDER=0" "CST_NO_N8=1" "CST_SMALL_KERNELS=0" "CST_NO_PC_ENCODER=1" "CST_PC_COMBINED=1" "CST_NO_PC_WIDE=1" "CST_DQ_DECODER=1" "CST_LANE_GEO=small" "CST_LANE_GEO=big" "CST_RAGGED_GROUP=8" "CST_RAGGED_GROUP=32"; do
  echo "$env" >> $out
  env $env timeout 900 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -E "passed|failed|FAILED|error" | head -12 >> $out
done
cat $out
