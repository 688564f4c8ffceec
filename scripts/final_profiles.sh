#!/bin/bash
# usage (GPU box, repo root): scripts/final_profiles.sh <tag> -- every measurement the round's profiles/ files come from, raw
# outputs under gpurun_out/<tag>_*; scripts/make_profile_summary.py <tag> and a few copies turn them into profiles/<tag>_*.
set -u
tag=${1:-r03}
export TMPDIR=/tmp
bash scripts/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1
bash scripts/pmc_all.sh $tag > gpurun_out/${tag}_pmc_all.log 2>&1
bash scripts/pmc_c3.sh ${tag}c3 > gpurun_out/${tag}_c3_counters.txt 2>&1
bash scripts/pmc_per_symbol.sh ${tag}ps > gpurun_out/${tag}_per_symbol_counters.txt 2>&1
python scripts/bench_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_api_variants.txt
python scripts/bench_per_symbol.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_per_symbol.txt
python scripts/bench_dropin_single_stream.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_dropin_single_stream.txt
scripts/microbench/bin/encstep > gpurun_out/${tag}_encstep.txt 2>&1
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.json
