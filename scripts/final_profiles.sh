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
bash scripts/pmc_sublanes.sh $tag > gpurun_out/${tag}_pmc_sublanes.log 2>&1          # -> gpurun_out/<tag>_sublane_counters.md
python scripts/bench_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_api_variants.txt
python scripts/bench_per_symbol.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_per_symbol.txt
python scripts/bench_dropin_single_stream.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_dropin_single_stream.txt
scripts/microbench/bin/encstep > gpurun_out/${tag}_encstep.txt 2>&1
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.json
# the summaries are made HERE (gpurun merges at most 64 MiB back, the raw rocprofv3 outputs are more): profiles/<tag>_* of this
# copy go home as gpurun_out/<tag>_profiles/, the raw directories stay behind
python scripts/make_profile_summary.py $tag > gpurun_out/${tag}_summary.log 2>&1
python scripts/make_sq_counters_md.py $tag >> gpurun_out/${tag}_summary.log 2>&1
python scripts/collect_profiles.py $tag >> gpurun_out/${tag}_summary.log 2>&1
mkdir -p gpurun_out/${tag}_profiles && cp profiles/${tag}_* profiles/traffic.json gpurun_out/${tag}_profiles/ 2>/dev/null
cp gpurun_out/${tag}_sublane_counters.md gpurun_out/${tag}_profiles/ 2>/dev/null
find gpurun_out -mindepth 1 -maxdepth 1 -type d ! -name "${tag}_profiles" -exec rm -rf {} +
du -sh gpurun_out
