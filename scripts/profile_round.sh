#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh r01
# rocprofv3 kernel-trace statistics of the default bench run plus three separate PMC passes (HBM fetch, HBM write,
# L2 hit/miss) -- counters never share a run with the trace.  Raw outputs under gpurun_out/<tag>_*; turn them into
# the committed summaries with  python scripts/make_profile_summary.py <tag>.
set -u
tag=${1:-r01}
export TMPDIR=/tmp
R=$PWD
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
# the slab strides bench.py measures at first use (batched.tuned_stride) are measured HERE, by an unprofiled run of the same
# command, and remembered: the profiled runs below then hold the command's own launches only, not 63 measuring launches per
# kernel at 21 other strides
export CST_STRIDE_CACHE=$R/gpurun_out/${tag}_strides.json
rm -f $CST_STRIDE_CACHE
$B > gpurun_out/${tag}_unprofiled_bench.json 2>/dev/null
mkdir -p gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_l2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o bench -- $B > gpurun_out/${tag}_stats/bench.json 2> gpurun_out/${tag}_stats/err.log
# the timed loop alone (round 6): `--headline-only` launches nothing but the headline pair, hot -- its trace rows are what the bench
# line's encode_ms / decode_ms must agree with (the full run's rows also hold the flush legs, foreign-words legs and plain_decode calls)
mkdir -p gpurun_out/${tag}_headline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_headline -o bench -- $B --headline-only --no-check > gpurun_out/${tag}_headline/bench.json 2> gpurun_out/${tag}_headline/err.log
find gpurun_out/${tag}_headline -mindepth 2 -name "*.csv" -exec mv {} gpurun_out/${tag}_headline/ \;
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -o pmc -- $B > /dev/null 2> gpurun_out/${tag}_fetch/err.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -o pmc -- $B > /dev/null 2> gpurun_out/${tag}_write/err.log
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${tag}_l2 -o pmc -- $B > /dev/null 2> gpurun_out/${tag}_l2/err.log
# flatten rocprofv3's per-host subdirectory
for d in stats fetch write l2; do find gpurun_out/${tag}_$d -mindepth 2 -name "*.csv" -exec mv {} gpurun_out/${tag}_$d/ \; ; done
ls gpurun_out/${tag}_stats gpurun_out/${tag}_fetch | head -20
tail -1 gpurun_out/${tag}_stats/bench.json | cut -c1-300
