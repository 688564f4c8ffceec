#!/bin/bash
# Code placement (round 5; the round-4 experiment of profiles/r04_placement.txt repeated with pinned loop heads): every generated
# loop head sits on a 64-byte boundary (scripts/asmgen.py).  Builds, per kernel, the loop at phases 0, 4, 8, 16, 32 bytes behind
# the boundary (GEN_ALIGN_PAD s_nop) and WITHOUT the directive (GEN_ALIGN=0: where the compiler's code in front happens to put it),
# then times them alternately on the GPU box.  usage: scripts/placement_ab.sh build | run
set -e
cd "$(dirname "$0")/.."
V="p0:GEN_ALIGN_PAD=0 p1:GEN_ALIGN_PAD=1 p2:GEN_ALIGN_PAD=2 p4:GEN_ALIGN_PAD=4 p8:GEN_ALIGN_PAD=8 free:GEN_ALIGN=0"
if [ "$1" = build ]; then
  scripts/exp_variants_par.sh scripts/gen_encode_loop_pc.py constriction_amd/csrc/cst_ans_pc.hip $(for v in $V; do echo pc_$v; done)
  scripts/exp_variants_par.sh scripts/gen_decode_loop.py constriction_amd/csrc/cst_api.hip $(for v in $V; do echo dec_$v; done)
  scripts/exp_variants_par.sh scripts/gen_range_decode_loop.py constriction_amd/csrc/cst_range_fast.hip $(for v in $V; do echo rdec_$v; done)
  exit 0
fi
for round in 1 2; do
  for v in $V; do n=${v%%:*}
    AB_LIB=constriction_amd/lib/variants/pc_$n.so python scripts/bench_one_lib.py ans 12 encode
    AB_LIB=constriction_amd/lib/variants/dec_$n.so python scripts/bench_one_lib.py ans 12 decode
    AB_LIB=constriction_amd/lib/variants/rdec_$n.so python scripts/bench_one_lib.py range 12 decode
    AB_LIB=constriction_amd/lib/variants/rdec_$n.so python scripts/bench_one_lib.py range 24 decode
  done
done
