#!/bin/bash
# Regenerates every checked-in main-loop statement (constriction_amd/csrc/*.inc) from its generator, variants included.
set -e
cd "$(dirname "$0")/.."
for g in scripts/gen_*loop*.py; do python "$g" > /dev/null; done
GEN_PT_SUB=1 python scripts/gen_pt_decode_loop.py > /dev/null
GEN_PT_SUB=2 python scripts/gen_pt_decode_loop.py > /dev/null
GEN_PT_CK=1 python scripts/gen_pt_encode_loop.py > /dev/null
GEN_RANGE_CK=1 python scripts/gen_range_encode_loop.py > /dev/null
GEN_RANGE_SUB=1 python scripts/gen_range_decode_loop.py > /dev/null
GEN_W16_PACKED=1 python scripts/gen_encode_loop_w16.py > /dev/null
GEN_SMALL_N8=1 python scripts/gen_decode_loop_small.py > /dev/null
GEN_SMALL_N8=16 python scripts/gen_decode_loop_small.py > /dev/null
GEN_N16=1 python scripts/gen_decode_loop_n8.py > /dev/null
GEN_B16_NARROW=1 python scripts/gen_decode_loop_b16.py > /dev/null
GEN_B16_NARROW=2 python scripts/gen_decode_loop_b16.py > /dev/null
for n in 1 2 4; do GEN_B16_SMALL=1 GEN_B16_NARROW=$n python scripts/gen_decode_loop_b16.py > /dev/null; done
GEN_W16_PACKED=1 python scripts/gen_decode_loop_w16.py > /dev/null
git status --short constriction_amd/csrc | grep '\.inc' | wc -l
