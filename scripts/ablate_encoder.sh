#!/bin/bash
# Round 4: what the encoder's coder chain would cost if a helper wave took the tile work (timing only; results wrong).
# usage (build container): scripts/ablate_encoder.sh build    -> constriction_amd/lib/variants/*.so
#       (GPU box):         scripts/ablate_encoder.sh run      -> encode ms per variant
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  scripts/exp_variants.sh scripts/gen_encode_loop.py constriction_amd/csrc/cst_api.hip base: \
      notile:GEN_NO_TILEWORK=1 \
      notile_lring:GEN_NO_TILEWORK=1,GEN_LOCAL_RING=1 \
      notile_lring_amm:GEN_NO_TILEWORK=1,GEN_LOCAL_RING=1,GEN_ADDR_MINMAX=1 \
      notile_lring_amm_bar:GEN_NO_TILEWORK=1,GEN_LOCAL_RING=1,GEN_ADDR_MINMAX=1,GEN_BARRIER=1 \
      amm:GEN_ADDR_MINMAX=1
else
  for v in base notile notile_lring notile_lring_amm notile_lring_amm_bar amm; do
    AB_LIB=constriction_amd/lib/variants/$v.so timeout 300 python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
  done
fi
