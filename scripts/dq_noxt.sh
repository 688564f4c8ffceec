export CST_DQ_DECODER=1
for st in 1552 2080; do export STRIDE=$st; bash scripts/ab_variants.sh "ans 32 64 12" q_base q_noxt q_base q_noxt; done
unset CST_DQ_DECODER
STRIDE=2080 bash scripts/ab_variants.sh "ans 32 64 12" q_base
