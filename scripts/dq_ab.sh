for st in 1552 1648 2048 1664; do
  export STRIDE=$st
  bash scripts/ab_variants.sh "ans 32 64 12" dbase dspread
  echo -n "dq "; CST_DQ_DECODER=1 python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
done
