export COLD=1
for st in 1552 1648 2080; do
  export STRIDE=$st
  echo -n "lane "; python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
  echo -n "dq   "; CST_DQ_DECODER=1 python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
done
