export STRIDE=2080
for kind in 1 read; do
  export COLD=$kind
  echo -n "COLD=$kind lane "; python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
  echo -n "COLD=$kind dq   "; CST_DQ_DECODER=1 python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
done
