# split (loader + storer waves, the default) against combined helper waves, alternating
for i in 1 2 3 4 5 6; do
  echo -n "split    "; python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
  echo -n "combined "; CST_PC_COMBINED=1 python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
done
