for i in 1 2 3; do python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1; done
python scripts/bench_min.py ans 32 64 12 64 2>/dev/null | tail -1
python scripts/bench_min.py ans 32 64 12 128 2>/dev/null | tail -1
