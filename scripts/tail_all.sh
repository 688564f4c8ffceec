timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
CST_NO_PC_ENCODER=1 python scripts/bench_min.py ans 32 64 12 2>/dev/null | tail -1
python scripts/bench_min.py ans 32 64 24 2>/dev/null | tail -1
python scripts/bench_min.py ans 32 64 12 4096 symbol_major 2>/dev/null | tail -1
python scripts/bench_c5.py 2>/dev/null | tail -2
