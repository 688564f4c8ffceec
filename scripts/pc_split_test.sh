python -m pytest tests/test_gpu_pc_encoder.py -x -q 2>&1 | tail -5
for i in 1 2 3; do python scripts/bench_min.py ans 32 64 12 2>&1 | tail -1; done
for i in 1 2; do CST_PC_COMBINED=1 python scripts/bench_min.py ans 32 64 12 2>&1 | tail -1; done
for i in 1 2; do python scripts/bench_min.py ans 32 64 12 2>&1 | tail -1; done
