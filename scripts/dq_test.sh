python scripts/bench_min.py ans 32 64 12 2>&1 | tail -1
for st in 1648 2048; do STRIDE=$st python scripts/bench_min.py ans 32 64 12 2>&1 | tail -1; done
CST_NO_DQ_DECODER=1 python scripts/bench_min.py ans 32 64 12 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_ans_batch.py tests/test_gpu_pc_encoder.py -x -q 2>&1 | tail -8
