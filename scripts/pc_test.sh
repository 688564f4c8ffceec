python -m pytest tests/test_gpu_pc_encoder.py -x -q 2>&1 | tail -15
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
