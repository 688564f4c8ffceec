set -x
python -m pytest tests/test_gpu_ans_batch.py -m gpu -x -q -k "tuned_stride or encode_status" 2>&1 | tail -3
for i in 1 2 3; do
python bench.py --no-configs --no-cpu-baseline --slab-stride default | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('DEFAULT', d['value'], d['encode_ms'], d['decode_ms'], d['slab_stride_words'], d['roofline']['frac'], d['after_cache_flush'])"
python bench.py --no-configs --no-cpu-baseline --slab-stride tuned | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('TUNED', d['value'], d['encode_ms'], d['decode_ms'], d['slab_stride_words'], d['roofline']['frac'], d['after_cache_flush'])"
done
