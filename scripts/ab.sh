#!/bin/bash
# usage: scripts/ab.sh lib1.so lib2.so ...   -> encode/decode ms of bench.py for each experimental build
for v in "$@"; do
  CST_LIB_PATH=$PWD/constriction_amd/lib/$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check 2>/dev/null | tail -1 | V=$v python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['V'], d['encode_ms'], d['decode_ms'])"
done
