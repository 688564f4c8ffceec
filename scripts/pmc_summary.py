#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter_collection.csv for the coder kernels (per wave, per step)."""
import csv, collections, sys
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if any(t in k for t in ('ans_', 'range_', 'gaussian', 'entries', 'decode_wave', 'decode_rows')):
        agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    waves = m.get('SQ_WAVES', 1)
    print(k)
    for c, val in sorted(m.items()):
        extra = ''
        if c in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_VALU',
                 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_LDS_IDX_ACTIVE', 'SQ_INST_CYCLES_VMEM'):
            extra = f'  -> {val / waves * 4 / steps:8.1f} cycles/step/wave'
        elif c.startswith('SQ_INSTS'):
            extra = f'  -> {val / waves / steps:8.2f} instr/step/wave'
        print(f'   {c:28s} {val:16.1f}{extra}')
