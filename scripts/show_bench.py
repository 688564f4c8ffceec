import json,sys
ls=[json.loads(l) for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin) if l.startswith('{')]
# the detail record (the line before the last, or bench_detail.json itself) holds everything; older captures have one big line
d=next((l['bench_detail'] for l in ls if 'bench_detail' in l), ls[-1])
print(d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'], d['slab_stride_words'], d['roofline']['frac'], d['roofline']['frac_cold'], d.get('bit_exact'))
for c in d.get('configs', []):
    print("  %-70s enc %.3f dec %.3f  %s  fr %s/%s stride %s ok=%s" % (c['workload'][:70], c.get('encode_ms',0), c.get('decode_ms',0), c.get('Msymbols_per_s'), c.get('encode_frac'), c.get('decode_frac'), c.get('slab_stride_words'), c.get('bit_exact')))
print(d.get('cpu_baseline',{}).get('value'), d.get('configs_error'))
print('rate', d.get('rate')); print('end_to_end', d.get('end_to_end')); print('cpu', d.get('cpu_baseline'))
