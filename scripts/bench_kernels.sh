#!/bin/bash
# quick GPU check: BA parity tests + per-kernel times of the default bench
timeout 600 python -m pytest tests/test_ba_gpu.py -q -m gpu -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'obs/s', d['value'], d['parity_vs_oracle'])
print('  '.join('%s %.1f' % (k, v['avg_us']) for k, v in d['kernels'].items()))
"
