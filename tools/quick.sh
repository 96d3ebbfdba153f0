#!/bin/bash
# usage: bash tools/quick.sh "<bench args>" ...   -> value, ms/step, per-kernel us for each arg set
for a in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline $a 2>&1 | grep "^BENCH_DETAIL " | tail -1 | cut -c14- | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('== $a', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step']*1e3,1), 'us/step')
for k,v in d['kernels'].items():
    print('   %-40s %8.1f us %7.0f GB/s %6.1f TF' % (k, v['ms']*1e3, v.get('gbs',0), v.get('tflops',0)))
"
done
