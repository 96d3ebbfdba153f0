#!/bin/bash
# Round 6: what the C3 / C2 step would cost without its ids-only sort branch / without its applies (timing-only
# ablations, WRONG results): the ceiling of any K7 rework.  (ARX_ABL_SKIP_SORT alone -- applies walking the FIRST
# batch's lists against this batch's buffers -- faults on the GPU: only together with ARX_ABL_SKIP_APPLY.)  usage: tools/r06_abl.sh [rounds]
set -u
N=${1:-2}
OUT=gpurun_out/r06abl; mkdir -p $OUT
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --subs ${SUBS:-c2} --repeats 3 --no-rooflines 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$tag', 'C3 %.1f us' % (1e3*j['ms_per_step']), ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j.get('sub',{}).items() if 'ms_per_step' in v))" | tee -a $OUT/log.txt
}
for i in $(seq $N); do
  run base X=1
  run noapply ARX_ABL_SKIP_APPLY=1
  run neither ARX_ABL_SKIP_SORT=1 ARX_ABL_SKIP_APPLY=1
done
