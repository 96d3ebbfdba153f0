#!/bin/bash
# Round-5 A/B on one box, alternating, N rounds: the C3 / C2 / C3-MIX step with variant libraries (tools/build_variant.sh)
# and / or ARX_K7_RIDER modes.  usage: tools/r05_ab.sh [rounds] [variant libs ...]
set -u
N=${1:-2}; shift || true
OUT=gpurun_out/r05ab; mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --subs ${SUBS:-c2,c3mix} --repeats 3 --no-rooflines 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'C3 %.1f us' % (1e3*j['ms_per_step']), ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j['sub'].items() if 'ms_per_step' in v))" | tee -a $OUT/log.txt
}
for i in $(seq $N); do
  run win ARX_K7_RIDER=win
  for v in "$@"; do run $v ARX_LIB=$PWD/a-recsys_amd/arx/lib/exp/$v.so; done
done
