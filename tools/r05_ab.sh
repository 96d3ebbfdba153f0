#!/bin/bash
# Round-5 A/B on one box, alternating, N rounds: the C3 / C2 step with
#   new        this build (item tables' K7 pass under the user-gradient product, split apply)
#   late       ARX_K7_LATE_ITEMS=1 (split apply at the step's tail)
#   riderwin   ARX_K7_RIDER_WIN=1 (rounds 3-4: window + finish launches over the one-hot list, then the token apply)
# usage: tools/r05_ab.sh [rounds] [variant libs ...]
set -u
N=${1:-2}; shift || true
OUT=gpurun_out/r05ab; mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --subs c2,c3mix --repeats 3 --no-rooflines 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'C3 %.1f us' % (1e3*j['ms_per_step']), 'C2 %.1f us' % (1e3*j['sub']['c2']['ms_per_step']), 'C3mix %.1f us' % (1e3*j['sub']['c3mix']['ms_per_step']))" | tee -a $OUT/log.txt
}
for i in $(seq $N); do
  run new A=1
  run late ARX_K7_LATE_ITEMS=1
  run riderwin ARX_K7_RIDER_WIN=1
  for v in "$@"; do run $v ARX_LIB=$PWD/a-recsys_amd/arx/lib/exp/$v.so; done
done
