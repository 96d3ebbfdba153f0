#!/bin/bash
# Round 6: does a higher priority for the kernels of the ids-only sort branch shorten the C3 / C2 step?
# (hipKernelNodeAttributePriority on the captured nodes, ARX_GRAPH_PRIO=1; a high-priority capture stream, ARX_K7_STREAM_PRIO=1)
set -u
N=${1:-2}
OUT=gpurun_out/r06prio; mkdir -p $OUT
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --subs ${SUBS:-c2,c3mix} --repeats 3 --no-rooflines 2>$OUT/err_$tag.txt | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$tag', 'C3 %.1f us' % (1e3*j['ms_per_step']), ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j.get('sub',{}).items() if 'ms_per_step' in v))" | tee -a $OUT/log.txt
  grep "graph nodes at priority" $OUT/err_$tag.txt | head -3 | tee -a $OUT/log.txt
}
for i in $(seq $N); do
  run base X=1
  run nodeprio ARX_GRAPH_PRIO=1 ARX_GRAPH_PRIO_VERBOSE=1
  run streamprio ARX_K7_STREAM_PRIO=1
  run both ARX_GRAPH_PRIO=1 ARX_K7_STREAM_PRIO=1
done
