#!/bin/bash
# A/B of K7 window-apply variants (tools/build_variant.sh) -- run on the GPU box from the repo root
set -u
OUT=gpurun_out/k7ab; mkdir -p $OUT
for v in base nb4 nb8 nb4ru4; do
  if [ $v = base ]; then unset ARX_LIB; else export ARX_LIB=$PWD/a-recsys_amd/arx/lib/exp/$v.so; fi
  echo "== $v" | tee -a $OUT/log.txt
  python tools/scatterbench.py mulhot100k mulhot1m 2>&1 | grep list | tee -a $OUT/log.txt
  python bench.py --mulhot --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C3 ms/step', j['ms_per_step'], {k:round(v,4) for k,v in j['kernels_ms'].items() if 'scatter' in k or 'mulhot' in k})" | tee -a $OUT/log.txt
done
unset ARX_LIB
python tools/gatherbench.py 2>&1 | tee $OUT/gather.txt
