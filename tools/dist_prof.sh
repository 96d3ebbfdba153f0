#!/bin/bash
# kernel stats of the sharded step on one rank (world 1): tools/dist_prof.sh <tag>
tag=${1:-dist}
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_$tag -- python $R/bench.py --gpus 2 --steps 100 --warmup 20 > $R/gpurun_out/${tag}.log 2>&1
f=$(find /tmp/dp_$tag -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${tag}_kernel_stats.csv
tail -1 $R/gpurun_out/${tag}.log | cut -c1-200
