#!/bin/bash
# kernel timeline of the sharded step on one rank (world 1): tools/dist_trace.sh <tag> <marker>
tag=${1:-dtr}; marker=${2:-k_shard_route}
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dt_$tag -- python $R/bench.py --gpus 2 --steps 60 --warmup 20 > $R/gpurun_out/${tag}.log 2>&1
f=$(find /tmp/dt_$tag -name "*kernel_trace.csv" | head -1)
cp $f $R/gpurun_out/${tag}_kernel_trace.csv
python $R/tools/trace_gaps.py $f "$marker" 30 | tee $R/gpurun_out/${tag}.txt | cut -c1-150
tail -1 $R/gpurun_out/${tag}.log | cut -c1-160
