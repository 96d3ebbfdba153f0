#!/bin/bash
# usage: tools/trace_cmd.sh <tag> <marker> <bench args...>
tag=$1; marker=$2; shift; shift
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python $R/bench.py --no-cpu-baseline --steps 60 --warmup 20 "$@" > $R/gpurun_out/tr_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f "$marker" 30 | tee $R/gpurun_out/tr_$tag.txt
