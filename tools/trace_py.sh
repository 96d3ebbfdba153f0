#!/bin/bash
# usage: tools/trace_py.sh <tag> <marker kernel> <python script + args>   -- per-step timeline of any script (tools/trace_gaps.py)
tag=$1; marker=$2; shift; shift
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python "$@" > $R/gpurun_out/tr_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f "$marker" 10 | tee $R/gpurun_out/tr_$tag.txt
