#!/bin/bash
# per-step kernel sequence of the C1-sized step: tools/c1_trace.sh
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/c1tr -- python $R/tools/c1_hostprof.py > /tmp/c1tr.log 2>&1
f=$(find /tmp/c1tr -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f k_gemm_nt_areg 500 | head -40
