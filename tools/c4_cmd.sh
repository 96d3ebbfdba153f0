#!/bin/bash
# C4 (LSTM) check: parity tests, bench line, kernel stats.  usage: tools/c4_cmd.sh <tag>
tag=${1:-c4}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_lstm_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/lstm_bench.py --batch 1024 2>&1 | tail -1 | tee gpurun_out/${tag}_bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/tools/lstm_bench.py --batch 1024 --steps 30 --warmup 5 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
head -24 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
