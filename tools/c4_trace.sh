R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c4 -- python $R/tools/lstm_bench.py --batch 1024 --steps 40 --warmup 10 > $R/gpurun_out/tr_c4.log 2>&1
f=$(find /tmp/tr_c4 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f k_copy_words 15 | tee $R/gpurun_out/tr_c4.txt
