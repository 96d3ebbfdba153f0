set -u
R=$PWD; O=$R/gpurun_out/r06b; mkdir -p $O
bash tools/r06_prio.sh 2 > $O/prio_stdout.txt 2>&1
cp gpurun_out/r06prio/log.txt $O/prio_log.txt
for shape in "51200 64" "16384 128"; do
  ARX_LIB=$R/a-recsys_amd/arx/lib/exp/sc_rows_trace.so SC_PHASE=4 timeout 120 python tools/sc_trace.py $shape 2>&1 | grep -v amdgpu.ids > "$O/rows_trace_$(echo $shape | tr ' ' '_').txt"
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/bagsks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bagsks -o ks -- python $R/bench.py --gpus 1 --workload c5 --sharded-bags --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines > $O/bags_w1.json 2>/dev/null
f=$(find /tmp/bagsks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bags_w1_kernel_stats.csv
cd $R
python -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/bench_tests.txt
cat $O/prio_log.txt $O/bench_tests.txt
