#!/bin/bash
# Round-2 evidence run (GPU box, repo root): kernel stats + FETCH/WRITE PMC of the bench workloads,
# the K7 microbenchmarks and the past-LLC K1 gather.  Outputs under gpurun_out/.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# calibration copy
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/cal_$C -o pmc -- python $REPO/tools/pmc_calib.py > /dev/null 2>&1
  f=$(find /tmp/cal_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summarize.py $f $C > $OUT/calib_pmc_$C.csv
done
cd $REPO
bash tools/profile.sh r02_c3_b16384
bash tools/profile.sh r02_c2_b16384 --workload c2
bash tools/profile.sh r02_c3mix_b16384 --workload c3mix
cd /tmp
# K7 microbenchmarks: per-kernel split
for B in 16384 65536; do
  rm -rf /tmp/k7_$B
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7_$B -o ks -- python $REPO/tools/k7bench.py $B > $OUT/r02_k7bench_$B.txt 2>/dev/null
  f=$(find /tmp/k7_$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r02_k7bench_${B}_kernel_stats.csv
done
# K1 past the LLC: time + FETCH_SIZE
GB_V=1000002,4000000 python $REPO/tools/gatherbench.py > $OUT/r02_k1_past_llc.txt 2>/dev/null
GB_V=1000002 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/k1pmc -o pmc -- python $REPO/tools/gatherbench.py > /dev/null 2>&1
f=$(find /tmp/k1pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $REPO/tools/pmc_summarize.py $f FETCH_SIZE > $OUT/r02_k1_past_llc_pmc_fetch.csv
# C4 kernel stats
rm -rf /tmp/c4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4 -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 > $OUT/r02_c4_lstm_b1024.json 2>/dev/null
f=$(find /tmp/c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r02_c4_lstm_b1024_kernel_stats.csv
ls -la $OUT | tail -30
