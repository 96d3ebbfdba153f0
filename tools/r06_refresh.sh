#!/bin/bash
# Round-6 last evidence pass on the FINAL library (k_run_apply's 128-entry work items came after tools/profile_r06.sh
# had run): calibration copy, kernel stats + FETCH/WRITE PMC of C3 / C2 / C3-MIX, C4 kernel stats.  Outputs: gpurun_out/.
set -u
REPO=$(pwd); mkdir -p $REPO/gpurun_out
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/cal_$C -o pmc -- python $REPO/tools/pmc_calib.py > /dev/null 2>&1
  f=$(find /tmp/cal_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summarize.py $f $C > $REPO/gpurun_out/calib_pmc_$C.csv
done
cd $REPO
bash tools/profile.sh r06_c3_b16384 > /dev/null 2>&1
bash tools/profile.sh r06_c2_b16384 --workload c2 > /dev/null 2>&1
bash tools/profile.sh r06_c3mix_b16384 --workload c3mix > /dev/null 2>&1
cd /tmp
for LS in mw mce; do
  rm -rf /tmp/c4$LS
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4$LS -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 --loss $LS > $REPO/gpurun_out/r06_c4${LS}_lstm_b1024.json 2>/dev/null
  f=$(find /tmp/c4$LS -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/gpurun_out/r06_c4${LS}_lstm_b1024_kernel_stats.csv
done
cd $REPO
ls -la gpurun_out | tail -30
