#!/bin/bash
# Round-6 evidence run (GPU box, repo root): kernel stats + FETCH/WRITE PMC of the bench workloads (C3, C2), of the
# PHYSICAL K1 measurement (2 GB table, past the LLC: the gather's headline), the calibration copy, the scorer's SQ
# counters, the C4 steps, the C3 step timeline, and the bench lines as the driver runs them.  Outputs: gpurun_out/r06/.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# calibration copy (FETCH_SIZE / WRITE_SIZE -> bytes)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/cal_$C -o pmc -- python $REPO/tools/pmc_calib.py > /dev/null 2>&1
  f=$(find /tmp/cal_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summarize.py $f $C > $REPO/gpurun_out/calib_pmc_$C.csv
done
cd $REPO
bash tools/profile.sh r06_c3_b16384
bash tools/profile.sh r06_c2_b16384 --workload c2
bash tools/profile.sh r06_c3mix_b16384 --workload c3mix      # (the layout whose token list is never sorted: csc.hip)
# K1 past the LLC: the launch bench.py's `k1` sub-result times, under the two PMC passes + kernel stats
cd /tmp
rm -rf /tmp/k1ks
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1ks -o ks -- python $REPO/tools/k1_physical.py > $OUT/r06_k1_past_llc.json 2>/dev/null
f=$(find /tmp/k1ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/gpurun_out/r06_k1_past_llc_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
  rm -rf /tmp/k1_$lc
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/k1_$lc -o pmc -- python $REPO/tools/k1_physical.py > /dev/null 2>&1
  f=$(find /tmp/k1_$lc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summarize.py $f $C > $REPO/gpurun_out/r06_k1_past_llc_pmc_${lc}.csv
done
cd $REPO
mv gpurun_out/r06_c* gpurun_out/r06_k1* $OUT/ 2>/dev/null
cp gpurun_out/calib_pmc_*.csv $OUT/ 2>/dev/null
# scorer kernels alone + SQ accounting
python tools/scorerbench.py 16384 1024 128 > $OUT/r06_scorerbench_c3.txt 2>&1
bash tools/pmc_scorer.sh gpurun_out/r06/pmc_scorer > /dev/null 2>&1
cat $OUT/pmc_scorer/pmc_set0.txt > $OUT/r06_scorer_sq_counters.txt 2>/dev/null
cat $OUT/pmc_scorer/pmc_set1.txt >> $OUT/r06_scorer_sq_counters.txt 2>/dev/null
# C4 kernel stats ('mw' and the sampled softmax 'mce')
cd /tmp
for LS in mw mce; do
  rm -rf /tmp/c4$LS
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4$LS -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 --loss $LS > $OUT/r06_c4${LS}_lstm_b1024.json 2>/dev/null
  f=$(find /tmp/c4$LS -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r06_c4${LS}_lstm_b1024_kernel_stats.csv
done
cd $REPO
# per-step timeline of the C3 step (who is on the critical path) + the timing-only ablations of its K7 halves
bash tools/trace_cmd.sh r06_c3 k_sc_prep --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
mv gpurun_out/tr_r06_c3.txt $OUT/r06_c3_step_timeline.txt 2>/dev/null
bash tools/r06_abl.sh 2 > /dev/null 2>&1
cp gpurun_out/r06abl/log.txt $OUT/r06_k7_halves_ablation.txt 2>/dev/null
# the C3-MIX step timeline (static token order + split apply), the sampled softmax at d = 128
bash tools/trace_cmd.sh r06_c3mix k_sc_prep --workload c3mix --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
mv gpurun_out/tr_r06_c3mix.txt $OUT/r06_c3mix_step_timeline_csc.txt 2>/dev/null
# the bench lines as the driver runs them
timeout 900 python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
cp bench_detail.json $OUT/r06_bench_default_detail.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_steps20.json 2>/dev/null
tail -c 400 $OUT/r06_bench_steps20.json
ls -la $OUT
