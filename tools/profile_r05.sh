#!/bin/bash
# Round-5 evidence run (GPU box, repo root): kernel stats + FETCH/WRITE PMC of the bench workloads, the scorer's
# SQ counters, its stand-alone kernel times, the MFMA issue-slot probe, the C4 step.  Outputs under gpurun_out/r05/.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
bash tools/profile.sh r05_c3_b16384
bash tools/profile.sh r05_c2_b16384 --workload c2
mv gpurun_out/r05_c* $OUT/ 2>/dev/null
# scorer kernels alone: times (C3 shape and C4's per-step shape), then the SQ accounting
python tools/scorerbench.py 16384 1024 128 > $OUT/r05_scorerbench_c3.txt 2>&1
python tools/scorerbench.py 51200 1024 64 > $OUT/r05_scorerbench_c4.txt 2>&1
bash tools/pmc_scorer.sh gpurun_out/r05/pmc_scorer > /dev/null 2>&1
cat $OUT/pmc_scorer/pmc_set0.txt > $OUT/r05_scorer_sq_counters.txt
cat $OUT/pmc_scorer/pmc_set1.txt >> $OUT/r05_scorer_sq_counters.txt
# issue-slot probe (hand-laid MFMA streams with fillers)
python tools/probe/gen_mfma_issue.py > tools/probe/mfma_issue.hip 2>/dev/null   # (generated: not tracked since round 5)
if [ -f tools/probe/mfma_issue.hip ]; then
  [ -x tools/probe/mfma_issue.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_issue.bin tools/probe/mfma_issue.hip 2>/dev/null
  timeout 120 tools/probe/mfma_issue.bin > $OUT/r05_mfma_issue_probe.txt 2>&1
fi
# C4 kernel stats
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/c4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4 -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 > $OUT/r05_c4_lstm_b1024.json 2>/dev/null
f=$(find /tmp/c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_c4_lstm_b1024_kernel_stats.csv
# C4 with the sampled softmax 'mce' (the k_mc_flow family): kernel stats in step, its launches stand-alone beside the
# materialising path's, the fused and the materialising step loss by loss, the step's timeline
rm -rf /tmp/c4mce
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4mce -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 --loss mce > $OUT/r05_c4mce_lstm_b1024.json 2>/dev/null
f=$(find /tmp/c4mce -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_c4mce_lstm_b1024_kernel_stats.csv
rm -rf /tmp/tr_c4mce
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c4mce -- python $REPO/tools/lstm_bench.py --batch 1024 --steps 40 --warmup 10 --loss mce > /dev/null 2>&1
f=$(find /tmp/tr_c4mce -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $REPO/tools/trace_gaps.py $f k_copy_words 15 > $OUT/r05_c4mce_step_timeline.txt
python $REPO/tools/mcebench.py 51200 1024 1024 2>&1 | grep -v amdgpu.ids > $OUT/r05_mcebench_c4.txt
python $REPO/tools/mcebench.py 16384 1024 2>&1 | grep -v amdgpu.ids > $OUT/r05_mcebench_b16384.txt
python $REPO/tools/mce_ab_steps.py 2>&1 | grep "^step" > $OUT/r05_mce_ab_steps.txt
python $REPO/tools/dxbench.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_dxbench.txt
# K7 of the C3 / C2 step alone (phase split)
rm -rf /tmp/k7g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7g -o ks -- python $REPO/tools/k7grp_bench.py 16384 65536 > $OUT/r05_k7grp_bench.txt 2>/dev/null
f=$(find /tmp/k7g -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_k7grp_bench_kernel_stats.csv
cd $REPO
# per-step timeline of the C3 step (who is on the critical path)
bash tools/trace_cmd.sh r05_c3 k_sc_prep --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
mv gpurun_out/tr_r05_c3.txt $OUT/r05_c3_step_timeline.txt 2>/dev/null
ls -la $OUT
# the full default line (headline + subs + CPU baseline), as the driver runs it
cd $REPO
timeout 900 python bench.py > $OUT/r05_bench_default.json 2> $OUT/r05_bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_steps20.json 2>/dev/null
tail -c 600 $OUT/r05_bench_default.json
