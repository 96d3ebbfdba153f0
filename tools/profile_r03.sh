#!/bin/bash
# Round-3 evidence run (GPU box, repo root): kernel stats + FETCH/WRITE PMC of the bench workloads,
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
bash tools/profile.sh r03_c3_b16384
bash tools/profile.sh r03_c2_b16384 --workload c2
bash tools/profile.sh r03_c3mix_b16384 --workload c3mix
cd /tmp
# K7 microbenchmarks: per-kernel split
for B in 16384 65536; do
  rm -rf /tmp/k7_$B
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7_$B -o ks -- python $REPO/tools/k7bench.py $B > $OUT/r03_k7bench_$B.txt 2>/dev/null
  f=$(find /tmp/k7_$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r03_k7bench_${B}_kernel_stats.csv
done
# K1 past the LLC: time + FETCH_SIZE
python $REPO/tools/k1_physical.py > $OUT/r03_k1_past_llc.txt 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/k1pmc -o pmc -- python $REPO/tools/k1_physical.py > /dev/null 2>&1
f=$(find /tmp/k1pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $REPO/tools/pmc_summarize.py $f FETCH_SIZE > $OUT/r03_k1_past_llc_pmc_fetch.csv
# C4 kernel stats
rm -rf /tmp/c4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4 -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 > $OUT/r03_c4_lstm_b1024.json 2>/dev/null
f=$(find /tmp/c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r03_c4_lstm_b1024_kernel_stats.csv
ls -la $OUT | tail -30
# MFMA utilisation of the three scorer GEMMs (north_star: "MFMA utilisation on the GEMMs"): SQ counters, PMC-only pass
cd /tmp
GB_B=16384 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pgm -o pmc -- python $REPO/tools/gemmbench.py > $OUT/r03_gemm_times.txt 2>/dev/null
f=$(find /tmp/pgm -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > $OUT/r03_gemm_mfma_counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if 'gemm' not in k and 'splitk' not in k:
        continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-34s mean %.6g  (n=%d)' % (name, sum(v) / len(v), len(v)))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c:
        mb, sb = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES']), sum(c['SQ_BUSY_CYCLES']) / len(c['SQ_BUSY_CYCLES'])
        print('   MFMA busy / SQ busy cycles         %.3f' % (mb / sb))
PY
# K7 of the C3 / C2 step alone (phase split, per-kernel)
rm -rf /tmp/k7g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7g -o ks -- python $REPO/tools/k7grp_bench.py 16384 65536 > $OUT/r03_k7grp_bench.txt 2>/dev/null
f=$(find /tmp/k7g -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r03_k7grp_bench_kernel_stats.csv
K7_MODE=c2 python $REPO/tools/k7grp_bench.py 16384 >> $OUT/r03_k7grp_bench.txt 2>/dev/null
ls -la $OUT | tail -30
