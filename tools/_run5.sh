REPO=/root/repo
cd $REPO; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
for L in mce mw; do
rm -rf /tmp/c4$L
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4$L -o ks -- python $REPO/tools/lstm_bench.py --batch 1024 --loss $L > $REPO/gpurun_out/c4${L}_prof.json 2>/dev/null
f=$(find /tmp/c4$L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/gpurun_out/c4${L}_kernel_stats.csv
done
cd $REPO
python - <<'PY'
import csv
for L in ('mce','mw'):
    rows=list(csv.DictReader(open("gpurun_out/c4%s_kernel_stats.csv"%L)))
    print(L)
    tot=0
    for r in rows[:32]:
        print("  %-64s calls=%5s avg=%8.1f us  %5.1f%%" % (r['Name'].replace('void ','').replace('arx::(anonymous namespace)::','')[:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
