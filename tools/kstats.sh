#!/bin/bash
# usage: bash tools/kstats.sh <tag> <bench args...>   -> gpurun_out/<tag>_kernel_stats.csv + top-25 print
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o ks -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_bench.json 2>/tmp/ks_$TAG.err
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $OUT/${TAG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
for r in rows[:25]:
    print("%-60s calls=%5s avg=%9.1f us tot%%=%s" % (r['Name'][:60].replace('void ',''), r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
