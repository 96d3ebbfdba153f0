#!/bin/bash
# usage: bash tools/kstats_cmd.sh <tag> <command...>  -> per-kernel stats of an arbitrary command
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o ks -- "$@" > $OUT/${TAG}_stdout.txt 2>/tmp/ks_$TAG.err
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $OUT/${TAG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
for r in rows[:${TOPN:-12}]:
    print("  %-58s calls=%5s avg=%9.1f us" % (r['Name'][:58].replace('void ',''), r['Calls'], float(r['AverageNs'])/1e3))
PY
