#!/bin/bash
# usage: tools/kstats_of.sh <python script + args>   -- per-kernel average durations (us) under rocprofv3 --kernel-trace --stats
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kso
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kso -o ks -- python "$@" > /tmp/kso.log 2>&1
f=$(find /tmp/kso -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    n = r['Name'].replace('arx::(anonymous namespace)::', '').replace('void ', '')
    print('%9.2f us  x%-5s %s' % (float(r['AverageNs']) / 1e3, r['Calls'], n[:70]))
PY
