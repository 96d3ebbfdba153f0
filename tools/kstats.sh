#!/bin/bash
# kernel stats of one command under rocprofv3 (GPU box): tools/kstats.sh <rows> <command ...>
N=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o x -- "$@" > /tmp/kst.log 2>&1
tail -n 1 /tmp/kst.log | cut -c1-400
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
python - "$f" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print("  %-72s calls %5s avg %8.1f us %5.1f%%" % (r["Name"].replace("void ", "").replace("arx::(anonymous namespace)::", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
