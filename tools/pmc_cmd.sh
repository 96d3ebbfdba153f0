#!/bin/bash
# usage: bash tools/pmc_cmd.sh <tag> "<COUNTERS space separated>" <command...>
# one rocprofv3 --pmc pass (no traces), per-kernel means printed and saved to gpurun_out/<tag>_pmc_<counter>.csv
TAG=$1; CTRS=$2; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pmc_$TAG -o pmc -- "$@" > /dev/null 2>/tmp/pmc_$TAG.err
f=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 /tmp/pmc_$TAG.err; exit 1; fi
for c in $CTRS; do
  python $REPO/tools/pmc_summarize.py $f $c > $OUT/${TAG}_pmc_$c.csv
  echo "-- $c"; head -${TOPN:-8} $OUT/${TAG}_pmc_$c.csv | cut -c1-60,150-
done
