#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline numbers are checked against.
# Run on the GPU box from the repo root:  bash tools/profile.sh <tag> [extra bench args]
# Outputs (copied to gpurun_out/, then committed under profiles/ by hand):
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats   (same command as the bench line)
#   <tag>_pmc_fetch.csv      --pmc FETCH_SIZE   (own pass: FETCH_SIZE takes 3 of 4 TCC slots)
#   <tag>_pmc_write.csv      --pmc WRITE_SIZE   (own pass)
# PMC passes never combine with sys/hip/hsa traces (node-crash guard of this pool).
set -u
TAG=${1:-r01}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --subs= $*"

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o ks -- \
  python "$REPO/bench.py" $ARGS > "$OUT/${TAG}_bench_under_rocprof.json" 2> /tmp/ks.err
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats.csv"

for C in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$lc -o pmc -- \
    python "$REPO/bench.py" --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-graph --no-rooflines --subs= $* > /dev/null 2> /tmp/pmc_$lc.err
  f=$(find /tmp/pmc_${TAG}_$lc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python "$REPO/tools/pmc_summarize.py" "$f" $C > "$OUT/${TAG}_pmc_${lc}.csv"
  else
    tail -5 /tmp/pmc_$lc.err
  fi
done
ls -la "$OUT" | tail -8
