R=$PWD; cd /tmp; export TMPDIR=/tmp
for f in "" 1; do
  rm -rf /tmp/pl_$f
  ARX_NO_POOL_BITMAP=$f timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pl_$f -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-rooflines --subs= > /dev/null 2>&1
  g=$(find /tmp/pl_$f -name "*counter_collection.csv" | head -1)
  echo "NO_BITMAP=[$f]"; python $R/tools/pmc_summarize.py $g FETCH_SIZE | grep -i "loss_margin" | rev | cut -d, -f1-4 | rev
done
