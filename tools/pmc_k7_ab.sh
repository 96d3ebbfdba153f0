# PMC FETCH/WRITE of the window apply kernels for two libarx variants: tools/pmc_k7_ab.sh <lib-or-empty> ...
R=$PWD; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pk
    ARX_LIB=$lib timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pk -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-rooflines --subs= > /dev/null 2>&1
    g=$(find /tmp/pk -name "*counter_collection.csv" | head -1)
    echo "lib=[$lib] $C"; python $R/tools/pmc_summarize.py $g $C | grep -i "sparse_win\|sparse_finish" | rev | cut -d, -f1-4 | rev
  done
done
