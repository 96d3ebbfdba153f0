#!/bin/bash
# same-box A/B of the scorer's forward launches by rocprofv3 kernel durations: old.so (committed) against the tree's lib
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -n "${ONE_SHAPE:-}" ]; then SHAPES=("16384 1024 128"); else SHAPES=("16384 1024 128" "51200 1024 64"); fi
for v in ${VARIANTS:-old new}; do
  for shape in "${SHAPES[@]}"; do
    if [ $v = new ]; then unset ARX_LIB; else export ARX_LIB=$R/a-recsys_amd/arx/lib/exp/$v.so; fi
    rm -rf /tmp/prof_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o x -- python $R/tools/scorerbench.py $shape > /tmp/sb_$v.log 2>&1
    echo "== $v $shape"
    f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
    python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'k_sc_' in n:
        print("  %-40s calls %5s avg %7.2f us" % (n.replace("void ", "").replace("arx::(anonymous namespace)::", "")[:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  done
done
