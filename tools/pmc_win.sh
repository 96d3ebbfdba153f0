#!/bin/bash
# SQ wave-lifetime accounting of the K7 apply kernels inside the bench step: tools/pmc_win.sh [bench args]
# (PMC-only pass, no trace domains).  mean wave lifetime = SQ_WAVE_CYCLES / SQ_WAVES (shader clocks).
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/pw -- python $R/bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 "$@" > /tmp/pw.log 2>&1
f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:48]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    import os
    if not any(t in k for t in os.environ.get('PMC_FILTER', 'sparse,gather').split(',')):
        continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-22s mean %.5g  (n=%d)' % (name, sum(v) / len(v), len(v)))
PY
