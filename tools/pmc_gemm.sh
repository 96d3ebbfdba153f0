#!/bin/bash
# SQ stall accounting of the scorer GEMMs: tools/pmc_gemm.sh  (PMC-only pass, no trace domains)
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
GB_B=16384 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pg -- python $R/tools/gemmbench.py > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if 'gemm' not in k and 'splitk' not in k:
        continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-28s mean %.4g  (n=%d)' % (name, sum(v) / len(v), len(v)))
PY
