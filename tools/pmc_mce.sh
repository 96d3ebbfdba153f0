#!/bin/bash
# SQ accounting of the 'mce' family's kernels (csrc/scorer.hip k_mc_flow / k_mc_rows): tools/pmc_mce.sh [out_file] [mcebench args]
# PMC-only passes (no tracing domains), two counter sets; per-kernel means over the dispatches of tools/mcebench.py
R=$PWD; OUT=${1:-gpurun_out/pmc_mce.txt}; shift; : > $R/$OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
rm -rf /tmp/pbm
timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pbm -- python $R/tools/mcebench.py "$@" > /tmp/pbm.log 2>&1
f=$(find /tmp/pbm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee -a $R/$OUT
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'k_mc_' not in k: continue
    k = k[k.index('k_mc_'):][:40]
    acc[k][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for name, v in sorted(c.items()):
        vals = list(v.values())
        print('   %-28s mean %.5g  (n=%d)' % (name, sum(vals) / len(vals), len(vals)))
PY
done
