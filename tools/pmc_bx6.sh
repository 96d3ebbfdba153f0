#!/bin/bash
# SQ accounting of the bf16x6 logits kernel: tools/pmc_bx6.sh  (PMC-only passes)
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
rm -rf /tmp/pb
timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pb -- python $R/tools/bx6_bench.py > /tmp/pb.log 2>&1
f=$(find /tmp/pb -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Grid_Size"]) != int(__import__("os").environ.get("GRID", "131072")): continue
    k = r['Kernel_Name'][:50]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if 'bx6' not in k: continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-28s mean %.4g  (n=%d)' % (name, sum(v) / len(v), len(v)))
PY
done
