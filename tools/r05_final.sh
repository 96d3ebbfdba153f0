#!/bin/bash
# Round-5 final check on one box: the whole GPU suite, then the evidence run (tools/profile_r05.sh).
set -u
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r05/pytest_final.txt
cat gpurun_out/r05/pytest_final.txt
bash tools/profile_r05.sh > gpurun_out/r05/profile_log.txt 2>&1
tail -3 gpurun_out/r05/profile_log.txt | cut -c1-300
