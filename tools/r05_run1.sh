set -u
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05/pytest1.txt
cat gpurun_out/r05/pytest1.txt
timeout 900 bash tools/r05_ab.sh 2
bash tools/trace_cmd.sh r05_c3 k_sc_prep --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
cat gpurun_out/tr_r05_c3.txt
