set -u
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_k7_rider_modes_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r05/pytest_modes.txt
cat gpurun_out/r05/pytest_modes.txt
timeout 900 bash tools/r05_ab.sh 2
ARX_K7_RIDER=flow bash tools/trace_cmd.sh r05_c3_flow k_sc_prep --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
cat gpurun_out/tr_r05_c3_flow.txt
ARX_K7_RIDER=flow timeout 300 python bench.py --no-cpu-baseline --subs "" --repeats 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(j.get('roofline_hbm')))"
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_k7_rider_modes_gpu.py 2>&1 | tail -15 > gpurun_out/r05/pytest1.txt
cat gpurun_out/r05/pytest1.txt
