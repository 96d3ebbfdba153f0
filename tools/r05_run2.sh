set -u
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r05/pytest2.txt
cat gpurun_out/r05/pytest2.txt
timeout 600 python bench.py --no-cpu-baseline --subs c2,c3mix,c4 --repeats 3 --no-rooflines 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C3 %.1f us' % (1e3*j['ms_per_step']), {k: round(1e3*v['ms_per_step'],1) for k,v in j['sub'].items() if isinstance(v,dict) and 'ms_per_step' in v})"
timeout 300 python tools/lstm_bench.py --batch 1024 2>&1 | tail -3
