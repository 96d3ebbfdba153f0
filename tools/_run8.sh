cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t_all.txt
timeout 900 python bench.py --subs c4,c4mce --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
tail -3 gpurun_out/t_all.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
for k in ('c4','c4mce'):
    s=d['sub'][k]; print(k, s['ms_per_step'], s.get('roofline_scorer'))
PY
