set -u
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_bench_gpu.py tests/test_sampler_gpu.py -q -x 2>&1 | tail -25 > gpurun_out/r05/pytest4.txt
cat gpurun_out/r05/pytest4.txt
for f in "--sharded-rep-tokens" "--sharded-bags"; do
  timeout 600 python bench.py --gpus 1 --workload c5 $f --subs "" --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5 world-1 [$f]', '%.1f us/step' % (1e3*j['ms_per_step']), '%.1f M/s' % (j['value']/1e6), j['config'].get('step_form'), 'roofline', j['roofline']['kernel'][:40], round(j['roofline']['frac'],3), round(1e3*j['roofline']['ms_per_launch'],1),'us')" | tee -a gpurun_out/r05/c5_world1.txt
done
