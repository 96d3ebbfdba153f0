set -u
# the two exchanges of the sharded id-only step at world 1 (same kernels, no wire): what the logits form costs in launches / GEMMs
R=$PWD; O=$R/gpurun_out/r06_exchange; mkdir -p $O
for ex in rows logits; do
  timeout 900 python bench.py --workload c5 --exchange $ex --subs "" --no-rooflines --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$ex', 'world 1, B_loc %d, S %d:' % (j['config']['batch_per_gpu'], j['config']['n_sampled']), '%.1f us/step' % (1e3*j['ms_per_step']), j['config'].get('step_form'), '| predicted link time at 8 ranks %.1f us' % j['roofline_comm_predicted']['at_8_ranks']['us_total_at_link_rate'])" | tee -a $O/exchange.txt
done
