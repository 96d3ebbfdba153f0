set -u
# round 6: C3 / C2 with the sampled softmax 'mce' on the fused family at d = 128 against the materialising path (ARX_MCE_FUSED=0)
R=$PWD; O=$R/gpurun_out/r06_mce128; mkdir -p $O; rm -f $O/ab.txt
run() {
  timeout 600 python bench.py --no-cpu-baseline --subs c3mce,c2mce --steps 10 --warmup 5 --repeats 1 --no-rooflines 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$1', ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j.get('sub',{}).items() if 'ms_per_step' in v))" | tee -a $O/ab.txt
}
for e in 1 2; do
  unset ARX_MCE_FUSED; run "fused (k_mc_flow<.,128>)"
  ARX_MCE_FUSED=0 run "materialising (K4x + K6)"
done
python tools/mcebench.py 16384 1024 16384 128 2>&1 | tail -8 | tee $O/mcebench_d128.txt
