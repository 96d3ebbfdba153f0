set -u
# round 6: the static token order of the riding bag table (csrc/csc.hip) against the per-step expansion + radix sort
# (ARX_K7_CSC=0), alternating runs on one box, both apply forms (ARX_K7_RIDER); then the step timeline of the default.
# usage (GPU box): bash tools/r06_csc_ab.sh     -> gpurun_out/r06_csc/{tests,ab,c3mix_timeline}.txt
R=$PWD; O=$R/gpurun_out/r06_csc; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "cat_multi_bags" 2>&1 | tail -2 > $O/tests.txt
ARX_K7_CSC=1 timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_hmf_gpu.py -m gpu -x -q 2>&1 | tail -2 | sed "s/^/ARX_K7_CSC=1: /" >> $O/tests.txt
cat $O/tests.txt
run() {
  timeout 600 python bench.py --no-cpu-baseline --subs c3mix --repeats 3 --no-rooflines 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$1', 'C3 %.1f us' % (1e3*j['ms_per_step']), ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j.get('sub',{}).items() if 'ms_per_step' in v))" | tee -a $O/ab.txt
}
for e in 1 2 3; do
  unset ARX_K7_CSC ARX_K7_RIDER; run "default (csc+split where the entity table is virtual: MIX)"
  ARX_K7_CSC=1 run "csc everywhere, split"
  ARX_K7_CSC=1 ARX_K7_RIDER=win run "csc everywhere, win"
  ARX_K7_CSC=0 run "radix everywhere, win"
done
unset ARX_K7_CSC ARX_K7_RIDER
bash tools/trace_cmd.sh r06_c3mix k_sc_prep --workload c3mix --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
cp gpurun_out/tr_r06_c3mix.txt $O/c3mix_timeline.txt
head -30 $O/c3mix_timeline.txt
