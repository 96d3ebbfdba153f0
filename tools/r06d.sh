set -u
R=$PWD; O=$R/gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "adagrad or k7 or sparse" 2>&1 | tail -4 > $O/tests.txt
python -m pytest tests/test_hmf_gpu.py tests/test_fullsize_gpu.py tests/test_k7_rider_modes_gpu.py tests/test_dist_gpu.py tests/test_pooled_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O/tests.txt
cat $O/tests.txt
for e in new old new old; do
  if [ $e = old ]; then export ARX_K7_LDS_SORT=0; else unset ARX_K7_LDS_SORT; fi
  python bench.py --no-cpu-baseline --subs c2,c3mix,c5w1 --repeats 3 --no-rooflines 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$e', 'C3 %.1f us' % (1e3*j['ms_per_step']), ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j.get('sub',{}).items() if 'ms_per_step' in v))" | tee -a $O/ab.txt
done
unset ARX_K7_LDS_SORT
bash tools/trace_cmd.sh r06d_c3 k_sc_prep --subs "" --no-rooflines --repeats 1 > /dev/null 2>&1
cp gpurun_out/tr_r06d_c3.txt $O/c3_timeline_lds_sort.txt
head -32 $O/c3_timeline_lds_sort.txt
