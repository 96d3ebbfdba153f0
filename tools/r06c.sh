set -u
R=$PWD; O=$R/gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "scorer" 2>&1 | tail -4 > $O/tests.txt
python -m pytest tests/test_lstm_gpu.py tests/test_hmf_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O/tests.txt
for e in new old new old; do
  if [ $e = old ]; then export ARX_SC_ROWS_OLD=1; else unset ARX_SC_ROWS_OLD; fi
  echo "== $e" >> $O/sb.txt
  python tools/scorerbench.py 51200 1024 64 2>&1 | grep -E "prep|hinge|rows|fwd" >> $O/sb.txt
  python tools/scorerbench.py 16384 1024 128 2>&1 | grep -E "rows|fwd" >> $O/sb.txt
  python bench.py --no-cpu-baseline --subs c2,c4 --repeats 3 --no-rooflines 2>/dev/null | grep "^BENCH_DETAIL " | cut -c14- | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$e', 'C3 %.1f us' % (1e3*j['ms_per_step']), ' '.join('%s %.1f us' % (k, 1e3*v['ms_per_step']) for k, v in j.get('sub',{}).items() if 'ms_per_step' in v))" >> $O/sb.txt
done
cat $O/tests.txt $O/sb.txt
