cd /root/repo; mkdir -p gpurun_out
for i in 1 2; do python tools/mcebench.py 51200 1024 1024 2>&1 | grep "flow dU\|bwd dI  "; done > gpurun_out/mc_il.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mce_scorer" 2>&1 | tail -2 >> gpurun_out/mc_il.txt
cat gpurun_out/mc_il.txt
