cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mce_scorer" 2>&1 | tail -30 > gpurun_out/t_mce_kernel.txt
timeout 300 python tools/mcebench.py 51200 1024 1024 > gpurun_out/mcebench_c4.txt 2>&1
timeout 300 python tools/mcebench.py 16384 1024 > gpurun_out/mcebench_b16k.txt 2>&1
timeout 900 python -m pytest tests/test_hmf_gpu.py tests/test_lstm_gpu.py -x -q -m gpu -k "mce" 2>&1 | tail -30 > gpurun_out/t_mce_steps.txt
cat gpurun_out/t_mce_kernel.txt gpurun_out/mcebench_c4.txt gpurun_out/mcebench_b16k.txt gpurun_out/t_mce_steps.txt
