cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mce_scorer" 2>&1 | tail -5 > gpurun_out/t_mce_kernel.txt
python tools/mcebench.py 51200 1024 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/mcebench_c4.txt
python tools/mcebench.py 16384 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/mcebench_b16k.txt
cat gpurun_out/t_mce_kernel.txt gpurun_out/mcebench_c4.txt gpurun_out/mcebench_b16k.txt
