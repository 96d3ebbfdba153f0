cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mce_scorer or mw_scorer" 2>&1 | tail -5 > gpurun_out/t_mce_kernel.txt
python tools/mcebench.py 51200 1024 1024 2>&1 | grep -v amdgpu.ids | head -8 > gpurun_out/mcebench_c4.txt
python tools/lstm_bench.py --loss mce 2>&1 | tail -1 | cut -c1-200 > gpurun_out/lstm_mce.json
python tools/lstm_bench.py --loss mw 2>&1 | tail -1 | cut -c1-200 > gpurun_out/lstm_mw.json
timeout 1500 python -m pytest tests -x -q -m gpu -k "mce" 2>&1 | tail -5 > gpurun_out/t_mce_all.txt
cat gpurun_out/t_mce_kernel.txt gpurun_out/mcebench_c4.txt gpurun_out/lstm_mce.json gpurun_out/lstm_mw.json gpurun_out/t_mce_all.txt
