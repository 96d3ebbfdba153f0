cd /root/repo
mkdir -p gpurun_out
python tools/mcebench.py 51200 1024 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/mcebench_c4.txt
python tools/lstm_bench.py --loss mce 2>&1 | tail -1 > gpurun_out/lstm_mce.json
ARX_MCE_FUSED=0 python tools/lstm_bench.py --loss mce 2>&1 | tail -1 > gpurun_out/lstm_mce_unfused.json
python tools/lstm_bench.py --loss mw 2>&1 | tail -1 > gpurun_out/lstm_mw.json
timeout 1500 python -m pytest tests -x -q -m gpu -k "mce" 2>&1 | tail -8 > gpurun_out/t_mce_all.txt
cat gpurun_out/mcebench_c4.txt gpurun_out/lstm_mce.json gpurun_out/lstm_mce_unfused.json gpurun_out/lstm_mw.json gpurun_out/t_mce_all.txt
