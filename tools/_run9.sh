cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -60 > gpurun_out/t_all.txt
cat gpurun_out/t_all.txt
