cd /root/repo
mkdir -p gpurun_out
python tools/mcebench.py 51200 1024 1024 2>&1 | grep -v amdgpu.ids | head -7 > gpurun_out/mc_slots.txt
echo "== 3 slots" >> gpurun_out/mc_slots.txt
ARX_LIB=$PWD/a-recsys_amd/arx/lib/exp/mc_s3.so python tools/mcebench.py 51200 1024 1024 2>&1 | grep -v amdgpu.ids | head -7 >> gpurun_out/mc_slots.txt
python tools/mce_ab_steps.py 2>&1 | grep -v amdgpu.ids | tail -36 > gpurun_out/mce_ab_steps.txt
TOPN=30 bash tools/kstats_cmd.sh c4mce python tools/lstm_bench.py --loss mce --steps 20 > gpurun_out/c4mce_kstats.txt 2>&1
cat gpurun_out/mc_slots.txt gpurun_out/mce_ab_steps.txt gpurun_out/c4mce_kstats.txt
