#!/bin/bash
# c5w1 (one rank, graphs on / off) and the N=2 one-GPU gloo rig (path check, not a number)
cd "$(dirname "$0")/.."
py='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); s=j.get("sub",{}).get("c5w1", j)
        print(s["config"].get("hipgraph_captures"), s["config"].get("hipgraph_replays"), s["n_gpus"], "ms/step", round(s["ms_per_step"],4), "value", round(s["value"]/1e6,2), (s.get("roofline_comm") or {}).keys())'
echo "== c5w1 graphs"; timeout 300 python bench.py --subs c5w1 --steps 20 --warmup 5 --no-rooflines --no-cpu-baseline --repeats 1 2>/dev/null | python -c "$py"
echo "== c5w1 eager"; ARX_DIST_EAGER=1 timeout 300 python bench.py --subs c5w1 --steps 20 --warmup 5 --no-rooflines --no-cpu-baseline --repeats 1 2>/dev/null | python -c "$py"
for e in 0 1; do
echo "== N=2 on one GPU over gloo, eager=$e"
ARX_DIST_EAGER=${e/0/} ARX_DIST_BACKEND=gloo ARX_DIST_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 100 --warmup 10 --n-items 20000000 2>/dev/null | python -c "$py"
done
