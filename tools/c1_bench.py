"""C1: BASELINE.json configs[0] -- MovieLens-1m-shaped HMF (6040 users, 3883 items, V = 3100
candidate items, d = 32, B = 64, cross-entropy over the full V, no attributes, lr 1.0), the
reference's own CPU-runnable case (examples/run_hmf.sh 32 1 False 100 False).

Times the HIP path (hipGraph replay, device-resident batches) and the oracle's fp32 restatement
of the TF1 CPU graph on the same shapes, and reports the relative loss error of the first
steps.  usage: python tools/c1_bench.py [--steps 2000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "a-recsys_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--cpu-steps", type=int, default=200)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.synthetic import SyntheticHMF
    from oracle import ref_graph as rg
    d, B = 32, 64
    syn = SyntheticHMF(n_users=6040, n_items=3883, logit_size=3100, seed=0)
    params = syn.glorot_params(d, seed=1, scale=0.5)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    model = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 1.0, 1.0, syn.u_attr, syn.i_attr, i2l,
                               l2i, loss_function='ce', params=params)
    ref = rg.RefLatentProductModel(d, B, 1.0, syn.u_attr, syn.i_attr, i2l, l2i, loss_function='ce',
                                   params=params, dtype=np.float32)
    dev = model.rt.device
    rng = np.random.default_rng(0)
    host = [syn.sample_batch(B, rng) for _ in range(64)]
    batches = [(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev)) for u, i in host]
    # parity of the first steps (same batches, both start from `params`)
    rel = 0.0
    pairs = []
    for k in range(3):
        u, i = host[k]
        l_ref = float(ref.step(list(u), list(i), loss='ce'))
        l_got = float(model.step(None, list(u), list(i), loss='ce'))
        rel = max(rel, abs(l_got - l_ref) / abs(l_ref))
        pairs.append((l_got, l_ref))
    for k in range(50):
        model.step_async(None, *batches[k % 64], loss='ce')
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(args.steps):
        model.step_async(None, *batches[k % 64], loss='ce')
    torch.cuda.synchronize()
    gpu_wall = time.time() - t0
    t0 = time.time()
    for k in range(args.cpu_steps):
        u, i = host[k % 64]
        ref.step(list(u), list(i), loss='ce')
    cpu_wall = time.time() - t0
    try:
        import threadpoolctl
        th = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        th = os.cpu_count() or 1
    print(json.dumps({
        "config": "C1: ML-1m shape, 6040 users, 3883 items, V=3100, d=32, B=64, ce, lr 1.0",
        "gpu_interactions_per_s": B * args.steps / gpu_wall, "gpu_us_per_step": 1e6 * gpu_wall / args.steps,
        "cpu_restatement_interactions_per_s": B * args.cpu_steps / cpu_wall,
        "cpu_ms_per_step": 1e3 * cpu_wall / args.cpu_steps, "cpu_threads": int(th),
        "loss_rel_err_first3_steps_vs_fp32_restatement": rel, "losses_gpu_vs_cpu": pairs}))


if __name__ == "__main__":
    main()
