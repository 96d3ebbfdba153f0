"""cProfile of the host side of host-fed C3 steps (where do the ~0.2 ms of enqueue per step go)."""
import cProfile
import pstats
import sys
import numpy as np
import torch
sys.path.insert(0, "a-recsys_amd")
sys.path.insert(0, ".")
from arx.hmf.hmf_model import LatentProductModel
from arx.utils.synthetic import SyntheticHMF
import bench

B, S, d = 16384, 1024, 128
label, kw = bench.WORKLOADS["c3"]
syn = SyntheticHMF(n_users=1000000, n_items=1000000, permute_logits=False, seed=0, zipf_items=1.0, **kw)
model = LatentProductModel(1000000, 1000000, d, 1, B, 0.1, 1.0, syn.u_attr, syn.i_attr, syn.item2logit[:1000000],
                           syn.logit_ind2item_ind, loss_function='mw', n_sampled=S, use_graph=True)
model.prepare_warp(syn.positives_csr(), syn.positives_csr())
rng = np.random.default_rng(1)
bs = [tuple(np.ascontiguousarray(x, dtype=np.int32) for x in syn.sample_batch(B, rng)) for _ in range(16)]
pool = torch.from_numpy(syn.sample_pool(S, rng).astype(np.int32)).to(model.rt.device)
for k in range(20):
    model.step_async(None, bs[k % 16][0], bs[k % 16][1], None, pool if k == 0 else None, None, loss='mw')
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(300):
    model.step_async(None, bs[k % 16][0], bs[k % 16][1], None, None, None, loss='mw')
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
# the synchronous step() of the reference API (a float back per step): no back-pressure, the host costs are the true ones
import time
for mode in ("host", "dev"):
    if mode == "dev":
        bs = [tuple(torch.from_numpy(x).to(model.rt.device) for x in b) for b in bs]
    for k in range(10):
        model.step(None, bs[k % 16][0], bs[k % 16][1], None, None, None, loss='mw')
    t0 = time.time()
    for k in range(300):
        model.step(None, bs[k % 16][0], bs[k % 16][1], None, None, None, loss='mw')
    print("synchronous step(), ids from %s: %.1f us per step" % (mode, (time.time() - t0) / 300 * 1e6))
    t0 = time.time()
    for k in range(300):
        model.step_async(None, bs[k % 16][0], bs[k % 16][1], None, None, None, loss='mw')
    torch.cuda.synchronize()
    print("step_async, ids from %s: %.1f us per step" % (mode, (time.time() - t0) / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for k in range(300):
    model.step(None, bs[k % 16][0], bs[k % 16][1], None, None, None, loss='mw')
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
