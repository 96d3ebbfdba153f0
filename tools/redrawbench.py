"""Where a pool redraw's time goes (C3 bench workload): sampler, staging the pool, the step that follows.
usage: python tools/redrawbench.py"""
import sys, os, time
sys.path.insert(0, '/root/repo/a-recsys_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
args = bench.parse() if hasattr(bench, 'parse') else None
args.n_items = 1000000
from arx.hmf.hmf_model import LatentProductModel
from arx.utils.synthetic import SyntheticHMF
from arx.utils.prepare_train import DeviceSampler
B, S, d = 16384, 1024, 128
syn = SyntheticHMF(n_users=1000000, n_items=1000000, permute_logits=False, seed=0, zipf_items=1.05, item_mulhot=True)
model = LatentProductModel(1000000, 1000000, d, 1, B, 0.1, 1.0, syn.u_attr, syn.i_attr, syn.item2logit[:1000000],
                           syn.logit_ind2item_ind, loss_function='mw', n_sampled=S, use_graph=True)
model.prepare_warp(syn.positives_csr(), syn.positives_csr())
dev = model.rt.device
sampler = DeviceSampler(syn.item_population, syn.p_sample, device=dev, seed=1)
rng = np.random.default_rng(1)
bs = [tuple(torch.from_numpy(x).to(dev) for x in syn.sample_batch(B, rng)) for _ in range(8)]
def sync(): torch.cuda.synchronize()
pool = sampler.sample(S)
for k in range(30):
    model.step_async(None, bs[k % 8][0], bs[k % 8][1], None, pool if k == 0 else None, None, loss='mw')
sync()
def t(fn, n=10):
    sync(); t0 = time.time()
    for _ in range(n): fn()
    sync(); return (time.time() - t0) / n * 1e6
print('plain step            %.0f us' % t(lambda: model.step_async(None, bs[0][0], bs[0][1], None, None, None, loss='mw'), 50))
print('sampler.sample        %.0f us' % t(lambda: sampler.sample(S)))
p2 = sampler.sample(S)
print('step with a new pool  %.0f us' % t(lambda: model.step_async(None, bs[0][0], bs[0][1], None, p2, None, loss='mw')))
print('update_sampled_pool   %.0f us' % t(lambda: model.att_emb.update_sampled_pool(p2) if hasattr(model, 'att_emb') else None))
