"""cProfile of the host side of a C1-sized step loop (B=64): python tools/c1_hostprof.py"""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "a-recsys_amd")):
    sys.path.insert(0, _p)
import numpy as np, torch
from arx.hmf.hmf_model import LatentProductModel
from arx.utils.synthetic import SyntheticHMF
syn = SyntheticHMF(n_users=6040, n_items=3883, logit_size=3100, seed=0)
model = LatentProductModel(syn.n_users, syn.n_items, 32, 1, 64, 1.0, 1.0, syn.u_attr, syn.i_attr,
                           syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind, loss_function='ce')
dev = model.rt.device
rng = np.random.default_rng(0)
batches = [tuple(torch.from_numpy(a).to(dev) for a in syn.sample_batch(64, rng)) for _ in range(64)]
for k in range(50):
    model.step_async(None, *batches[k % 64], loss='ce')
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(3000):
    model.step_async(None, *batches[k % 64], loss='ce')
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
