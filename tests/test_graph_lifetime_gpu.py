"""Lifetime of captured step graphs (round 6): a plan -- and with it its executable hipGraph -- may be destroyed at
ANY time by Python's cyclic collector, e.g. in the middle of ANOTHER model's replay loop.  On ROCm 7.0 an
hipGraphExecDestroy beside in-flight launches of another executable graph made that graph's next hipGraphLaunch
segfault (bench.py's default run, 2 of 3); arx_graph_destroy / arx_graph_feeds_destroy therefore synchronise the device
first.  This test replays one model's captured step while a second model's plans are collected under it and checks
the surviving model against an undisturbed twin, bit for bit."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(syn, params, d, B, S):
    from arx.hmf.hmf_model import LatentProductModel
    m = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.5, 1.0, syn.u_attr, syn.i_attr,
                           syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind, loss_function='mw', n_sampled=S,
                           params=params)
    pos = syn.positives_dict()
    m.prepare_warp(pos, pos)
    return m


def test_plan_collected_under_another_models_replay_loop(dev):
    import torch
    from arx.utils.synthetic import SyntheticHMF
    d, B, S = 64, 32, 128
    syn = SyntheticHMF(n_users=300, n_items=400, item_mulhot=True, mulhot_vocab=100, avg_len=5, max_len=12, seed=0)
    params = syn.glorot_params(d, seed=1, scale=0.5)
    rng = np.random.default_rng(0)
    pool = syn.sample_pool(S, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    batches = [syn.sample_batch(B, rng) for _ in range(8)]
    losses = []
    for disturb in (False, True):
        a = _model(syn, params, d, B, S)
        victims = [_model(syn, params, d, B, S) for _ in range(3)] if disturb else []
        for v in victims:                              # eager step, capture, one replay: three live graphs
            for k in range(3):
                v.step(None, list(batches[k][0]), list(batches[k][1]), None, pool if k == 0 else None, id2idx, loss='mw')
        n = 120
        buf = torch.zeros(n, dtype=torch.float32, device=a.rt.device)
        for k in range(n):
            u, i = batches[k % len(batches)]
            node = a.step_async(None, list(u), list(i), None, pool if k == 0 else None, id2idx, loss='mw')
            buf[k:k + 1].copy_(node.read().reshape(1))          # (device-to-device: the loop never waits for the GPU)
            if disturb and k in (20, 50, 80) and victims:
                victims.pop()                          # a model dies in the middle of the loop ...
                gc.collect()                           # ... and its plans, graphs included, are destroyed right here
        torch.cuda.synchronize()
        losses.append(buf.cpu().tolist())
        del a
        gc.collect()
    assert losses[0] == losses[1]
