"""Fused against materialising 'mce' through the sequence model, step by step on the same batches
(ARX_MCE_FUSED is read when a plan is built).  Prints the per-step losses of both."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "a-recsys_amd")):
    sys.path.insert(0, _p)
import numpy as np
import torch


def run(fused, steps=35):
    os.environ['ARX_MCE_FUSED'] = '1' if fused else '0'
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.lstm.seqModel import SeqModel
    from arx.utils.synthetic import SyntheticHMF
    N, B, L, S, size = 100000, 1024, 50, 1024, 64
    syn = SyntheticHMF(n_users=N, n_items=N, permute_logits=False, seed=0)
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, L, False, None, syn.logit_ind2item_ind)
    model = SeqModel([L], size, 1, 5.0, B, 0.5, 0.99, emb, loss='mce', use_concat=False, START_ID=N)
    emb.prepare_warp(syn.positives_csr(), syn.positives_csr())
    d_ = model.rt.device
    rng = np.random.default_rng(1)
    pool = syn.sample_pool(S, rng).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(d_)
    out = []
    for step in range(steps):
        users = rng.integers(0, N, size=B).astype(np.int32)
        tg = np.stack([syn.sample_batch(B, rng)[1] for _ in range(L)], 0).astype(np.int32)
        inp = np.concatenate([np.full((1, B), N, dtype=np.int32), tg[:-1]], 0)
        lens = rng.integers(10, L + 1, size=B)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float32)
        ps = pool if step == 0 else None
        out.append(float(model.step(None, t(users), t(inp), t(tg), t(w), 0, t(ps) if ps is not None else None, None)))
    return out


a, b = run(True), run(False)
for i, (x, y) in enumerate(zip(a, b)):
    print("step %2d fused %.7g materialising %.7g rel %.2e" % (i, x, y, abs(x - y) / abs(y)))
