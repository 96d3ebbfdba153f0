"""C4: LSTM sequence model (lstm/seqModel.py) on the HIP path -- targets/s.

BASELINE.json configs[3]: d = h = 64, L = 50, 1 M items, S = 1024 sampled negatives ('mw'),
use_concat=False, clip 5.0, Adagrad lr 0.5.  One step = L time steps x B
sequences: lookups, persistent LSTM fwd/bwd (MFMA gate GEMM), scorer GEMMs over [L*B, S],
loss, clip-by-global-norm, sparse + dense Adagrad.  Metric: sum(target weights) / wall
(lstm/run.py:466-470).

usage: python tools/lstm_bench.py [--batch 1024] [--steps 30] [--warmup 5] [--n-items 1000000]
Prints one JSON line.
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
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-items", type=int, default=1000000)
    ap.add_argument("--n-users", type=int, default=1000000)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--len", type=int, default=50)
    ap.add_argument("--n-sampled", type=int, default=1024)
    ap.add_argument("--loss", default="mw", choices=["mw", "mce"])
    args = ap.parse_args()
    torch.cuda.set_device(0)
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.lstm.seqModel import SeqModel
    from arx.utils.synthetic import SyntheticHMF

    B, L, S, size = args.batch, args.len, args.n_sampled, args.size
    t0 = time.time()
    syn = SyntheticHMF(n_users=args.n_users, n_items=args.n_items, permute_logits=False, seed=0)
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    i2l = syn.item2logit[:args.n_items]
    START = args.n_items
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, L, False, None, syn.logit_ind2item_ind)
    model = SeqModel([L], size, 1, 5.0, B, 0.5, 0.99, emb, loss=args.loss, use_concat=False,
                     START_ID=START)
    emb.prepare_warp(syn.positives_csr(), syn.positives_csr())
    dev = model.rt.device
    rng = np.random.default_rng(1)
    total = args.steps + args.warmup
    nb = min(total, 8)
    batches = []
    wsum = 0.0
    for _ in range(nb):
        users = rng.integers(0, args.n_users, size=B).astype(np.int32)
        tg = np.stack([syn.sample_batch(B, rng)[1] for _ in range(L)], 0).astype(np.int32)
        inp = np.concatenate([np.full((1, B), START, dtype=np.int32), tg[:-1]], 0)
        lens = rng.integers(10, L + 1, size=B)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float32)
        batches.append((torch.from_numpy(users).to(dev), torch.from_numpy(inp).to(dev),
                        torch.from_numpy(tg).to(dev), torch.from_numpy(w).to(dev), float(w.sum())))
    pool = torch.from_numpy(syn.sample_pool(S, rng)).to(dev)
    setup_s = time.time() - t0

    def run(k0, k1):
        tot = 0.0
        node = None
        for k in range(k0, k1):
            u, i, t, w, ws = batches[k % nb]
            node = model.step_async(None, u, i, t, w, 0, pool if k == 0 else None, None)
            tot += ws
        return tot, node

    run(0, args.warmup)
    torch.cuda.synchronize()
    t1 = time.time()
    tot_w, node = run(args.warmup, total)
    torch.cuda.synchronize()
    wall = time.time() - t1
    loss = float(node.read().item()) / max(batches[(total - 1) % nb][4], 1.0)   # per target
    print(json.dumps({
        "metric": "LSTM seqModel training targets/s", "value": tot_w / wall, "unit": "targets/s",
        "ms_per_step": 1e3 * wall / args.steps, "steps": args.steps, "warmup": args.warmup,
        "config": {"workload": "C4: LSTM d=h=%d, L=%d, B=%d sequences, %d items, S=%d 'mw', clip 5.0"
                               % (size, L, B, args.n_items, S),
                   "timestep_rows_per_s": L * B * args.steps / wall, "final_loss": loss,
                   "setup_s": setup_s}}))


if __name__ == "__main__":
    main()
