"""K7 on multi-hot lookups with the REAL bag structure (C3: Zipf-popular target items + the
pool, ~20 category tokens per item over a 100 k-row table): contribution-level pass
(arx_bag_expand_padded + arx_sparse_adagrad) vs the two-stage merge (arx_sparse_adagrad_bags).

usage: python tools/k7bench.py [B ...]     (run on the GPU box; default B = 16384 65536)
Prints event-timed us per call (hipGraph replay) and algorithmic GB/s
((16d+4)/unique row + 4d/source row + 12/contribution).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np
import torch

from arx import ops
from arx.utils.synthetic import SyntheticHMF


def timed(call, iters=30):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    g = ops.CapturedGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g.begin()
        try:
            call()
        finally:
            g.end()
    torch.cuda.current_stream().wait_stream(side)
    g.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [16384, 65536]
    dev = torch.device('cuda:0')
    d, S = 128, 1024
    syn = SyntheticHMF(n_users=1000, n_items=1000000, item_mulhot=True, permute_logits=False, seed=0, n_pos=4)
    ia = syn.i_attr
    vals = torch.from_numpy(np.asarray(ia.features_mulhot[0], dtype=np.int32)).to(dev)
    starts = torch.from_numpy(np.asarray(ia.mulhot_starts[0], dtype=np.int32)).to(dev)
    lens = torch.from_numpy(np.asarray(ia.mulhot_lengths[0], dtype=np.int32)).to(dev)
    Vf = int(ia._embedding_classes_list_mulhot[0])
    max_len = int(np.asarray(ia.mulhot_lengths[0]).max())
    rng = np.random.default_rng(1)
    for B in Bs:
        items = rng.choice(syn.n_items, size=B, p=syn.p_item).astype(np.int32)
        pool = rng.choice(syn.n_items, size=S, replace=False, p=syn.p_item).astype(np.int32)
        E = torch.randn(Vf, d, device=dev) * 0.05
        acc = torch.full((Vf, d), 0.1, device=dev)
        G = torch.randn(B + S, d, device=dev) * 1e-3
        lr = torch.tensor([0.1], device=dev)
        t_items, t_pool = torch.from_numpy(items).to(dev), torch.from_numpy(pool).to(dev)
        ln = np.asarray(ia.mulhot_lengths[0])
        ncontrib = int(ln[items].sum() + ln[pool].sum())
        alltok = np.concatenate([np.asarray(ia.features_mulhot[0])[np.asarray(ia.mulhot_starts[0])[e]:
                                 np.asarray(ia.mulhot_starts[0])[e] + ln[e]] for e in np.unique(np.concatenate([items, pool]))])
        uniq = len(np.unique(alltok))
        uniq_items = len(np.unique(np.concatenate([items, pool])))
        alg = uniq * (16 * d + 4) + (B + S) * 4 * d + ncontrib * 12
        # --- contribution-level pass (round-1 path)
        cap = (B + S) * max_len
        ks = torch.empty(cap, dtype=torch.int32, device=dev)
        ss = torch.empty(cap, dtype=torch.int32, device=dev)
        cs = torch.empty(cap, dtype=torch.float32, device=dev)
        ws = ops.Workspace(dev)

        def old():
            ops.bag_expand_padded(vals, starts, lens, t_items, max_len, 0, 0.5, ks[:B * max_len], ss[:B * max_len],
                                  cs[:B * max_len])
            ops.bag_expand_padded(vals, starts, lens, t_pool, max_len, B, 0.5, ks[B * max_len:], ss[B * max_len:],
                                  cs[B * max_len:])
            ops.sparse_adagrad(E, acc, None, None, ks, ss, cs, G, None, lr, ws, n=cap)
        us_old = timed(old)
        # --- two-stage merge
        args = ops.BagSiteArgs([(t_items, 0, 0.5), (t_pool, B, 0.5)], max_len)
        ws2 = ops.Workspace(dev)

        def new(phase=3):
            ops.sparse_adagrad_bags(E, acc, None, None, vals, starts, lens, args, G, None, lr, ws2, phase=phase)
        us_new = timed(new)
        us_p1 = timed(lambda: new(1))          # (phase 2 alone cannot be replayed: the sort's first
        us_p2 = us_new - us_p1                 #  launch resets the run lists the apply appends to)
        print('B=%6d contributions=%8d unique rows=%6d unique items=%6d  alg %6.1f MB' %
              (B, ncontrib, uniq, uniq_items, alg / 1e6))
        print('   contribution-level: %7.1f us  %6.0f GB/s   merged: %7.1f us  %6.0f GB/s  (merge + apply = total - sorts: %7.1f us  %6.0f GB/s)'
              % (us_old, alg / us_old / 1e3, us_new, alg / us_new / 1e3, us_p2, alg / us_p2 / 1e3), flush=True)


if __name__ == '__main__':
    main()
