"""K7 of a C3 / C2 step alone, through the C ABI: the fused one-hot pass over (item id table, user
table) with the item's multi-hot table riding on it (arx_sparse_adagrad_cat_multi_bags), phase 1
(grouping / sorts: ids only) and phase 2 (apply) timed separately as hipGraph replays.

usage: python tools/k7grp_bench.py [B ...]        K7_MODE=c2: the one-hot pass alone
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np
import torch

from arx import ops
from arx.utils.synthetic import SyntheticHMF
from k7bench import timed


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [16384]
    dev = torch.device('cuda:0')
    d, S = 128, 1024
    n_items = n_users = 1000000
    syn = SyntheticHMF(n_users=1000, n_items=n_items, item_mulhot=True, permute_logits=False, seed=0, n_pos=4)
    ia = syn.i_attr
    vals = torch.from_numpy(np.asarray(ia.features_mulhot[0], dtype=np.int32)).to(dev)
    starts = torch.from_numpy(np.asarray(ia.mulhot_starts[0], dtype=np.int32)).to(dev)
    lens = torch.from_numpy(np.asarray(ia.mulhot_lengths[0], dtype=np.int32)).to(dev)
    Vf = int(ia._embedding_classes_list_mulhot[0])
    ln = np.asarray(ia.mulhot_lengths[0])
    max_len = int(ln.max())
    rng = np.random.default_rng(1)
    mode = os.environ.get('K7_MODE', 'c3')
    for B in Bs:
        items = rng.choice(n_items, size=B, p=syn.p_item).astype(np.int32)
        pool = rng.choice(n_items, size=S, replace=False, p=syn.p_item).astype(np.int32)
        users = rng.integers(0, n_users, size=B).astype(np.int32)

        def tab(V):
            return (torch.randn(V, d, device=dev) * 0.05, torch.full((V, d), 0.1, device=dev),
                    torch.zeros(V, dtype=torch.int32, device=dev))
        T_it, T_us, T_bag = tab(n_items), tab(n_users), tab(Vf)
        G = torch.randn(2 * B + S, d, device=dev) * 1e-3
        lr = torch.tensor([0.1], device=dev)
        t_items, t_pool, t_users = [torch.from_numpy(x).to(dev) for x in (items, pool, users)]
        args = ops.MultiCatArgs([(T_it[0], T_it[1], None, None, T_it[2]), (T_us[0], T_us[1], None, None, T_us[2])],
                                [(1, None, t_users, 0, 1.0), (0, None, t_items, B, 0.5), (0, None, t_pool, 2 * B, 0.5)])
        n = args.total
        kb_ = torch.empty(n, dtype=torch.int32, device=dev)
        sb_ = torch.empty(n, dtype=torch.int32, device=dev)
        cb_ = torch.empty(n, dtype=torch.float32, device=dev)
        ws, bws = ops.Workspace(dev), ops.Workspace(dev)

        def run(phase):
            if mode == 'c2':
                ops.sparse_adagrad_cat_multi(args, G, None, lr, kb_, sb_, cb_, ws, phase=phase)
            else:
                ops.sparse_adagrad_cat_multi_bags(args, G, None, lr, kb_, sb_, cb_, ws, T_bag[0], T_bag[1], None, None,
                                                  vals, starts, lens, max_len, bws, phase=phase, bag_aux_cnt=T_bag[2])
        uniq_it = np.unique(np.concatenate([items, pool]))
        st_, va_ = np.asarray(ia.mulhot_starts[0]), np.asarray(ia.features_mulhot[0])
        alltok = np.concatenate([va_[st_[e]:st_[e] + ln[e]] for e in uniq_it])
        ntok_rows = len(np.unique(alltok))
        ncontrib = int(ln[items].sum() + ln[pool].sum())
        rows = len(np.unique(users)) + len(uniq_it) + (ntok_rows if mode != 'c2' else 0)
        alg = rows * (16 * d + 4) + n * 4 * d + (n + (ncontrib if mode != 'c2' else 0)) * 12
        us_all = timed(lambda: run(3))
        us_p1 = timed(lambda: run(1))
        # (phase 2 alone can only be replayed when every pass uses the run-centric apply: the window
        # apply appends to run lists that the sort's first launch resets)
        if mode == 'c2':
            run(1)
            us_p2 = timed(lambda: run(2))
        else:
            us_p2 = us_all - us_p1
        print('%s B=%6d one-hot %6d  distinct items %6d  token rows %6d  token contributions %7d (%d after the per-item merge)  alg %6.1f MB'
              % (mode, B, n, len(uniq_it), ntok_rows, ncontrib, len(alltok), alg / 1e6))
        print('   all %7.1f us  %5.0f GB/s | phase 1 %7.1f us | phase 2 (apply) %7.1f us  %5.0f GB/s = %.3f of 8 TB/s'
              % (us_all, alg / us_all / 1e3, us_p1, us_p2, alg / us_p2 / 1e3, alg / us_p2 / 1e3 / 8000), flush=True)


if __name__ == '__main__':
    main()
