"""Ranking metrics P/R/MAP/NDCG@{2,5,10,20,30} of the recommend path
(py3 twin of utils/eval_metrics.py:3-74; "next" #4 of SURVEY 8f).  Pinned
against the importable reference through tests/golden/reference_helpers.json."""
from __future__ import annotations

import numpy as np

_N_MAX = 30   # the reference truncates the NDCG discounts at 30 (eval_metrics.py:61)


def _per_user(pred, truth, Ns):
    truth = set(truth)
    hit = np.asarray([1.0 if r in truth else 0.0 for r in pred], dtype=np.float64)
    l, n_t = len(hit), len(truth)
    z = [0.0] * len(Ns)
    if l == 0:
        return {'prec': z, 'recall': z, 'map': z, 'ndcg': z}
    cum = np.cumsum(hit)
    at = [min(n, l) - 1 for n in Ns]
    prec = [cum[a] / (a + 1) for a in at]
    rec = [cum[a] / n_t for a in at]
    ap = np.cumsum(hit * cum / np.arange(1, l + 1))
    mapv = [ap[a] / min(min(n_t, n), l) for a, n in zip(at, Ns)]
    n_max = min(l, _N_MAX)
    disc = 1.0 / np.log2(2.0 + np.arange(n_max))
    dcg = np.cumsum(hit[:n_max] * disc)
    ideal = np.where(np.arange(n_max) < n_t, disc, 0.0)
    idcg = np.cumsum(ideal)
    ndcg = [dcg[a] / idcg[a] for a in at]
    return {'prec': prec, 'recall': rec, 'map': mapv, 'ndcg': ndcg}


def metrics(X, T, Ns=(2, 5, 10, 20, 30), metrics=('prec', 'recall', 'map', 'ndcg')):
    """X: {user: ranked item list}; T: {user: ground-truth items}.  Averages over
    ALL users of T (users missing from X count as zeros), like the reference."""
    Ns = list(Ns)
    res = {m: [0.0] * len(Ns) for m in metrics}
    for u, t in T.items():
        if u not in X:
            continue
        pu = _per_user(X[u], t, Ns)
        for m in metrics:
            for i in range(len(Ns)):
                res[m][i] += pu[m][i]
    n_users = float(len(T))
    return {m: [v / n_users for v in res[m]] for m in metrics}
