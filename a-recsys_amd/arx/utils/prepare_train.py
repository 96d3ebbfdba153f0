"""Host-side feeders of the hot path (py3 twins of utils/prepare_train.py:7-57):
negative-pool sampler, item-frequency distribution, positive sets.  Pinned
against the importable reference by tests/golden (make_golden.py)."""
from __future__ import annotations

import numpy as np


def sample_items(items, n, p=None, replace=False):
    """prepare_train.py:7-17 -- np.random.choice(items, n, replace, p) + id->slot dict."""
    if p is not None and len(p):
        item_sampled = np.random.choice(items, n, replace=replace, p=p)
    else:
        item_sampled = np.random.choice(items, n, replace=replace)
    item_sampled_id2idx = {}
    for i, item in enumerate(item_sampled):
        item_sampled_id2idx[item] = i
    return item_sampled, item_sampled_id2idx


def item_frequency(data_tr, power):
    """prepare_train.py:19-35 -- p(item) ~ (count / total)^power, normalised."""
    item_counts = {}
    item_population = set([])
    for rec in data_tr:
        i = rec[1]
        item_counts[i] = 1 if i not in item_counts else item_counts[i] + 1
        item_population.add(i)
    item_population = list(item_population)
    counts = [item_counts[v] for v in item_population]
    count_sum = sum(counts) * 1.0
    p_item_unormalized = [np.power(c / count_sum, power) for c in counts]
    p_item_sum = sum(p_item_unormalized)
    p_item = [f / p_item_sum for f in p_item_unormalized]
    return item_population, p_item


def positive_items(data_tr, data_va):
    """prepare_train.py:37-57 -- {user: [items]} for train and validation."""
    hist, hist_va = {}, {}
    for rec in data_tr:
        hist.setdefault(rec[0], set()).add(rec[1])
    for rec in data_va:
        hist_va.setdefault(rec[0], set()).add(rec[1])
    return ({u: list(s) for u, s in hist.items()}, {u: list(s) for u, s in hist_va.items()})
