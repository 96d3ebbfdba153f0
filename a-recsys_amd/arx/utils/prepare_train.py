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


class DeviceSampler(object):
    """On-device twin of sample_items(items, n, p, replace=False) for large item sets
    (SURVEY 8f #1): same distribution (sequential weighted draws without replacement), drawn as
    an exponential race by arx_sample_wor -- not numpy's random stream.  `items` are the global
    item ids the positions of `p` refer to (item_frequency's item_population)."""

    def __init__(self, items, p, device=None, seed=0):
        import torch
        from .. import ops
        self._ops = ops
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.items = torch.as_tensor(np.asarray(items, dtype=np.int32)).to(dev)
        pw = np.ones(len(items), dtype=np.float32) if p is None or not len(p) else np.asarray(p, dtype=np.float32)
        self.w = torch.from_numpy(np.ascontiguousarray(pw)).to(dev)
        self.ws = ops.Workspace(dev)
        self.seed = int(seed)
        self.counter = 0

    def sample(self, n, out=None):
        """-> int32 device tensor [n] of item ids (draw order).  The id->slot map the reference
        builds next (prepare_train.py:12-16) is the model's device slot map (update_sampled_pool)."""
        import torch
        pos = torch.empty((n,), dtype=torch.int32, device=self.w.device)
        self._ops.sample_wor(self.w, n, self.seed, self.counter, pos, self.ws)
        self.counter += 1
        ids = self.items[pos.long()]
        if out is not None:
            out.copy_(ids)
            return out
        return ids
