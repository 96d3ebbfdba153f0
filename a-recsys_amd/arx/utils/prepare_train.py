"""Host-side feeders of the hot path (py3 twins of utils/prepare_train.py:7-57):
negative-pool sampler, item-frequency distribution, positive sets.  Pinned
against the importable reference by tests/golden (make_golden.py)."""
from __future__ import annotations

import numpy as np


def sample_items(items, n, p=None, replace=False):
    """prepare_train.py:7-17: draw the step's shared negative pool with numpy's legacy global
    stream (np.random.choice -- the goldens pin the exact ids for a given np.random.seed) and
    return it with its id -> pool-slot dictionary (later duplicates win, as in the reference)."""
    weights = p if (p is not None and len(p)) else None
    pool = np.random.choice(items, n, replace=replace, p=weights)
    return pool, {item: slot for slot, item in enumerate(pool)}


def item_frequency(data_tr, power):
    """prepare_train.py:19-35: the sampler's distribution p(item) ~ (count / total) ** power over the
    items seen in training, normalised.  Returns (item_population, p_item) in the iteration
    order of a Python set of the item ids (that order is part of the reference's behaviour:
    sample_items indexes into it).  The float arithmetic keeps the reference's operation order
    (per-item power of count/total, left-to-right sum, per-item divide) so that the goldens
    match bit for bit."""
    seen = {}
    for rec in data_tr:
        seen[rec[1]] = seen.get(rec[1], 0) + 1
    item_population = list(set(seen))
    total = float(sum(seen[i] for i in item_population))
    raw = [np.power(seen[i] / total, power) for i in item_population]
    norm = sum(raw)
    return item_population, [w / norm for w in raw]


def positive_items(data_tr, data_va):
    """prepare_train.py:37-57: {user: [distinct items]} of the training and of the validation
    interactions (the sets the loss masks take out of the negatives)."""
    def by_user(records):
        sets = {}
        for rec in records:
            sets.setdefault(rec[0], set()).add(rec[1])
        return {user: list(items) for user, items in sets.items()}
    return by_user(data_tr), by_user(data_va)


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
        if isinstance(items, torch.Tensor):           # device-resident population / weights are taken as they are
            self.items = items.to(dev, torch.int32)
        else:
            self.items = torch.as_tensor(np.asarray(items, dtype=np.int32)).to(dev)
        if isinstance(p, torch.Tensor):
            self.w = p.to(dev, torch.float32).contiguous()
        else:
            pw = np.ones(len(items), dtype=np.float32) if p is None or not len(p) else np.asarray(p, dtype=np.float32)
            self.w = torch.from_numpy(np.ascontiguousarray(pw)).to(dev)
        self.ws = ops.Workspace(dev)
        self.seed = int(seed)
        self.counter = 0
        # large populations: keys that cannot be among the n smallest are dropped before the sort
        # (arx_sample_wor_capped); the draw is the un-capped one.  The cap depends on n: _cap_for
        self._caps = {}
        self._npos = None

    @classmethod
    def from_interactions(cls, item_ids, n_items, power=0.5, device=None, seed=0):
        """item_frequency (prepare_train.py:19-35) + the sampler over it, on device: `item_ids` =
        the item column of the training interactions (device int32 tensor or array).  The
        population is ALL ids 0..n_items-1 with weight (count/total)^power, 0 for unseen items
        (the reference lists only the seen ones, in set order; the draw law is the same)."""
        import torch
        from .. import ops
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        ids = item_ids if isinstance(item_ids, torch.Tensor) else torch.as_tensor(np.asarray(item_ids, dtype=np.int32))
        ids = ids.to(dev, torch.int32).contiguous()
        counts = torch.zeros(n_items, dtype=torch.int32, device=dev)
        w = torch.empty(n_items, dtype=torch.float32, device=dev)
        ops.item_frequency(ids, n_items, counts, w, total=int(ids.shape[0]), power=power)
        s = cls(torch.arange(n_items, dtype=torch.int32, device=dev), w, device=dev, seed=seed)
        s.counts = counts
        return s

    def _cap_for(self, n):
        """Key threshold t of the capped race for draws of n items: an item survives the cap with
        probability 1 - exp(-w t), so t is chosen such that the EXPECTED number of survivors,
        sum_i (1 - exp(-w_i t)), is 3 n (the count is a sum of independent indicators, standard deviation
        <= sqrt(3 n): fewer than n of 3 n expected is ~37 sigma away at n = 1024; and no more than 4 n -- the
        power of two the survivors' one-workgroup sort pads to: 3072 + 5 sigma < 4096.  Round 3 aimed at 7.5 n:
        the sort of 8192 padded entries and the list appends were 150 of the redraw's 166 us at 1 M items).
        `8 n / sum(w)` -- round 2 -- only equals that while
        w t << 1 for every item: with heavy-tailed weights (w ~ rank^-1.5, 1 M items, n = 1000) it
        let ~630 keys through, the draw came back short and the missing positions indexed
        items[-1].  0.0 = no cap (small populations, or too few positive weights for one to pay).
        Computed once per n (a few reductions over the weights, with host reads: set-up work)."""
        import torch
        if n in self._caps:
            return self._caps[n]
        N = int(self.w.numel())
        if self._npos is None:
            # (set-up: one pass + host read) NaN / inf weights would collapse the cap's bisection to ~0 and
            # the whole pool to one repeated item (advisor, round 3)
            if not bool(torch.isfinite(self.w).all().item()) or bool((self.w < 0).any().item()):
                raise ValueError("DeviceSampler: weights must be finite and >= 0")
            self._npos = int((self.w > 0).sum().item())
        if self._npos < n:
            raise ValueError("DeviceSampler.sample(%d): only %d items have a positive weight" % (n, self._npos))
        cap = 0.0
        target = 3.0 * n
        if N > (1 << 16) and 16 * n < N and self._npos >= 2 * target:
            w = self.w.clamp(min=0)

            def expected(t):
                return float((-torch.expm1(w * (-t))).sum(dtype=torch.float64).item())
            hi = target / max(float(w.sum(dtype=torch.float64).item()), 1e-300)
            while expected(hi) < target:          # (terminates: expected(inf) = npos >= 2 * target)
                hi *= 2.0
            lo = 0.0
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                if expected(mid) < target:
                    lo = mid
                else:
                    hi = mid
            cap = hi
        self._caps[n] = cap
        return cap

    def sample_with_keys(self, n):
        """-> (item ids int32 [n], race keys float32 [n], ascending): this sampler's part of ONE draw over
        an item set sharded over several ranks (arx.dist.draw_global_pool keeps the n smallest keys of
        the union).  Where the shard has fewer than n items of positive weight the tail is
        (id -1, key +inf)."""
        import torch
        if self._npos is None:
            self._npos = int((self.w > 0).sum().item())
        m = min(int(n), self._npos)
        dev = self.w.device
        ids = torch.full((n,), -1, dtype=torch.int32, device=dev)
        keys = torch.full((n,), float('inf'), dtype=torch.float32, device=dev)
        if m > 0:
            pos = torch.empty((m,), dtype=torch.int32, device=dev)
            self._ops.sample_wor(self.w, m, self.seed, self.counter, pos, self.ws, key_cap=self._cap_for(m),
                                 out_keys=keys[:m])
            # a short capped draw (compact list overflow / fewer than m keys under the cap) leaves -1 positions:
            # they stay (id -1, key +inf) instead of silently becoming items[-1] (advisor, round 3)
            ok = pos >= 0
            ids[:m] = torch.where(ok, self.items[pos.clamp(min=0).long()], torch.full_like(pos, -1))
            keys[:m] = torch.where(ok, keys[:m], torch.full_like(keys[:m], float('inf')))
        self.counter += 1
        return ids, keys

    def sample(self, n, out=None):
        """-> int32 device tensor [n] of item ids (draw order).  The id->slot map the reference
        builds next (prepare_train.py:12-16) is the model's device slot map (update_sampled_pool)."""
        import torch
        pos = torch.empty((n,), dtype=torch.int32, device=self.w.device)
        cap = self._cap_for(int(n))
        self._ops.sample_wor(self.w, n, self.seed, self.counter, pos, self.ws, key_cap=cap)
        self.counter += 1
        # positions the capped race could not fill come back as -1: they become id -1 instead of letting -1 index
        # items[-1] (advisor, round 3).  A negative id is an EMPTY slot downstream (advisor, round 4): the step's
        # lookup launch writes a zero row for it, the slot map skips it and K7's key builders drop it -- no read of
        # cat_map[-1] / E[-1]; the sharded pool (dist.set_pool, draw_global_pool) rejects it with a ValueError.
        if out is None:
            out = torch.empty((n,), dtype=torch.int32, device=self.w.device)
        self._ops.take_i32(self.items, pos, out, fill=-1)         # (one launch; the torch form was six)
        return out
