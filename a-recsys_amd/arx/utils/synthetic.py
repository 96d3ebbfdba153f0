"""Seeded synthetic workloads in the reference's attribute layout.

Produces `Attributes` objects exactly as utils/preprocess.py:169-326 would
(vocab rows 0=_UNK, 1=_START; categorical maps int[N+1] ending with the START
row; multi-hot CSR values/starts/lengths with the trailing START bag; the
logit-ordered `*_tr` copies), an item<->logit index map, per-user positive sets
as CSR, a Zipf item popularity and an interaction sampler.  numpy only.
"""
from __future__ import annotations

import numpy as np

from ..attributes.attribute import Attributes

UNK_ID, START_ID = 0, 1


def _zipf_probs(n, a):
    p = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), a)
    return p / p.sum()


def _cat_feature(n):
    """id feature: entity n -> vocab row n+2; row N is the START entity."""
    m = np.arange(n + 1, dtype=np.int32) + 2
    m[n] = START_ID
    return m, n + 2


def _mulhot_feature(rng, n, vocab, avg_len, max_len, zipf_a, with_id_token=False):
    """with_id_token: MIX layout (comb_attribute.py:100-148: ONE bag per entity holding its id
    token 'id<n>' and its attribute tokens, one shared table): the bag of entity e starts with
    the id token row 2 + e, the attribute tokens live behind the n id rows."""
    lens = np.clip(rng.poisson(avg_len, size=n), 1, max_len - (1 if with_id_token else 0)).astype(np.int32)
    total = int(lens.sum())
    p = _zipf_probs(vocab, zipf_a)
    perm = rng.permutation(vocab)
    vals = (perm[rng.choice(vocab, size=total, p=p)] + 2).astype(np.int32)
    rows = vocab + 2
    if with_id_token:
        vals = vals + n
        out = np.empty(total + n, dtype=np.int32)
        first = np.concatenate([[0], np.cumsum(lens + 1)[:-1]])
        is_id = np.zeros(total + n, dtype=bool)
        is_id[first] = True
        out[is_id] = np.arange(n, dtype=np.int32) + 2
        out[~is_id] = vals
        vals, lens, rows = out, lens + 1, vocab + n + 2
    values = np.concatenate([vals, np.array([START_ID], dtype=np.int32)])
    lengths = np.concatenate([lens, np.array([1], dtype=np.int32)])
    starts = np.zeros(n + 2, dtype=np.int32)
    starts[1:] = np.cumsum(lengths)
    return values, starts, lengths, rows


class SyntheticHMF(object):
    """One synthetic HMF workload (configs C1/C2/C3/C5 of SURVEY 8d)."""

    def __init__(self, n_users, n_items, logit_size=None, item_mulhot=False, user_mulhot=False,
                 mulhot_vocab=100000, avg_len=20, max_len=64, n_pos=20, zipf_items=1.05,
                 zipf_tokens=1.0, seed=0, permute_logits=True, item_id_feature=True, item_mix=False):
        """item_mix: MIX-style items (one bag = id token + attribute tokens over one table of
        n_items + mulhot_vocab + 2 rows; implies item_mulhot, no separate id feature)."""
        if item_mix:
            item_mulhot, item_id_feature = True, False
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.n_users, self.n_items = n_users, n_items
        V = logit_size if logit_size is not None else n_items
        self.logit_size = V

        # ---- users ----
        ucat, uv = _cat_feature(n_users)
        u_mul = []
        if user_mulhot:
            u_mul.append(_mulhot_feature(rng, n_users, max(8, mulhot_vocab // 100), 4, 8, zipf_tokens))
        self.u_attr = Attributes(1, [ucat], len(u_mul), [m[0] for m in u_mul], None,
                                 [m[1] for m in u_mul], [m[2] for m in u_mul], [uv],
                                 [m[3] for m in u_mul])
        # ---- items ----
        i_cat, v_cat = [], []
        if item_id_feature:
            icat, iv = _cat_feature(n_items)
            i_cat.append(icat)
            v_cat.append(iv)
        i_mul = []
        if item_mulhot:
            i_mul.append(_mulhot_feature(rng, n_items, mulhot_vocab, avg_len, max_len, zipf_tokens,
                                         with_id_token=item_mix))
        self.i_attr = Attributes(len(i_cat), i_cat, len(i_mul), [m[0] for m in i_mul], None,
                                 [m[1] for m in i_mul], [m[2] for m in i_mul], v_cat,
                                 [m[3] for m in i_mul])
        # ---- popularity + item <-> logit maps ----
        self.item_perm = rng.permutation(n_items)
        self.p_item = np.zeros(n_items, dtype=np.float64)
        self.p_item[self.item_perm] = _zipf_probs(n_items, zipf_items)
        if V < n_items:
            logit2item = np.sort(self.item_perm[:V])           # the V most popular items
        else:
            logit2item = np.arange(n_items)
        if permute_logits:
            logit2item = logit2item[rng.permutation(len(logit2item))]
        self.logit_ind2item_ind = logit2item.astype(np.int64)
        self.item2logit = np.full(n_items + 1, -1, dtype=np.int32)
        self.item2logit[logit2item] = np.arange(V, dtype=np.int32)
        self.in_logits = self.item2logit[:n_items] >= 0
        self._set_target_prediction()
        # ---- positives: n_pos per user from the popularity law (restricted to logit items) ----
        p = self.p_item * self.in_logits
        p = p / p.sum()
        self.p_pos = p
        items = rng.choice(n_items, size=(n_users, n_pos), p=p).astype(np.int32)
        self.pos_items = items.reshape(-1)
        self.pos_ptr = (np.arange(n_users + 2, dtype=np.int64) * n_pos).astype(np.int32)
        self.pos_ptr[-1] = self.pos_ptr[-2]                      # START user: no positives
        self.n_pos = n_pos
        # sampler distribution p ~ count^0.5 (run_hmf.py:62 power, prepare_train.py:19-35)
        cnt = np.bincount(self.pos_items, minlength=n_items).astype(np.float64)
        w = np.power(cnt / max(cnt.sum(), 1.0), 0.5) * self.in_logits
        self.item_population = np.nonzero(w > 0)[0].astype(np.int64)
        self.p_sample = (w[self.item_population] / w[self.item_population].sum())

    def _set_target_prediction(self):
        ia = self.i_attr
        l2i = self.logit_ind2item_ind
        cat_tr = [np.asarray(ia.features_cat[i])[l2i].astype(np.int32) for i in range(ia.num_features_cat)]
        vals_tr, seg_tr, len_tr = [], [], []
        for i in range(ia.num_features_mulhot):
            starts = np.asarray(ia.mulhot_starts[i])[l2i]
            lens = np.asarray(ia.mulhot_lengths[i])[l2i]
            total = int(lens.sum())
            seg = np.repeat(np.arange(len(l2i), dtype=np.int32), lens)
            base = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
            idx = np.arange(total) + base
            vals_tr.append(np.asarray(ia.features_mulhot[i])[idx].astype(np.int32))
            seg_tr.append(seg)
            len_tr.append(lens.astype(np.float32).reshape(-1, 1))
        ia.set_target_prediction(cat_tr, vals_tr, seg_tr, len_tr)

    # ---- dict forms (the reference's host structures) for small configs ----
    def item_ind2logit_ind_dict(self):
        return {int(i): int(j) for j, i in enumerate(self.logit_ind2item_ind)}

    def positives_dict(self):
        return {u: self.pos_items[self.pos_ptr[u]:self.pos_ptr[u + 1]].tolist()
                for u in range(self.n_users)}

    def positives_csr(self):
        return (self.pos_ptr, self.pos_items)

    # ---- streams ----
    def sample_batch(self, B, rng=None):
        """(users, items): a uniform user and one of that user's positives."""
        rng = rng or self.rng
        users = rng.integers(0, self.n_users, size=B).astype(np.int32)
        k = rng.integers(0, self.n_pos, size=B)
        items = self.pos_items[self.pos_ptr[users] + k].astype(np.int32)
        return users, items

    def sample_pool(self, S, rng=None):
        """prepare_train.py:7-17 sample_items over item_population with p ~ count^0.5."""
        rng = rng or self.rng
        pool = rng.choice(self.item_population, size=S, replace=False, p=self.p_sample)
        return pool.astype(np.int32)

    def glorot_params(self, d, seed=0, item_output=False, scale=None):
        """Explicit initial tables keyed by the reference's variable names."""
        rng = np.random.default_rng(seed)
        out = {}

        def tab(name, V, dd):
            lim = scale if scale is not None else np.sqrt(6.0 / (V + dd))
            out[name] = rng.uniform(-lim, lim, size=(V, dd)).astype(np.float32)

        ua, ia = self.u_attr, self.i_attr
        for i in range(ua.num_features_cat):
            tab('userembed_cat_%d' % i, ua._embedding_classes_list_cat[i], d)
        for i in range(ua.num_features_mulhot):
            tab('userembed_mulhot_%d' % i, ua._embedding_classes_list_mulhot[i], d)
        prefixes = ['item'] + (['item_output'] if item_output else [])
        for pf in prefixes:
            for i in range(ia.num_features_cat):
                tab('%sembed_cat_%d' % (pf, i), ia._embedding_classes_list_cat[i], d)
                tab('%s_bias_cat_%d' % (pf, i), ia._embedding_classes_list_cat[i], 1)
            for i in range(ia.num_features_mulhot):
                tab('%sembed_mulhot_%d' % (pf, i), ia._embedding_classes_list_mulhot[i], d)
                tab('%s_bias_mulhot_%d' % (pf, i), ia._embedding_classes_list_mulhot[i], 1)
        return out
