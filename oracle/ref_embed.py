"""Embedding-space (sparse) fp64 restatement of ONE training step of the hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py ("parity unpinned" like the rest of oracle/).

oracle/ref_graph.py / ref_lstm.py keep the reference's own DENSE form (whole attribute table times
u^T, dense table gradient, dense Adagrad: embed_attribute.py:171,188, hmf_model.py:146-151) --
faithful, but O(table) per step, i.e. unusable at BASELINE.json's sizes (1 M x 128 tables,
B = 16384).  This file computes the SAME step over the rows the step touches only:

    pool / target / user rows are gathered first (mean over features, mean over bag tokens),
    logits = U . Pbar^T + bbar, and every lookup's gradient is scattered back onto its table
    rows (duplicates summed first, then ONE Adagrad update per touched row).

Equal to the dense form by linearity of the mean and because Adagrad leaves zero-gradient rows
alone (acc += 0, w -= 0).  tests/test_oracle_cpu.py proves the equality against
ref_graph.RefLatentProductModel and ref_lstm.RefSeqModel at small sizes (losses and every
parameter / slot to 1e-10); tests/test_fullsize_gpu.py then uses it as the full-size checker.

Tables stay in the caller's fp32 arrays (`base`, read only); rows this object has updated live
in an fp64 overlay, so three consecutive 1 M-row steps cost memory for the touched rows only.
"""
from __future__ import annotations

import numpy as np

from . import ref_lstm


class SparseTable(object):
    """fp32 base array [V, d] (or [V] for a bias) + fp64 overlay of updated rows and slots."""

    def __init__(self, base, acc0=0.1):
        self.base = base
        self.acc0 = acc0
        self.idx = np.zeros((0,), dtype=np.int64)            # sorted, unique
        shape = (0,) + tuple(base.shape[1:])
        self.val = np.zeros(shape, dtype=np.float64)
        self.acc = np.zeros(shape, dtype=np.float64)

    def rows(self, r):
        r = np.asarray(r, dtype=np.int64)
        out = self.base[r].astype(np.float64)
        if len(self.idx):
            p = np.searchsorted(self.idx, r)
            p[p >= len(self.idx)] = 0
            hit = self.idx[p] == r
            out[hit] = self.val[p[hit]]
        return out

    def slots(self, r):
        r = np.asarray(r, dtype=np.int64)
        out = np.full((len(r),) + tuple(self.base.shape[1:]), self.acc0, dtype=np.float64)
        if len(self.idx):
            p = np.searchsorted(self.idx, r)
            p[p >= len(self.idx)] = 0
            hit = self.idx[p] == r
            out[hit] = self.acc[p[hit]]
        return out

    def adagrad(self, r, g, lr):
        """r: unique rows, g: their summed gradients.  acc += g^2; w -= lr * g / sqrt(acc)."""
        w, a = self.rows(r), self.slots(r)
        a = a + g * g
        w = w - lr * g / np.sqrt(a)
        self.put(r, w, a)

    def sgd(self, r, g, lr):
        self.put(r, self.rows(r) - lr * g, self.slots(r))

    def put(self, r, w, a):
        allr = np.concatenate([self.idx, r])
        allv = np.concatenate([self.val, w], 0)
        alla = np.concatenate([self.acc, a], 0)
        order = np.argsort(allr, kind='stable')
        allr, allv, alla = allr[order], allv[order], alla[order]
        last = np.ones(len(allr), dtype=bool)
        last[:-1] = allr[1:] != allr[:-1]                    # the newest entry of a row wins
        self.idx, self.val, self.acc = allr[last], allv[last], alla[last]


def _merge(rows, vals):
    """Sum `vals` over equal `rows` -> (unique rows ascending, sums)."""
    rows = np.asarray(rows, dtype=np.int64)
    if len(rows) == 0:
        return rows, vals
    order = np.argsort(rows, kind='stable')
    rs, vs = rows[order], vals[order]
    head = np.ones(len(rs), dtype=bool)
    head[1:] = rs[1:] != rs[:-1]
    return rs[head], np.add.reduceat(vs, np.nonzero(head)[0], axis=0)


class _Attr(object):
    """Feature maps of one entity kind in the reference's layout."""

    def __init__(self, att, prefix, with_bias):
        self.cat = [np.asarray(x, dtype=np.int64) for x in (att.features_cat or [])][:att.num_features_cat]
        n = att.num_features_mulhot
        self.vals = [np.asarray(x, dtype=np.int64) for x in (att.features_mulhot or [])][:n]
        self.starts = [np.asarray(x, dtype=np.int64) for x in (att.mulhot_starts or [])][:n]
        self.lens = [np.asarray(x, dtype=np.int64) for x in (att.mulhot_lengths or [])][:n]
        self.emb = (['%sembed_cat_%d' % (prefix, i) for i in range(len(self.cat))] +
                    ['%sembed_mulhot_%d' % (prefix, i) for i in range(n)])
        self.bias = (['%s_bias_cat_%d' % (prefix, i) for i in range(len(self.cat))] +
                     ['%s_bias_mulhot_%d' % (prefix, i) for i in range(n)]) if with_bias else None

    def features(self, no_id=False, no_attribute=False):
        """[(feature index in emb/bias lists, kind, maps)] (embed_attribute.py:356-373)."""
        ncat = len(self.cat)
        cats = list(range(ncat))
        muls = list(range(len(self.vals)))
        if no_attribute:
            cats, muls = cats[:1], []
        if no_id:
            cats = cats[1:]
        return [(i, 'cat') for i in cats] + [(ncat + i, 'mulhot') for i in muls]

    def lookup(self, feat, kind, ids):
        """-> (table rows [n_contrib], segment of each contribution, 1/len per contribution,
        first contribution of every segment or None for one-hot features)."""
        ids = np.asarray(ids, dtype=np.int64)
        if kind == 'cat':
            return self.cat[feat][ids], np.arange(len(ids)), np.ones(len(ids)), None
        k = feat - len(self.cat)
        st, ln = self.starts[k][ids], self.lens[k][ids]          # every bag holds >= 1 token
        offs = np.concatenate([[0], np.cumsum(ln)[:-1]])
        seg = np.repeat(np.arange(len(ids)), ln)
        rows = self.vals[k][np.arange(int(ln.sum())) + np.repeat(st - offs, ln)]
        return rows, seg, np.repeat(1.0 / ln, ln), offs


class _Model(object):
    def __init__(self, u_attr, i_attr, params32, acc0=0.1):
        self.ua = _Attr(u_attr, 'user', False)
        self.ia = _Attr(i_attr, 'item', True)
        self.t = {k: SparseTable(np.asarray(v).reshape(-1) if np.asarray(v).ndim == 2 and np.asarray(v).shape[1] == 1
                                 and 'bias' in k else np.asarray(v), acc0) for k, v in params32.items()}
        self.n_items = (len(self.ia.cat[0]) if self.ia.cat else len(self.ia.lens[0])) - 1

    # ---- forward: mean over features of (row | mean of bag rows); bias alike ----
    def embed(self, attr, ids, with_bias, no_id=False, no_attribute=False):
        feats = attr.features(no_id=no_id, no_attribute=no_attribute)
        sites, e, b = [], 0.0, 0.0
        for fi, kind in feats:
            rows, seg, w, offs = attr.lookup(fi, kind, ids)
            v = self.t[attr.emb[fi]].rows(rows) * w[:, None]
            e = e + (v if offs is None else np.add.reduceat(v, offs, axis=0))     # segment mean
            if with_bias:
                bv = self.t[attr.bias[fi]].rows(rows) * w
                b = b + (bv if offs is None else np.add.reduceat(bv, offs))
            sites.append((fi, rows, seg, w))
        F = max(len(feats), 1)
        return e / F, (b / F if with_bias else None), (attr, sites, F)

    # ---- backward of embed: contributions (table name, rows, values) ----
    @staticmethod
    def embed_bwd(ctx, d_e, d_b, out):
        attr, sites, F = ctx
        for fi, rows, seg, w in sites:
            if d_e is not None:
                out.append((attr.emb[fi], rows, d_e[seg] * (w / F)[:, None]))
            if d_b is not None and attr.bias is not None:
                out.append((attr.bias[fi], rows, d_b[seg] * (w / F)))

    def mask(self, users, pool, pos_ptr, pos_items):
        """embed_attribute.py:729-741: mask[r, s] = False where pool[s] is a positive of user r."""
        B, S = len(users), len(pool)
        slot = np.full(self.n_items + 2, -1, dtype=np.int64)
        slot[np.asarray(pool, dtype=np.int64)] = np.arange(S)
        m = np.ones((B, S), dtype=bool)
        users = np.asarray(users, dtype=np.int64)
        cnt = (pos_ptr[users + 1] - pos_ptr[users]).astype(np.int64)
        r = np.repeat(np.arange(B), cnt)
        first = np.repeat(pos_ptr[users].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
        it = np.asarray(pos_items)[np.arange(int(cnt.sum())) + first]
        s = slot[it]
        ok = s >= 0
        m[r[ok], s[ok]] = False
        return m

    @staticmethod
    def sampled_loss(kind, logits, t, mask):
        """'mw' (embed_attribute.py:641-649) / build-defined 'mce' -> (batch_loss, dlogits/d_bl, dt/d_bl)
        as functions of the upstream d_bl applied later (returned un-scaled: per unit d_bl)."""
        if kind == 'mw':
            v = np.where(mask, logits - t[:, None] + 1.0, 0.0)
            act = v > 0
            s = np.where(act, v, 0.0).sum(1)
            bl = np.log1p(s)
            d = act / (1.0 + s)[:, None]
        else:
            logits = np.minimum(logits, t[:, None] + 64.0)       # ref_graph.MCE_SAT: the build-defined saturation
            mx = np.maximum(np.where(mask, logits, -np.inf).max(1), t)
            ex = np.where(mask, np.exp(logits - mx[:, None]), 0.0)
            z = np.exp(t - mx) + ex.sum(1)
            bl = mx - t + np.log(z)
            d = ex / z[:, None]
        return bl, d, -d.sum(1)


class EmbedSpaceHMF(_Model):
    """hmf_model.py:67-151 with a sampled loss ('mw' | 'mce'), one Adagrad step."""

    def __init__(self, u_attr, i_attr, params32, lr, loss='mw', acc0=0.1):
        super().__init__(u_attr, i_attr, params32, acc0)
        self.lr, self.loss = float(lr), loss
        self.pool = None

    def step(self, users, items, pool, pos_ptr, pos_items):
        if pool is not None:
            self.pool = np.asarray(pool, dtype=np.int64)
        B = len(users)
        U, _, cu = self.embed(self.ua, users, False)
        P, pb, cp = self.embed(self.ia, self.pool, True)
        T, tb, ct = self.embed(self.ia, items, True)
        logits = U @ P.T + pb
        t = (U * T).sum(1) + tb
        m = self.mask(users, self.pool, pos_ptr, pos_items)
        bl, d, dt = self.sampled_loss(self.loss, logits, t, m)
        d, dt = d / B, dt / B                                    # d(mean)/d(batch_loss) = 1/B
        dU = d @ P + dt[:, None] * T
        contrib = []
        self.embed_bwd(cp, d.T @ U, d.sum(0), contrib)
        self.embed_bwd(ct, dt[:, None] * U, dt, contrib)
        self.embed_bwd(cu, dU, None, contrib)
        self.last = {'logits': logits, 'batch_loss': bl, 'mask': m}
        self.apply(contrib, 1.0)
        return float(bl.mean())

    def apply(self, contrib, scale):
        by = {}
        for name, rows, vals in contrib:
            by.setdefault(name, []).append((rows, vals))
        self.touched = {}
        for name, lst in by.items():
            r, g = _merge(np.concatenate([a for a, _ in lst]), np.concatenate([b for _, b in lst], 0))
            self.t[name].adagrad(r, g * scale, self.lr)
            self.touched[name] = r


class EmbedSpaceSeq(_Model):
    """lstm/seqModel.py:87-184,454-604 (use_concat=False, one layer, keep_prob = 1) with a sampled
    loss, clip_by_global_norm under TF-1.0's aggregation rule (oracle/ref_graph.py
    Grads.sq_norm_unmerged) and Adagrad -- the sparse twin of ref_lstm.RefSeqModel."""

    def __init__(self, u_attr, i_attr, params32, W, b, lr, max_gradient_norm, loss='mw',
                 no_user_id=True, acc0=0.1):
        super().__init__(u_attr, i_attr, params32, acc0)
        self.W, self.b = np.array(W, dtype=np.float64), np.array(b, dtype=np.float64)
        self.W_acc, self.b_acc = np.full(self.W.shape, acc0), np.full(self.b.shape, acc0)
        self.lr, self.clip, self.loss, self.no_user_id = float(lr), float(max_gradient_norm), loss, no_user_id
        self.pool = None

    def step(self, users, inputs, targets, weights, pool, pos_ptr, pos_items):
        """inputs / targets / weights: [L, B] arrays (time-major)."""
        if pool is not None:
            self.pool = np.asarray(pool, dtype=np.int64)
        inputs, targets = np.asarray(inputs, dtype=np.int64), np.asarray(targets, dtype=np.int64)
        w = np.asarray(weights, dtype=np.float64)
        L, B = inputs.shape
        zero_user = self.no_user_id and len(self.ua.cat) == 1        # embed_attribute.py:356-366
        if zero_user:
            u, cu = 0.0, None
        else:
            u, _, cu = self.embed(self.ua, users, False, no_id=self.no_user_id)
        it, _, ci = self.embed(self.ia, inputs.reshape(-1), False)
        x = 0.5 * (it.reshape(L, B, -1) + u)                          # :148-156
        hs, cs, gates = ref_lstm.lstm_fwd(x, self.W, self.b, 1.0)     # :477
        P, pb, cp = self.embed(self.ia, self.pool, True)
        T, tb, ct = self.embed(self.ia, targets.reshape(-1), True)
        H = hs.reshape(L * B, -1)
        logits = H @ P.T + pb                                          # :492
        t = (H * T).sum(1) + tb                                        # :493
        m = self.mask(users, self.pool, pos_ptr, pos_items)
        bl, d, dt = self.sampled_loss(self.loss, logits, t, np.tile(m, (L, 1)))
        tot = w.sum(0) + 1e-12                                         # :551-567
        cost = float(((bl.reshape(L, B) * w).sum(0) / tot).sum())      # :596
        g = (w / tot).reshape(-1)                                      # d cost / d batch_loss
        d, dt = d * g[:, None], dt * g
        dH = d @ P + dt[:, None] * T
        # ---- the gradient list tf.gradients hands to clip_by_global_norm (:179-180) ----
        dense, sparse = {}, {}          # name -> [per unrolled step: (rows, merged vals)] / [vals]
        total = []
        S = len(self.pool)
        for step in range(L):
            sl = slice(step * B, (step + 1) * B)
            c = []
            self.embed_bwd(cp, d[sl].T @ H[sl], d[sl].sum(0), c)       # matmul'd table: dense per step
            for name, rows, vals in c:
                r, v = _merge(rows, vals)        # rows shared between pool items are summed (dense matrix)
                dense.setdefault(name, []).append(v)
                total.append((name, r, v))
        c = []
        self.embed_bwd(ct, dt[:, None] * H, dt, c)                     # target lookups: IndexedSlices
        dz, dx, dW, db = ref_lstm.lstm_bwd(x, self.W, hs, cs, gates, dH.reshape(L, B, -1))
        self.embed_bwd(ci, 0.5 * dx.reshape(L * B, -1), None, c)       # input lookups: IndexedSlices
        if not zero_user:
            self.embed_bwd(cu, 0.5 * dx.sum(0), None, c)
        for name, rows, vals in c:
            sparse.setdefault(name, []).append(vals)
            total.append((name, rows, vals))
        sq = float((dW * dW).sum() + (db * db).sum())
        for name in set(dense) | set(sparse):
            if name in sparse:           # any IndexedSlices => everything concatenated, un-merged
                sq += sum(float((v * v).sum()) for v in dense.get(name, []))
                sq += sum(float((v * v).sum()) for v in sparse[name])
            else:                        # all dense => add_n first
                lst = [(n_, r, v) for n_, r, v in total if n_ == name]
                _, v = _merge(np.concatenate([r for _, r, _ in lst]), np.concatenate([v for _, _, v in lst], 0))
                sq += float((v * v).sum())
        gnorm = np.sqrt(sq)
        scale = self.clip / max(gnorm, self.clip)
        self.last = {'gnorm': gnorm, 'batch_loss': bl, 'hs': hs}
        by = {}
        for name, rows, vals in total:
            by.setdefault(name, []).append((rows, vals))
        self.touched = {}
        for name, lst in by.items():
            r, gsum = _merge(np.concatenate([a for a, _ in lst]), np.concatenate([b for _, b in lst], 0))
            self.t[name].adagrad(r, gsum * scale, self.lr)
            self.touched[name] = r
        for p, a, gr in ((self.W, self.W_acc, dW * scale), (self.b, self.b_acc, db * scale)):
            a += gr * gr
            p -= self.lr * gr / np.sqrt(a)
        return cost
