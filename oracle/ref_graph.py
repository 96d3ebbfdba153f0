"""numpy restatement of the reference's TF-1.0 op graph for the hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  "parity unpinned": the
reference has no golden vectors for this arithmetic and TensorFlow cannot run
here, so this file follows the reference's *call sites* (cited per function as
reference file:line, relative to the reference repo root) plus documented
TF-1.0 op semantics.  It deliberately keeps the
reference's *algorithm*: the scorer multiplies the whole attribute table by
u^T (embed_attribute.py:171,188) and back-propagates a dense table gradient
followed by a dense Adagrad update (hmf_model.py:146-151) -- it is NOT the
sparse embedding-space form the HIP path uses.  That makes it both the parity
checker and the "reference algorithm on CPU" baseline for bench.py.

dtype is a parameter: float64 = the oracle proper, float32 = what TF computes.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is only used to make the segment sums fast for the CPU baseline
    import scipy.sparse as _sp
except Exception:  # pragma: no cover
    _sp = None


# --------------------------------------------------------------------------
# TF op restatements
# --------------------------------------------------------------------------
def batch_slice2(target, b, s):
    """attributes/mulhot_index.py:48-52 -- concat_r target[b_r : b_r + s_r]."""
    target = np.asarray(target)
    if len(b) == 0:
        return np.zeros((0,), dtype=np.int32)
    return np.concatenate([target[int(b[i]):int(b[i]) + int(s[i])]
                           for i in range(len(b))]).astype(np.int32)


def batch_segids2(s):
    """attributes/mulhot_index.py:62-67 -- concat_r tile([r], [s_r])."""
    if len(s) == 0:
        return np.zeros((0,), dtype=np.int32)
    return np.concatenate([np.full((int(s[i]),), i, dtype=np.int32)
                           for i in range(len(s))])


def unsorted_segment_sum(data, segids, n):
    """tf.unsorted_segment_sum (embed_attribute.py:192,398,404)."""
    data = np.asarray(data)
    segids = np.asarray(segids)
    out_shape = (n,) + data.shape[1:]
    if data.shape[0] == 0:
        return np.zeros(out_shape, dtype=data.dtype)
    if _sp is not None and data.ndim == 2 and data.shape[0] > 4096:
        T = data.shape[0]
        m = _sp.csr_matrix((np.ones(T, dtype=data.dtype),
                            (segids, np.arange(T))), shape=(n, T))
        return np.asarray(m @ data, dtype=data.dtype)
    out = np.zeros(out_shape, dtype=data.dtype)
    np.add.at(out, segids, data)
    return out


MCE_SAT = 64.0      # saturation of the build-defined 'mce' exponent (csrc/common.h kMceSat)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# gradient container: what tf.gradients hands to the optimiser
# --------------------------------------------------------------------------
class Grads(object):
    """Per-variable list of dense tensors and IndexedSlices (indices, values),
    un-merged, exactly as tf.gradients collects them before aggregation."""

    def __init__(self):
        self.dense = {}
        self.sparse = {}

    def add_dense(self, name, g):
        self.dense.setdefault(name, []).append(g)

    def add_sparse(self, name, indices, values):
        self.sparse.setdefault(name, []).append(
            (np.asarray(indices, dtype=np.int64), values))

    def names(self):
        return sorted(set(self.dense) | set(self.sparse))

    def total(self, name, shape, dtype):
        """Dense sum of every contribution (duplicates summed)."""
        g = np.zeros(shape, dtype=dtype)
        for d in self.dense.get(name, []):
            g += d.reshape(shape)
        for idx, val in self.sparse.get(name, []):
            np.add.at(g, idx, val.reshape((len(idx),) + tuple(shape[1:])))
        return g

    def sq_norm_unmerged(self, name):
        """Sum of squares the way tf.clip_by_global_norm sees the aggregated
        gradient (seqModel.py:180): a variable with any IndexedSlices
        contribution is aggregated by *concatenating* all contributions as
        IndexedSlices (TF-1.0 gradients_impl._AggregatedGrads), and
        global_norm uses `.values` without merging duplicates; an all-dense
        list is add_n'ed first."""
        dn = self.dense.get(name, [])
        sp = self.sparse.get(name, [])
        if not sp:
            g = sum(dn[1:], dn[0].copy())
            return float(np.sum(np.square(g, dtype=np.float64)))
        tot = 0.0
        for d in dn:
            tot += float(np.sum(np.square(d, dtype=np.float64)))
        for _, v in sp:
            tot += float(np.sum(np.square(v, dtype=np.float64)))
        return tot


def adagrad_apply(param, acc, g, lr):
    """tf.train.AdagradOptimizer (hmf_model.py:147, seqModel.py:174):
    acc += g^2 ; var -= lr * g / sqrt(acc); accumulators start at 0.1.
    Duplicate sparse indices are summed before the single application
    (TF>=1.0 _apply_sparse_duplicate_indices); rows with g == 0 are unchanged,
    so applying the dense total is identical to the sparse application."""
    acc += g * g
    param -= lr * g / np.sqrt(acc)


# --------------------------------------------------------------------------
# EmbeddingAttribute restated
# --------------------------------------------------------------------------
class RefEmbeddingAttribute(object):
    """Restates attributes/embed_attribute.py:19-747 on numpy arrays.

    `params` maps the reference's variable names (embed_attribute.py:275-304:
    'userembed_cat_0', 'itemembed_mulhot_0', 'item_bias_cat_0', ... and the
    'item_output' twins) to arrays; `slots` holds the Adagrad accumulators.
    """

    def __init__(self, user_attributes, item_attributes, mb, n_sampled,
                 input_steps=0, item_output=False, item_ind2logit_ind=None,
                 logit_ind2item_ind=None, params=None, dtype=np.float32,
                 acc0=0.1):
        self.user_attributes = user_attributes
        self.item_attributes = item_attributes
        self.batch_size = mb
        self.n_sampled = n_sampled
        self.input_steps = input_steps
        self.item_output = item_output
        self.item_ind2logit_ind = item_ind2logit_ind
        self.logit_ind2item_ind = logit_ind2item_ind
        self.logit_size = len(logit_ind2item_ind) if logit_ind2item_ind is not None else None
        self.dt = np.dtype(dtype)
        self.params = {k: np.array(v, dtype=self.dt) for k, v in (params or {}).items()}
        self.slots = {k: np.full(v.shape, acc0, dtype=self.dt) for k, v in self.params.items()}
        # embed_attribute.py:308-318 -- attribute maps as constants
        self.att = {
            'user': self._maps(user_attributes),
            'item': self._maps(item_attributes),
        }
        self.att['item_output'] = self.att['item']
        self.pos_item_set = None
        self.pos_item_set_eval = None
        self.sampled = None

    @staticmethod
    def _maps(att):
        return ([np.asarray(x, dtype=np.int32) for x in (att.features_cat or [])][:att.num_features_cat],
                [np.asarray(x, dtype=np.int32) for x in (att.features_mulhot or [])][:att.num_features_mulhot],
                [np.asarray(x, dtype=np.int32) for x in (att.mulhot_starts or [])][:att.num_features_mulhot],
                [np.asarray(x, dtype=np.int32) for x in (att.mulhot_lengths or [])][:att.num_features_mulhot])

    # ---- variable name helpers (embed_attribute.py:265-306) ----
    def _emb_names(self, prefix, att):
        return (['%sembed_cat_%d' % (prefix, i) for i in range(att.num_features_cat)],
                ['%sembed_mulhot_%d' % (prefix, i) for i in range(att.num_features_mulhot)])

    def _bias_names(self, prefix, att):
        return (['%s_bias_cat_%d' % (prefix, i) for i in range(att.num_features_cat)],
                ['%s_bias_mulhot_%d' % (prefix, i) for i in range(att.num_features_mulhot)])

    def _out_prefix(self):
        return 'item_output' if self.item_output else 'item'

    # ---- a5: _get_embedded (embed_attribute.py:350-417) ----
    def get_embedded(self, prefix, inds, with_bias, no_id=False, no_attribute=False,
                     table_prefix=None):
        """Returns (cat_list, mulhot_list, bias, cache)."""
        attributes = self.user_attributes if prefix == 'user' else self.item_attributes
        tp = table_prefix or prefix
        ecat, emul = self._emb_names(tp, attributes)
        bcat, bmul = self._bias_names(tp, attributes) if with_bias else (None, None)
        inds = np.asarray(inds, dtype=np.int64)
        mb = len(inds)
        maps = self.att[prefix]
        cat_list, mulhot_list, bias_cat_list, bias_mulhot_list = [], [], [], []
        cache = {'sites': [], 'mb': mb}
        if no_id and attributes.num_features_cat == 1:   # :356-366
            d = attributes._embedding_size_list_cat[0]
            return [np.zeros((mb, d), dtype=self.dt)], [], None, cache
        n1 = 1 if no_attribute else attributes.num_features_cat
        n2 = 0 if no_attribute else attributes.num_features_mulhot
        for i in range(n1):
            if no_id and i == 0:
                continue
            cat_indices = maps[0][i][inds]                     # :374
            cat_list.append(self.params[ecat[i]][cat_indices])  # :375
            site = {'kind': 'cat', 'emb': ecat[i], 'idx': cat_indices, 'bias': None}
            if with_bias:
                bias_cat_list.append(self.params[bcat[i]][cat_indices])  # :379 [mb,1]
                site['bias'] = bcat[i]
            cache['sites'].append(site)
        for i in range(n2):
            begin_ = maps[2][i][inds]                           # :383
            size_ = maps[3][i][inds]                            # :384
            mulhot_indices = batch_slice2(maps[1][i], begin_, size_)   # :394
            mulhot_segids = batch_segids2(size_)                # :396
            embedded_flat = self.params[emul[i]][mulhot_indices]  # :397
            embedded_sum = unsorted_segment_sum(embedded_flat, mulhot_segids, mb)  # :398
            lengs = size_.astype(self.dt).reshape(mb, 1)        # :399
            mulhot_list.append(embedded_sum / lengs)            # :400
            site = {'kind': 'mulhot', 'emb': emul[i], 'idx': mulhot_indices,
                    'seg': mulhot_segids, 'lengs': lengs, 'bias': None}
            if with_bias:
                bflat = self.params[bmul[i]][mulhot_indices]    # :403
                bsum = unsorted_segment_sum(bflat, mulhot_segids, mb)
                bias_mulhot_list.append(bsum / lengs)           # :406
                site['bias'] = bmul[i]
            cache['sites'].append(site)
        if not with_bias:
            bias = None
        else:
            bl = bias_cat_list + bias_mulhot_list
            bias = np.squeeze(np.mean(np.stack(bl, 0), 0), axis=-1)   # :412  -> [mb]
        return cat_list, mulhot_list, bias, cache

    def get_embedded_bwd(self, cache, d_feats, d_bias, grads):
        """d_feats: list (one [mb,d] per produced feature, same order as
        cat_list + mulhot_list); d_bias: [mb] or None."""
        nfeat = len(cache['sites'])
        for k, site in enumerate(cache['sites']):
            g = d_feats[k]
            if site['kind'] == 'cat':
                if g is not None:
                    grads.add_sparse(site['emb'], site['idx'], g)   # gather grad -> IndexedSlices
                if site['bias'] is not None and d_bias is not None:
                    grads.add_sparse(site['bias'], site['idx'], (d_bias / nfeat).reshape(-1, 1))
            else:
                if g is not None:
                    gflat = (g / site['lengs'])[site['seg']]
                    grads.add_sparse(site['emb'], site['idx'], gflat)
                if site['bias'] is not None and d_bias is not None:
                    gb = ((d_bias / nfeat).reshape(-1, 1) / site['lengs'])[site['seg']]
                    grads.add_sparse(site['bias'], site['idx'], gb)

    # ---- a6: get_batch_user / get_batch_item (embed_attribute.py:222-254) ----
    def get_batch_user(self, user_input, concat=False, no_id=False):
        cat, mul, _, cache = self.get_embedded('user', user_input, False, no_id=no_id)
        feats = cat + mul
        if concat:
            out = np.concatenate(feats, axis=1)                # :415
        else:
            out = np.mean(np.stack(feats, 0), 0)               # :235
        cache['concat'] = concat
        cache['dims'] = [f.shape[1] for f in feats]
        return out, cache

    def get_batch_user_bwd(self, cache, d_out, grads):
        n = len(cache['dims'])
        if cache['concat']:
            offs = np.cumsum([0] + cache['dims'])
            d_feats = [d_out[:, offs[k]:offs[k + 1]] for k in range(n)]
        else:
            d_feats = [d_out / n for _ in range(n)]
        if len(cache['sites']) == 0:
            return
        self.get_embedded_bwd(cache, d_feats, None, grads)

    def get_batch_item(self, item_input, concat=False, no_attribute=False):
        cat, mul, bias, cache = self.get_embedded('item', item_input, True,
                                                  no_attribute=no_attribute)
        feats = cat + mul
        cache['concat'] = concat
        cache['dims'] = [f.shape[1] for f in feats]
        if concat:
            return np.concatenate(feats, axis=1), bias, cache
        return feats, bias, cache

    # ---- a7: sampled-pool staging (embed_attribute.py:320-348) ----
    def update_sampled(self, item_sampled):
        ia = self.item_attributes
        maps = self.att[self._out_prefix()]
        inds = np.asarray(item_sampled, dtype=np.int64)
        assert len(inds) == self.n_sampled
        cat = [maps[0][i][inds] for i in range(ia.num_features_cat)]         # :328
        mul_idx, mul_seg, mul_len, mul_l = [], [], [], []
        for i in range(ia.num_features_mulhot):
            begin_ = maps[2][i][inds]
            size_ = maps[3][i][inds]
            mul_idx.append(batch_slice2(maps[1][i], begin_, size_))          # :335
            mul_seg.append(batch_segids2(size_))                             # :336
            mul_len.append(size_.astype(self.dt).reshape(self.n_sampled, 1))  # :342
            mul_l.append(int(size_.sum()))                                   # :338,347
        self.sampled = (cat, mul_idx, mul_seg, mul_len, mul_l)
        return self.sampled

    def _full_pool(self):
        ia = self.item_attributes
        cat = [np.asarray(ia.full_cat_tr[i], dtype=np.int64) for i in range(ia.num_features_cat)]
        mul_idx = [np.asarray(ia.full_values_tr[i], dtype=np.int64) for i in range(ia.num_features_mulhot)]
        mul_seg = [np.asarray(ia.full_segids_tr[i], dtype=np.int64) for i in range(ia.num_features_mulhot)]
        mul_len = [np.asarray(ia.full_lengths_tr[i], dtype=self.dt).reshape(-1, 1)
                   for i in range(ia.num_features_mulhot)]
        return cat, mul_idx, mul_seg, mul_len

    # ---- a8: get_prediction (embed_attribute.py:148-206) ----
    def get_prediction(self, latent, pool='full', output_feat=1):
        """latent: [mb,d] array or list (one per feature).  Returns logits
        [mb, V|S] and a cache for get_prediction_bwd."""
        ia = self.item_attributes
        op = self._out_prefix()
        ecat, emul = self._emb_names(op, ia)
        bcat, bmul = self._bias_names(op, ia)
        if pool == 'full':
            indices_cat, indices_mulhot, segids_mulhot, lengths_mulhot = self._full_pool()
            V = self.logit_size
        else:
            indices_cat, indices_mulhot, segids_mulhot, lengths_mulhot, _ = self.sampled
            V = self.n_sampled
        n1 = 1 if output_feat == 0 else ia.num_features_cat      # :163
        n2 = 0 if output_feat == 0 else ia.num_features_mulhot   # :164
        innerps, sites = [], []
        for i in range(n1):
            u = latent[i] if isinstance(latent, list) else latent   # :169
            E, b = self.params[ecat[i]], self.params[bcat[i]]
            innerp = E @ u.T + b                                    # :171  [Vf, mb]
            inds = indices_cat[i]
            innerps.append(innerp[inds])                            # :172  [V, mb]
            sites.append({'kind': 'cat', 'emb': ecat[i], 'bias': bcat[i], 'u': u,
                          'inds': inds, 'li': i})
        offset = ia.num_features_cat
        for i in range(n2):
            u = latent[i + offset] if isinstance(latent, list) else latent   # :178
            E, b = self.params[emul[i]], self.params[bmul[i]]
            lengs = lengths_mulhot[i]
            inds, segids = indices_mulhot[i], segids_mulhot[i]      # :181-186
            innerp = E @ u.T + b                                    # :188-189
            site = {'kind': 'mulhot', 'emb': emul[i], 'bias': bmul[i], 'u': u,
                    'inds': inds, 'seg': segids, 'lengs': lengs, 'li': i + offset,
                    'of': output_feat}
            if output_feat == 1:
                innerps.append(unsorted_segment_sum(innerp[inds], segids, V) / lengs)   # :192-193
            elif output_feat == 2:
                looked = innerp[inds]
                out = np.full((V, looked.shape[1]), -np.inf, dtype=self.dt)
                np.maximum.at(out, segids, looked)                  # :195 segment_max
                innerps.append(out)
                site['looked'] = looked
                site['out'] = out
            elif output_feat == 3:
                score_max = innerp.max()                            # :197
                ex = np.exp(innerp[inds] - score_max)               # :198-200
                ssum = unsorted_segment_sum(ex, segids, V)
                innerps.append(score_max + np.log(1 + ssum))
                site['ex'] = ex
                site['ssum'] = ssum
            else:
                raise NotImplementedError('Attribute combination not implemented!')   # :202
            sites.append(site)
        logits = np.mean(np.stack(innerps, 0), 0).T                 # :205
        cache = {'sites': sites, 'V': V, 'latent_is_list': isinstance(latent, list),
                 'n_latent': len(latent) if isinstance(latent, list) else 1}
        return np.ascontiguousarray(logits), cache

    def get_prediction_bwd(self, cache, d_logits, grads):
        """Back-prop through the *reference form*: dense d innerp [Vf,mb], dense
        dE = d innerp . u, db = rowsum (matmul/add gradients)."""
        sites = cache['sites']
        nF = len(sites)
        V = cache['V']
        d_innerps = d_logits.T / nF                               # [V, mb]
        if cache['latent_is_list']:
            d_latent = [None] * cache['n_latent']
        else:
            d_latent = None
        for site in sites:
            E = self.params[site['emb']]
            u = site['u']
            d_innerp = np.zeros((E.shape[0], u.shape[0]), dtype=self.dt)
            if site['kind'] == 'cat':
                np.add.at(d_innerp, site['inds'], d_innerps)
            else:
                of = site['of']
                if of == 1:
                    dflat = (d_innerps / site['lengs'])[site['seg']]
                elif of == 2:
                    # segment_max gradient: to the arg-max entries (ties share equally in TF)
                    is_max = (site['looked'] == site['out'][site['seg']]).astype(self.dt)
                    cnt = unsorted_segment_sum(is_max, site['seg'], V)
                    dflat = is_max * (d_innerps / cnt)[site['seg']]
                else:
                    # of == 3; note: gradient through reduce_max(innerp) cancels exactly
                    dflat = site['ex'] * (d_innerps / (1 + site['ssum']))[site['seg']]
                    # d/d score_max of [score_max + log(1+sum exp(x-score_max))] = 1 - s/(1+s)
                    # flows to the arg-max element(s) of innerp
                    resid = (d_innerps * (1.0 - site['ssum'] / (1 + site['ssum']))).sum()
                    innerp = E @ u.T + self.params[site['bias']]
                    am = (innerp == innerp.max()).astype(self.dt)
                    d_innerp += am * (resid / am.sum())
                np.add.at(d_innerp, site['inds'], dflat)
            grads.add_dense(site['emb'], d_innerp @ u)             # dE [Vf,d]
            grads.add_dense(site['bias'], d_innerp.sum(1, keepdims=True))
            du = d_innerp.T @ E                                    # [mb,d]
            if cache['latent_is_list']:
                li = site['li']
                d_latent[li] = du if d_latent[li] is None else d_latent[li] + du
            else:
                d_latent = du if d_latent is None else d_latent + du
        return d_latent

    # ---- a9: get_target_score (embed_attribute.py:208-220) ----
    def get_target_score(self, latent, inds):
        cat, mul, i_bias, cache = self.get_embedded('item', inds, True,
                                                    table_prefix=self._out_prefix())
        feats = cat + mul
        target_item_emb = np.mean(np.stack(feats, 0), 0)          # :219
        score = np.sum(latent * target_item_emb, 1) + i_bias      # :220
        cache['latent'] = latent
        cache['temb'] = target_item_emb
        cache['nfeat'] = len(feats)
        return score, cache

    def get_target_score_bwd(self, cache, d_score, grads):
        d_latent = d_score[:, None] * cache['temb']
        d_temb = d_score[:, None] * cache['latent']
        n = cache['nfeat']
        self.get_embedded_bwd(cache, [d_temb / n] * n, d_score, grads)
        return d_latent

    # ---- a14: positive mask (embed_attribute.py:651-672, 721-745) ----
    def prepare_warp(self, pos_item_set, pos_item_set_eval):
        self.pos_item_set = pos_item_set
        self.pos_item_set_eval = pos_item_set_eval

    def mask_indices(self, user_input, loss, item_sampled_id2idx=None, forward_only=False):
        sampled = loss in ('mw', 'mce')                               # :717
        V = self.n_sampled if sampled else self.logit_size            # :724
        s_2idx = item_sampled_id2idx if sampled else self.item_ind2logit_ind   # :726
        item_set = self.pos_item_set_eval if forward_only else self.pos_item_set   # :727
        mask_indices, c = [], 0
        for u in user_input:
            offset = c * V
            if u in item_set:
                if sampled:
                    mask_indices.extend([s_2idx[v] + offset for v in item_set[u] if v in s_2idx])  # :739
                else:
                    mask_indices.extend([s_2idx[v] + offset for v in item_set[u]])  # :733
            c += 1
        return mask_indices, V

    def mask(self, user_input, loss, item_sampled_id2idx=None, forward_only=False):
        idx, V = self.mask_indices(user_input, loss, item_sampled_id2idx, forward_only)
        m = np.ones((len(user_input) * V,), dtype=bool)           # :655
        m[np.asarray(idx, dtype=np.int64)] = False                # :668 scatter_update
        return m.reshape(len(user_input), V)

    def target_mapping(self, item_target):
        """embed_attribute.py:679-684."""
        m = self.item_ind2logit_ind
        return [[m[v] for v in items] for items in item_target]

    # ---- a10-a13: losses (embed_attribute.py:525-649) ----
    def compute_loss(self, logits, item_target, loss='ce', mask=None, loss_func='log',
                     exp_p=1.005):
        """Returns (batch_loss [mb], cache)."""
        mb = logits.shape[0]
        rows = np.arange(mb)
        if loss == 'ce':                                           # :529-531
            mx = logits.max(1, keepdims=True)
            ex = np.exp(logits - mx)
            se = ex.sum(1, keepdims=True)
            lse = (np.log(se) + mx)[:, 0]
            tgt = np.asarray(item_target, dtype=np.int64)
            return lse - logits[rows, tgt], {'loss': 'ce', 'p': ex / se, 'tgt': tgt}
        if loss in ('warp', 'mw'):
            if loss == 'warp':                                     # :605-618
                tgt = np.asarray(item_target, dtype=np.int64)
                tl = logits[rows, tgt].reshape(mb, 1)
            else:                                                  # :641-649
                tgt = None
                tl = np.asarray(item_target, dtype=self.dt).reshape(mb, 1)
            logits2 = logits - tl + 1
            target = np.where(mask, logits2, 0)
            r = np.maximum(target, 0)
            s = r.sum(1)
            return np.log(1 + s), {'loss': loss, 'act': (target > 0), 's': s, 'tgt': tgt}
        if loss == 'mce':
            # BUILD-DEFINED (no reference arithmetic: 'mce' passes the assert at :527 and the feed
            # guard at :717 but has no branch below :529-549).  Sampled softmax in the shape of 'mw'
            # (:641-649): softmax cross-entropy over [target score || the sampled logits the mask
            # keeps], i.e.  log(1 + sum_s m_rs * exp(x_rs - t_r)); no log-Q correction, like 'mw'
            # has no |Y|/|Z| rescale.  SURVEY.md section 8(a) footnote.
            # Round 6: the exponent saturates at MCE_SAT = 64 (e_rs = exp(min(x_rs - t_r, 64))): identical to the
            # plain sampled softmax while no kept logit leads the target score by more than 64 (softmax weight
            # 1 - 1e-28 there), finite beyond -- one definition for every device path (csrc/common.h kMceSat).
            tl = np.asarray(item_target, dtype=self.dt).reshape(mb, 1)
            logits = np.minimum(logits, tl + MCE_SAT)
            mx = np.maximum(np.where(mask, logits, -np.inf).max(1, keepdims=True), tl)
            ex = np.where(mask, np.exp(logits - mx), 0)
            z = np.exp(tl - mx)[:, 0] + ex.sum(1)
            return (mx[:, 0] - tl[:, 0] + np.log(z)).astype(self.dt), {'loss': 'mce', 'p': ex / z.reshape(-1, 1)}
        if loss in ('rs', 'rs-sig', 'rs-sig2', 'bbpr'):           # :551-603
            tgt = np.asarray(item_target, dtype=np.int64)
            tl = logits[rows, tgt].reshape(mb, 1)
            if loss in ('rs', 'rs-sig'):
                pre = logits - tl + 1
                errors = np.maximum(pre, 0)
                derr = (pre > 0).astype(self.dt)
            else:
                errors = _sigmoid(logits - tl)
                derr = errors * (1 - errors)
            em = np.where(mask, errors, 0)
            dem = np.where(mask, derr, 0)
            if loss == 'rs-sig':
                sg = _sigmoid(em)
                dem = dem * 2 * sg * (1 - sg)
                em = sg * 2 - 1
            s = em.sum(1)
            if loss == 'bbpr':
                l, dl = s, np.ones_like(s)
            elif loss_func == 'log':
                l, dl = np.log(1 + s), 1 / (1 + s)
            elif loss_func == 'exp':
                l, dl = 1 - np.power(exp_p, -s), np.log(exp_p) * np.power(exp_p, -s)
            elif loss_func == 'poly':
                l, dl = np.power(s, exp_p), exp_p * np.power(s, exp_p - 1)
            elif loss_func == 'poly2':
                l, dl = np.power(1 + s, exp_p), exp_p * np.power(1 + s, exp_p - 1)
            elif loss_func == 'linear':
                l, dl = s, np.ones_like(s)
            elif loss_func == 'square':
                l, dl = np.square(s), 2 * s
            else:
                raise NotImplementedError(loss_func)
            return l.astype(self.dt), {'loss': loss, 'dem': dem, 'dl': dl, 'tgt': tgt}
        raise NotImplementedError('Error: not implemented other loss!!')   # :548

    def compute_loss_bwd(self, cache, d_batch_loss):
        """Returns (d_logits, d_target_score or None)."""
        loss = cache['loss']
        g = np.asarray(d_batch_loss, dtype=self.dt).reshape(-1, 1)
        if loss == 'ce':
            d = cache['p'] * g
            d[np.arange(d.shape[0]), cache['tgt']] -= g[:, 0]
            return d.astype(self.dt), None
        if loss == 'mce':
            d = (cache['p'] * g).astype(self.dt)
            return d, (-d.sum(1)).astype(self.dt)
        if loss in ('warp', 'mw'):
            d = cache['act'].astype(self.dt) * (g / (1 + cache['s']).reshape(-1, 1))
            dt = -d.sum(1)
            if loss == 'mw':
                return d.astype(self.dt), dt.astype(self.dt)
            d[np.arange(d.shape[0]), cache['tgt']] += dt
            return d.astype(self.dt), None
        d = cache['dem'] * (g * cache['dl'].reshape(-1, 1))
        dt = -d.sum(1)
        d[np.arange(d.shape[0]), cache['tgt']] += dt
        return d.astype(self.dt), None

    def warp_eval(self, logits, item_target, mask):
        """embed_attribute.py:620-639 -> [margin_rank, true_rank]."""
        mb = logits.shape[0]
        tl = logits[np.arange(mb), np.asarray(item_target, dtype=np.int64)].reshape(mb, 1)
        margin_rank = np.maximum(np.where(mask, logits - tl + 1, 0), 0).sum(1)
        true_rank = np.count_nonzero(np.where(mask, np.maximum(logits - tl, 0), 0), axis=1)
        return margin_rank, true_rank

    # ---- optimiser over a Grads container ----
    def apply_gradients(self, grads, lr, scale=1.0):
        """opt.apply_gradients (hmf_model.py:150): a variable with a dense (matmul)
        contribution gets the dense update over the whole table; a variable with
        only IndexedSlices gets TF's sparse apply: duplicates summed per unique
        row (_apply_sparse_duplicate_indices), then one update per touched row."""
        lr = self.dt.type(lr)
        for name in grads.names():
            p = self.params[name]
            if name in grads.dense:
                g = grads.total(name, p.shape, self.dt)
                if scale != 1.0:
                    g = g * self.dt.type(scale)
                adagrad_apply(p, self.slots[name], g, lr)
            else:
                idx = np.concatenate([i for i, _ in grads.sparse[name]])
                val = np.concatenate([v.reshape((len(i),) + p.shape[1:])
                                      for i, v in grads.sparse[name]], 0)
                uniq, inv = np.unique(idx, return_inverse=True)
                summed = np.zeros((len(uniq),) + p.shape[1:], dtype=self.dt)
                np.add.at(summed, inv, val)
                if scale != 1.0:
                    summed = summed * self.dt.type(scale)
                acc = self.slots[name][uniq] + summed * summed
                self.slots[name][uniq] = acc
                p[uniq] -= lr * summed / np.sqrt(acc)


# --------------------------------------------------------------------------
# HMF model restated (hmf/hmf_model.py)
# --------------------------------------------------------------------------
class RefLatentProductModel(object):
    """hmf/hmf_model.py:19-228 restated: graph assembly + step()."""

    def __init__(self, size, batch_size, learning_rate, user_attributes, item_attributes,
                 item_ind2logit_ind, logit_ind2item_ind, loss_function='ce',
                 n_sampled=None, params=None, dtype=np.float32, top_N_items=100,
                 nonlinear='linear', hidden_size=500, loss_func='log', loss_exp_p=1.005,
                 learning_rate_decay_factor=1.0, mw_eval_unmasked=True):
        # mw_eval_unmasked: step() only runs set_mask[loss_function] (hmf_model.py:209-210), so the
        # 'warp' mask of an 'mw' model's evaluation graph keeps its initial all-True value -- the
        # reference's eval loss of an 'mw' model does NOT mask the user's positives.  False: masked.
        self.mw_eval_unmasked = bool(mw_eval_unmasked)
        user_attributes.set_model_size(size)        # hmf_model.py:35
        item_attributes.set_model_size(size)        # :38
        self.loss_function = loss_function
        self.batch_size = batch_size
        self.n_sampled = n_sampled
        self.top_N_items = top_N_items
        self.learning_rate = float(learning_rate)
        self.learning_rate_decay_factor = learning_rate_decay_factor
        self.loss_func = loss_func
        self.loss_exp_p = loss_exp_p
        self.nonlinear = nonlinear
        self.global_step = 0
        self.att_emb = RefEmbeddingAttribute(user_attributes, item_attributes, batch_size,
                                             n_sampled, 0, False, item_ind2logit_ind,
                                             logit_ind2item_ind, params=params, dtype=dtype)
        self.dt = self.att_emb.dt
        if nonlinear in ('relu', 'tanh'):
            for k in ('w1', 'b1', 'w2', 'b2'):
                assert k in self.att_emb.params, 'MLP params must be supplied'

    def prepare_warp(self, pos, pos_eval):
        self.att_emb.prepare_warp(pos, pos_eval)

    def decay_learning_rate(self):
        self.learning_rate *= self.learning_rate_decay_factor   # hmf_model.py:57-58

    # ---- optional MLP (hmf_model.py:80-94), keep_prob == 1 ----
    def _act(self, x):
        return np.maximum(x, 0) if self.nonlinear == 'relu' else np.tanh(x)

    def _dact(self, y, x):
        return (x > 0).astype(self.dt) if self.nonlinear == 'relu' else 1 - y * y

    def _mlp_fwd(self, u, drops=None):
        """drops: None or three arrays mask/keep_prob (tf.nn.dropout after every activation)."""
        P = self.att_emb.params
        o = (lambda k, a: a * drops[k]) if drops is not None else (lambda k, a: a)
        h0 = self._act(u)
        h0d = o(0, h0)
        z1 = h0d @ P['w1'] + P['b1']
        h1 = self._act(z1)
        h1d = o(1, h1)
        z2 = h1d @ P['w2'] + P['b2']
        h2 = self._act(z2)
        return o(2, h2), (u, h0, h0d, z1, h1, h1d, z2, h2, drops)

    def _mlp_bwd(self, c, d_out, grads):
        P = self.att_emb.params
        u, h0, h0d, z1, h1, h1d, z2, h2, drops = c
        o = (lambda k, a: a * drops[k]) if drops is not None else (lambda k, a: a)
        dz2 = o(2, d_out) * self._dact(h2, z2)
        grads.add_dense('w2', h1d.T @ dz2)
        grads.add_dense('b2', dz2.sum(0))
        dh1 = o(1, dz2 @ P['w2'].T)
        dz1 = dh1 * self._dact(h1, z1)
        grads.add_dense('w1', h0d.T @ dz1)
        grads.add_dense('b1', dz1.sum(0))
        dh0 = o(0, dz1 @ P['w1'].T)
        return dh0 * self._dact(h0, u)

    def step(self, user_input, item_input, item_sampled=None, item_sampled_id2idx=None,
             forward_only=False, recommend=False, loss=None, keep_prob=1.0, user_mask=None,
             mlp_masks=None):
        """hmf_model.py:162-228 (session dropped).  tf.nn.dropout on the user embedding
        (embed_attribute.py:236, hmf_model.py:78): `user_mask` replays an externally drawn 0/1
        keep mask [mb, d] -- u * mask / keep_prob."""
        m = self.att_emb
        loss = loss or self.loss_function
        if item_sampled is not None and loss in ('mw', 'mce'):
            m.update_sampled(item_sampled)                          # :206-207
        u, c_user = m.get_batch_user(user_input, concat=False)      # :78
        drop = None
        if user_mask is not None and keep_prob < 1.0:
            drop = np.asarray(user_mask, dtype=self.dt) / self.dt.type(keep_prob)
            u = u * drop
        c_mlp = None
        if self.nonlinear in ('relu', 'tanh'):
            drops = None
            if mlp_masks is not None and keep_prob < 1.0 and not forward_only and not recommend:
                drops = [np.asarray(mk, dtype=self.dt) / self.dt.type(keep_prob) for mk in mlp_masks]
            u, c_mlp = self._mlp_fwd(u, drops)
        if recommend:
            logits, _ = m.get_prediction(u, 'full')
            # tf.nn.top_k(sorted=True): descending values, ties -> lower index first
            idx = np.argsort(-logits, axis=1, kind='stable')[:, :self.top_N_items]   # :154
            return idx.astype(np.int32)
        targets = m.target_mapping([item_input])[0]                 # :173
        if forward_only:
            # loss_eval: 'warp' over full V when training with 'mw' (:130,:144,:195); build-defined
            # 'mce' evaluates with the full softmax 'ce' (run_hmf.py:255,304 groups ce with mce)
            the_loss = 'warp' if loss == 'mw' else ('ce' if loss == 'mce' else loss)
            logits, _ = m.get_prediction(u, 'full')
            mask = None
            if the_loss != 'ce':
                if loss == 'mw' and self.mw_eval_unmasked:
                    mask = np.ones(logits.shape, dtype=bool)        # the 'warp' mask variable's initial value
                else:
                    mask = m.mask(user_input, the_loss, None, forward_only=True)
            bl, _ = m.compute_loss(logits, targets, the_loss, mask, self.loss_func, self.loss_exp_p)
            return self.dt.type(bl.mean())
        if loss in ('mw', 'mce'):
            logits, c_pred = m.get_prediction(u, 'sampled')         # :112
            tscore, c_t = m.get_target_score(u, item_input)         # :115
            mask = m.mask(user_input, loss, item_sampled_id2idx)
            bl, c_loss = m.compute_loss(logits, tscore, loss, mask)
        else:
            logits, c_pred = m.get_prediction(u, 'full')            # :118
            mask = None
            if loss != 'ce':
                mask = m.mask(user_input, loss)
            bl, c_loss = m.compute_loss(logits, targets, loss, mask, self.loss_func, self.loss_exp_p)
            c_t = None
        mb = len(user_input)
        the_loss = bl.mean()                                        # :140
        # ---- tf.gradients(self.loss, params) (:149) ----
        grads = Grads()
        d_bl = np.full((mb,), 1.0 / mb, dtype=self.dt)
        d_logits, d_t = m.compute_loss_bwd(c_loss, d_bl)
        d_u = m.get_prediction_bwd(c_pred, d_logits, grads)
        if c_t is not None:
            d_u = d_u + m.get_target_score_bwd(c_t, d_t, grads)
        if c_mlp is not None:
            d_u = self._mlp_bwd(c_mlp, d_u, grads)
        if drop is not None:
            d_u = d_u * drop
        m.get_batch_user_bwd(c_user, d_u, grads)
        m.apply_gradients(grads, self.learning_rate)                # :150
        self.global_step += 1
        # (kept for the tests' condition-aware bounds: the un-merged contributions and what produced them)
        self.last = {'logits': logits, 'batch_loss': bl, 'mask': mask, 'u': u, 'd_logits': d_logits,
                     'c_pred': c_pred, 'grads': grads}
        return self.dt.type(the_loss)
