"""numpy restatement of the LSTM sequence model (lstm/seqModel.py).

TEST INFRASTRUCTURE -- see oracle/__init__.py ("parity unpinned": TF-1.0's
LSTMCell / static_rnn / clip_by_global_norm semantics are restated from the
reference's call sites and documented op behaviour).
"""
from __future__ import annotations

import numpy as np

from .ref_graph import Grads, RefEmbeddingAttribute, adagrad_apply


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_fwd(x, W, b, forget_bias=1.0):
    """seqModel.py:99 LSTMCell + :477 static_rnn from zero state.
    x: [L,B,din]; W: [din+h, 4h]; b: [4h]; gate order i, j, f, o
    (TF-1.0 core_rnn_cell_impl.LSTMCell: c = sigmoid(f+forget_bias)*c_prev +
    sigmoid(i)*tanh(j); h = sigmoid(o)*tanh(c))."""
    L, B, din = x.shape
    h = W.shape[1] // 4
    dt = x.dtype
    hs = np.zeros((L, B, h), dt)
    cs = np.zeros((L, B, h), dt)
    gates = np.zeros((L, B, 4 * h), dt)
    hp = np.zeros((B, h), dt)
    cp = np.zeros((B, h), dt)
    for t in range(L):
        z = np.concatenate([x[t], hp], 1) @ W + b
        i, j, f, o = z[:, :h], z[:, h:2 * h], z[:, 2 * h:3 * h], z[:, 3 * h:]
        gi, gj, gf, go = _sig(i), np.tanh(j), _sig(f + forget_bias), _sig(o)
        c = gf * cp + gi * gj
        hh = go * np.tanh(c)
        hs[t], cs[t] = hh, c
        gates[t] = np.concatenate([gi, gj, gf, go], 1)
        hp, cp = hh, c
    return hs, cs, gates


def lstm_bwd(x, W, hs, cs, gates, dhs):
    """BPTT.  Returns dz [L,B,4h], dx [L,B,din], dW, db."""
    L, B, din = x.shape
    h = hs.shape[2]
    dt = x.dtype
    dz = np.zeros((L, B, 4 * h), dt)
    dh_rec = np.zeros((B, h), dt)
    dc = np.zeros((B, h), dt)
    for t in range(L - 1, -1, -1):
        gi, gj, gf, go = (gates[t][:, :h], gates[t][:, h:2 * h], gates[t][:, 2 * h:3 * h],
                          gates[t][:, 3 * h:])
        c = cs[t]
        cp = cs[t - 1] if t > 0 else np.zeros_like(c)
        dh = dhs[t] + dh_rec
        tc = np.tanh(c)
        dcc = dc + dh * go * (1 - tc * tc)
        dz[t] = np.concatenate([dcc * gj * gi * (1 - gi), dcc * gi * (1 - gj * gj),
                                dcc * cp * gf * (1 - gf), dh * tc * go * (1 - go)], 1)
        dc = dcc * gf
        dh_rec = dz[t] @ W[din:].T
    dx = dz @ W[:din].T
    hprev = np.concatenate([np.zeros((1, B, h), dt), hs[:-1]], 0)
    inp = np.concatenate([x, hprev], 2).reshape(L * B, din + h)
    dW = inp.T @ dz.reshape(L * B, 4 * h)
    db = dz.reshape(L * B, 4 * h).sum(0)
    return dz, dx, dW, db


class RefSeqModel(object):
    """lstm/seqModel.py:26-184 (single bucket, num_layers=1, keep_prob=1,
    use_concat=False) restated: inputs (u + mean_f(item_t))/2 (:148-156),
    LSTM, per-step scorer + loss (:480-493), sequence_loss (:524-604),
    clip_by_global_norm (:180), Adagrad (:174,182)."""

    def __init__(self, L, size, max_gradient_norm, batch_size, learning_rate, embAttr,
                 loss='mw', no_user_id=False, no_input_item_feature=False, output_feat=1,
                 params=None, withAdagrad=True, use_concat=False, num_layers=1):
        self.L = L
        self.size = size
        self.max_gradient_norm = max_gradient_norm
        self.batch_size = batch_size
        self.learning_rate = float(learning_rate)
        self.m = embAttr
        self.loss = loss
        self.no_user_id = no_user_id
        self.no_input_item_feature = no_input_item_feature
        self.output_feat = output_feat
        self.withAdagrad = withAdagrad
        dt = embAttr.dt
        self.dt = dt
        # MultiRNNCell([cell] * num_layers) (:99-103): one (W, b) per layer, every layer size -> size
        self.num_layers = num_layers
        self.Ws, self.bs, self.wnames, self.bnames = [], [], [], []
        for l in range(num_layers):
            wn, bn = ('lstm_w', 'lstm_b') if l == 0 else ('lstm_w_%d' % l, 'lstm_b_%d' % l)
            W = np.array(params[wn], dtype=dt)            # [din+h, 4h]
            b = np.array(params[bn], dtype=dt)            # [4h]
            embAttr.params[wn], embAttr.params[bn] = W, b
            embAttr.slots[wn] = np.full(W.shape, 0.1, dt)
            embAttr.slots[bn] = np.full(b.shape, 0.1, dt)
            self.Ws.append(W); self.bs.append(b); self.wnames.append(wn); self.bnames.append(bn)
        self.W, self.b = self.Ws[0], self.bs[0]
        # use_concat (:130-146): input_t = concat_f(user) . w_input_user + concat_f(item_t) . w_input_item
        self.use_concat = use_concat
        if use_concat:
            for nm in ('w_input_user', 'w_input_item'):
                w = np.array(params[nm], dtype=dt)
                embAttr.params[nm] = w
                embAttr.slots[nm] = np.full(w.shape, 0.1, dt)
        self.last = {}

    def step_recommend(self, user_input, item_inputs, positions, topk_n=30):
        """seqModel.py:326-353 + :514-517: per position top_k(softmax(full logits), topk_n,
        sorted=True) -> [(uid, values[topk_n], indexes[topk_n])] for position positions[i] of
        sequence i (ties: lower index first, like tf.nn.top_k)."""
        m, L = self.m, self.L
        xs = []
        if self.use_concat:
            Wu, Wi = m.params['w_input_user'], m.params['w_input_item']
            ut = 0.0
            if Wu.shape[0] > 0:
                ut = m.get_batch_user(user_input, concat=True, no_id=self.no_user_id)[0] @ Wu
            for t in range(L):
                it = m.get_batch_item(item_inputs[t], concat=True,
                                      no_attribute=self.no_input_item_feature)[0]
                xs.append(ut + it @ Wi)
        else:
            u, _ = m.get_batch_user(user_input, concat=False, no_id=self.no_user_id)
            for t in range(L):
                feats, _, _ = m.get_batch_item(item_inputs[t], concat=False,
                                               no_attribute=self.no_input_item_feature)
                it = np.mean(np.stack(feats, 0), 0)
                xs.append(np.mean(np.stack([u, it], 0), 0))
        hs = np.stack(xs, 0)
        for l in range(self.num_layers):                                    # keep_prob == 1 when recommending
            hs, _, _ = lstm_fwd(hs, self.Ws[l], self.bs[l], 1.0)
        results = []
        for i, pos in enumerate(positions):
            logits, _ = m.get_prediction(hs[pos], 'full', self.output_feat)
            x = logits[i]
            pr = np.exp(x - x.max())
            pr = pr / pr.sum()
            idx = np.argsort(-pr, kind='stable')[:topk_n]
            results.append((user_input[i], pr[idx], idx.astype(np.int32)))
        return results

    def step(self, user_input, item_inputs, targets, target_weights, item_sampled=None,
             item_sampled_id2idx=None, forward_only=False, keep_prob=1.0, masks=None):
        """seqModel.py:289-324; inputs are time-major python lists [L][B].
        Dropout (DropoutWrapper, :100,103): `masks` = {'in': [0/1 array [L,B,size] per layer],
        'out': 0/1 array [L,B,size]} replays externally drawn keep masks -- the layer input is
        x * mask / keep_prob (input_keep_prob, every layer), the top output likewise
        (output_keep_prob); the recurrent state is never dropped."""
        m, L, B, dt = self.m, self.L, self.batch_size, self.dt
        if item_sampled is not None and self.loss in ('mw', 'mce'):
            m.update_sampled(item_sampled)                                  # :306-307
        targets_mapped = m.target_mapping(targets)                          # :294
        w = np.asarray(target_weights, dtype=dt)                            # [L,B]
        # ---- inputs (:148-156) ----
        xs, c_items, its = [], [], []
        if self.use_concat:
            Wu, Wi = m.params['w_input_user'], m.params['w_input_item']
            if Wu.shape[0] > 0:
                u, c_user = m.get_batch_user(user_input, concat=True, no_id=self.no_user_id)   # :131
                ut = u @ Wu                                                                 # :138
            else:                                   # no_user_id with an id-only user: zero-width embed
                u, c_user, ut = np.zeros((B, 0), dtype=dt), {'sites': []}, 0.0
            for t in range(L):
                it, _, c_it = m.get_batch_item(item_inputs[t], concat=True,
                                               no_attribute=self.no_input_item_feature)     # :142
                xs.append(ut + it @ Wi)                                                     # :144-145
                its.append(it)
                c_items.append(c_it)
        else:
            u, c_user = m.get_batch_user(user_input, concat=False, no_id=self.no_user_id)
            for t in range(L):
                feats, _, c_it = m.get_batch_item(item_inputs[t], concat=False,
                                                  no_attribute=self.no_input_item_feature)
                it = np.mean(np.stack(feats, 0), 0)                             # :154
                xs.append(np.mean(np.stack([u, it], 0), 0))                     # :155
                c_items.append(c_it)
        x = np.stack(xs, 0)
        kp = dt.type(keep_prob)
        drop = masks is not None and keep_prob < 1.0 and not forward_only
        layers = []
        inp = x
        for l in range(self.num_layers):
            xin = inp * (np.asarray(masks['in'][l], dtype=dt) / kp) if drop else inp
            hs_l, cs_l, gates_l = lstm_fwd(xin, self.Ws[l], self.bs[l], 1.0)       # :477
            layers.append((xin, hs_l, cs_l, gates_l))
            inp = hs_l
        hs = inp * (np.asarray(masks['out'], dtype=dt) / kp) if drop else inp
        # ---- per-step scorer + loss (:480-493) ----
        the_loss = self.loss
        if forward_only:
            # losses_full (:510); the build-defined 'mce' evaluates with the full softmax 'ce'
            the_loss = 'warp' if self.loss == 'mw' else ('ce' if self.loss == 'mce' else self.loss)
        mask = None
        if the_loss in ('mw', 'mce'):
            mask = m.mask(user_input, the_loss, item_sampled_id2idx)
        elif the_loss == 'warp':
            mask = m.mask(user_input, 'warp', None, forward_only=forward_only)
        bls, caches = [], []
        for t in range(L):
            if the_loss in ('mw', 'mce'):
                logits, c_p = m.get_prediction(hs[t], 'sampled', self.output_feat)
                ts, c_t = m.get_target_score(hs[t], targets[t])
                bl, c_l = m.compute_loss(logits, ts, the_loss, mask)
            else:
                logits, c_p = m.get_prediction(hs[t], 'full', self.output_feat)
                c_t = None
                bl, c_l = m.compute_loss(logits, targets_mapped[t], the_loss, mask)
            bls.append(bl)
            caches.append((c_p, c_t, c_l))
        bls = np.stack(bls, 0)                                              # [L,B]
        # sequence_loss_by_example (:551-567) + reduce_sum over batch (:596)
        total_size = w.sum(0) + 1e-12
        log_perps = (bls * w).sum(0) / total_size
        cost = log_perps.sum()
        self.last = {'batch_loss': bls, 'hs': hs, 'x': x}
        if forward_only:
            return dt.type(cost)
        # ---- backward ----
        grads = Grads()
        d_bls = w / total_size                                              # [L,B]
        dhs = np.zeros_like(hs)
        for t in range(L):
            c_p, c_t, c_l = caches[t]
            d_logits, d_t = m.compute_loss_bwd(c_l, d_bls[t])
            dh = m.get_prediction_bwd(c_p, d_logits, grads)
            if c_t is not None:
                dh = dh + m.get_target_score_bwd(c_t, d_t, grads)
            dhs[t] = dh
        if drop:
            dhs = dhs * (np.asarray(masks['out'], dtype=dt) / kp)
        for l in reversed(range(self.num_layers)):
            xin, hs_l, cs_l, gates_l = layers[l]
            dz, dx, dW, db = lstm_bwd(xin, self.Ws[l], hs_l, cs_l, gates_l, dhs)
            grads.add_dense(self.wnames[l], dW)
            grads.add_dense(self.bnames[l], db)
            if drop:
                dx = dx * (np.asarray(masks['in'][l], dtype=dt) / kp)
            dhs = dx
        if self.use_concat:
            Wu, Wi = m.params['w_input_user'], m.params['w_input_item']
            dxs = dx.sum(0)
            for t in range(L):
                grads.add_dense('w_input_item', its[t].T @ dx[t])        # one dense grad per unrolled matmul
                d_it = dx[t] @ Wi.T
                c_it = c_items[t]
                if len(c_it['sites']):
                    offs = np.cumsum([0] + c_it['dims'])
                    m.get_embedded_bwd(c_it, [d_it[:, offs[k]:offs[k + 1]] for k in range(len(c_it['dims']))],
                                       None, grads)
            if Wu.shape[0] > 0:
                grads.add_dense('w_input_user', u.T @ dxs)
                if c_user['sites']:
                    m.get_batch_user_bwd(c_user, dxs @ Wu.T, grads)
        else:
            du = np.zeros_like(u)
            for t in range(L):
                du += dx[t] / 2
                c_it = c_items[t]
                n = len(c_it['sites'])
                if n:
                    m.get_embedded_bwd(c_it, [dx[t] / 2 / n] * n, None, grads)
            if c_user['sites']:
                m.get_batch_user_bwd(c_user, du, grads)
        # clip_by_global_norm (:180): norm over the aggregated-but-unmerged values
        sq = sum(grads.sq_norm_unmerged(n) for n in grads.names())
        gnorm = np.sqrt(sq)
        scale = self.max_gradient_norm / max(gnorm, self.max_gradient_norm)
        self.last['gnorm'] = gnorm
        if self.withAdagrad:
            m.apply_gradients(grads, self.learning_rate, scale=scale)
        else:
            for name in grads.names():
                p = m.params[name]
                p -= dt.type(self.learning_rate * scale) * grads.total(name, p.shape, dt)
        return dt.type(cost)
