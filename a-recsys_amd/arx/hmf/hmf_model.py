"""LatentProductModel -- the reference's HMF model (hmf/hmf_model.py:19-274) on MI355X.

Constructor / step / get_batch / get_permuted_batch keep the reference's
signatures; `session`, `GPU`, `run_op`, `run_meta` are accepted and ignored.
State (tables, Adagrad slots, mask, sampled pool) lives on the device and is
owned by the model.  step() is forward + backward + Adagrad as ONE captured
hipGraph replay of libarx.so kernels.
"""
from __future__ import annotations

import random

import os

import numpy as np
import torch

from .. import graph as G
from .. import ops
from ..attributes import embed_attribute
from ..utils.checkpoint import Saver


class _Var(object):
    """Stand-in for the tf.Variable handles the runners read with .eval()."""

    def __init__(self, getter):
        self._getter = getter

    def eval(self, session=None):
        return self._getter()


class _Op(object):
    def __init__(self, fn):
        self._fn = fn

    def run(self, session=None):
        return self._fn()

    __call__ = run


class MLP(G.Node):
    """hmf_model.py:80-94: drop(act(drop(act(drop(act(u)) . W1 + b1)) . W2 + b2)) -- tf.nn.dropout
    after every activation with the model's keep probability (identity in forward-only plans
    and when rt.keep_prob == 1).  The three keep masks stay readable in `keeps` (parity tests
    replay them through the oracle)."""

    requires_grad = True
    uses_dropout = True

    def __init__(self, rt, x, params, kind):
        self.w1, self.b1, self.w2, self.b2 = params
        super().__init__(rt, (x.shape[0], self.w2.w.shape[1]), (x,))
        self.kind = 0 if kind == 'relu' else 1
        n, hdim = x.shape[0], self.w1.w.shape[1]
        dev = rt.device
        f32 = torch.float32
        self.h0 = torch.empty(x.shape, dtype=f32, device=dev)          # act(u)
        self.h1 = torch.empty((n, hdim), dtype=f32, device=dev)        # act(z1)
        self.h2 = torch.empty(self.shape, dtype=f32, device=dev)       # act(z2)
        self.h0d, self.h1d = torch.empty_like(self.h0), torch.empty_like(self.h1)   # after dropout
        self.d1 = torch.empty((n, hdim), dtype=f32, device=dev)
        self.d0 = torch.empty(x.shape, dtype=f32, device=dev)
        self.keeps = [torch.empty(t.numel(), dtype=torch.uint8, device=dev) for t in (self.h0, self.h1, self.h2)]
        self.sids = []
        for _ in range(3):
            rt.dropout_calls += 1
            self.sids.append(rt.dropout_calls)
        self.kp_used = 1.0

    def _drop(self, k, src, dst, kp):
        ops.dropout_fwd_step(src, kp, self.rt.seed * 1000003 + self.sids[k], self.rt.step_dev, dst, self.keeps[k])

    def forward(self, train):
        x = self.inputs[0]
        rt = self.rt
        out = self.alloc_value()
        kp = rt.keep_prob if train else 1.0
        self.kp_used = kp
        drop = kp < 1.0
        ops.act_fwd(x.value, self.kind, self.h0)
        a0 = self.h0
        if drop:
            self._drop(0, self.h0, self.h0d, kp)
            a0 = self.h0d
        ops.gemm(a0, self.w1.w, self.h1, rt.ws, col_bias=self.b1.w)
        ops.act_fwd(self.h1, self.kind, self.h1)
        a1 = self.h1
        if drop:
            self._drop(1, self.h1, self.h1d, kp)
            a1 = self.h1d
        h2 = self.h2 if drop else out
        ops.gemm(a1, self.w2.w, h2, rt.ws, col_bias=self.b2.w)
        ops.act_fwd(h2, self.kind, h2)
        if drop:
            self._drop(2, self.h2, out, kp)

    def backward(self):
        x = self.inputs[0]
        rt = self.rt
        kp = self.kp_used
        drop = kp < 1.0
        dz2 = self.grad
        if drop:
            ops.dropout_bwd(dz2, self.keeps[2], kp, dz2)
        ops.act_bwd(self.h2 if drop else self.value, dz2, self.kind, dz2)      # through act(z2)
        a1 = self.h1d if drop else self.h1
        a0 = self.h0d if drop else self.h0
        ops.gemm(a1, dz2, self.w2.grad, rt.ws, transA=True)
        ops.col_sum(dz2, self.b2.grad, rt.ws)
        ops.gemm(dz2, self.w2.w, self.d1, rt.ws, transB=True)
        if drop:
            ops.dropout_bwd(self.d1, self.keeps[1], kp, self.d1)
        ops.act_bwd(self.h1, self.d1, self.kind, self.d1)
        ops.gemm(a0, self.d1, self.w1.grad, rt.ws, transA=True)
        ops.col_sum(self.d1, self.b1.grad, rt.ws)
        ops.gemm(self.d1, self.w1.w, self.d0, rt.ws, transB=True)
        if drop:
            ops.dropout_bwd(self.d0, self.keeps[0], kp, self.d0)
        ops.act_bwd(self.h0, self.d0, self.kind, self.d0)
        g = x.alloc_grad()
        ops.add_rows_bcast(1.0, self.d0, x.grad_beta(), g)
        for p in (self.w1, self.b1, self.w2, self.b2):
            p.touched = True


class TopK(G.Node):
    """hmf_model.py:154 tf.nn.top_k(logits, top_N_items, sorted=True)."""

    def __init__(self, rt, logits, k):
        super().__init__(rt, (logits.shape[0], k), (logits,))
        self.k = k
        self.indices = torch.empty((logits.shape[0], k), dtype=torch.int32, device=rt.device)

    def forward(self, train):
        ops.topk(self.inputs[0].value, self.k, self.alloc_value(), self.indices)


class StreamTopK(G.Node):
    """top_k over the FULL vocabulary without the [mb, V] logits -- same indices / values as
    TopK(Prediction), tf.nn.top_k's tie rule included.
    fused (round 5, the default where the scorer GEMM's small-K kernel applies): the first chunk of the pool gives
    every row its k best (GEMM -> radix select); the scorer GEMM over ALL the other columns then writes no logits --
    arx_gemm_nt_topk_filter keeps only what beats the row's k-th best so far, as short candidate lists in column order;
    one select over the lists and one merge finish.  A candidate list that overflows (scores rising along the
    vocabulary) sets a flag: overflowed() -- LatentProductModel.step re-runs the request on the chunked path.
    chunked: the GEMM runs over chunks of the pool rows, every chunk keeps its k best per row (radix select) and a
    merge folds them into the running result."""

    def __init__(self, rt, latent, pool, k, chunk=65536, want_lse=False):
        super().__init__(rt, (latent.shape[0], k), (latent, pool))
        self.k, self.chunk = k, max(int(chunk), k)
        # want_lse: also self.lse [B] = log sum exp over ALL the row's logits (seqModel.py:514-517 reports the winners'
        # softmax values exp(v - lse)): per column range out of the fused GEMM / per chunk, combined at the end
        self.want_lse = bool(want_lse)
        self.lse = torch.empty(latent.shape[0], dtype=torch.float32, device=rt.device) if want_lse else None
        self._lse_parts = None
        B, V, dev = latent.shape[0], pool.shape[0], rt.device
        f32, i32 = torch.float32, torch.int32
        self.indices = torch.empty((B, k), dtype=i32, device=dev)
        self._buf = torch.empty((B, min(self.chunk, V)), dtype=f32, device=dev)
        self._cv, self._ci = torch.empty((B, k), dtype=f32, device=dev), torch.empty((B, k), dtype=i32, device=dev)
        self._ov, self._oi = torch.empty((B, k), dtype=f32, device=dev), torch.empty((B, k), dtype=i32, device=dev)
        tail = V % self.chunk                       # a last chunk narrower than k keeps only `tail` entries
        kt = tail if 0 < tail < k else k
        self._tv, self._ti = torch.empty((B, kt), dtype=f32, device=dev), torch.empty((B, kt), dtype=i32, device=dev)
        self.fused = (os.environ.get('ARX_TOPK_FUSED', '1') != '0' and latent.shape[1] in (32, 64, 128)
                      and pool.shape[1] == latent.shape[1])
        self.overflow = torch.zeros(1, dtype=i32, device=dev)
        self.slack, self.min_capp = 4.0, 32          # candidate segment = slack x the expected survivors, >= min_capp
        self._cand = None

    def _cand_bufs(self, n0, V):
        """Candidate rows [B, parts * capp]: a column range's expected survivors are k (V - n0) / n0 / parts (scores in
        no particular order along the vocabulary); four times that, at least 32."""
        key = (n0, V, self.slack, self.min_capp)
        if self._cand is None or self._cand[0] != key:
            B, k, dev = self.shape[0], self.k, self.rt.device
            parts = ops.gemm_nt_topk_parts(B, V - n0)
            expect = k * (V - n0) / float(n0) / parts
            capp = self.min_capp
            while capp < self.slack * expect + self.min_capp:
                capp *= 2
            while parts * capp < k:
                capp *= 2
            cap = parts * capp
            self._cand = (key, capp, torch.empty((B, cap), dtype=torch.float32, device=dev),
                          torch.zeros((B, cap), dtype=torch.int32, device=dev),
                          torch.empty((B, k), dtype=torch.int32, device=dev), parts)
        return self._cand[1:]

    def _lse_buf(self, ncols):
        if self._lse_parts is None or self._lse_parts.shape[1] != ncols:
            B, dev = self.shape[0], self.rt.device
            self._lse_parts = torch.empty((B, ncols), dtype=torch.float32, device=dev)
            self._lse0 = torch.empty(B, dtype=torch.float32, device=dev)
        return self._lse_parts

    def overflowed(self):
        """True when the last fused run dropped candidates (device -> host read)."""
        return bool(self.fused and int(self.overflow.item()) != 0)

    def forward(self, train):
        latent, pool = self.inputs
        V, k = pool.shape[0], self.k
        run_v, run_i = self.alloc_value(), self.indices
        out_v, out_i = self._ov, self._oi
        if self.fused and V > self.chunk and pool.value.stride(0) % 4 == 0 and latent.value.stride(0) % 4 == 0:
            n0 = self.chunk
            lg = self._buf[:, :n0]
            bias = pool.bias_value
            ops.gemm(latent.value, pool.value[:n0], lg, self.rt.ws, transB=True,
                     col_bias=bias[:n0] if bias is not None else None)
            ops.topk_chunk(lg, k, 0, run_v, run_i)
            capp, cand_v, cand_i, cpos, parts = self._cand_bufs(n0, V)
            lp = None
            if self.want_lse:
                lp = self._lse_buf(parts + 1)
                ops.row_logsumexp(lg, self._lse0)
                lp[:, parts].copy_(self._lse0)
            ops.fill_f32(cand_v.view(-1), float('-inf'))
            ops.fill_i32(self.overflow, 0)
            ops.gemm_nt_topk_filter(latent.value, pool.value[n0:], bias[n0:] if bias is not None else None,
                                    run_v[:, k - 1], n0, cand_v, cand_i, capp, self.overflow, lse_part=lp)
            if self.want_lse:
                ops.row_logsumexp(lp, self.lse)
            ops.topk_chunk(cand_v, k, 0, self._cv, cpos)
            ops.take_rows_i32(cand_i, cpos, self._ci)
            ops.topk_merge(run_v, run_i, self._cv, self._ci, k, out_v, out_i)
            self.value.copy_(out_v)
            self.indices.copy_(out_i)
            return
        nch = (V + self.chunk - 1) // self.chunk
        lp = self._lse_buf(nch) if self.want_lse else None
        for c0 in range(0, V, self.chunk):
            c1 = min(V, c0 + self.chunk)
            kc = min(k, c1 - c0)
            lg = self._buf[:, :c1 - c0]
            bias = pool.bias_value[c0:c1] if pool.bias_value is not None else None
            ops.gemm(latent.value, pool.value[c0:c1], lg, self.rt.ws, transB=True, col_bias=bias)
            if self.want_lse:
                ops.row_logsumexp(lg, self._lse0)
                lp[:, c0 // self.chunk].copy_(self._lse0)
            if c0 == 0:                              # chunk >= k and V >= k: the first chunk fills all k
                ops.topk_chunk(lg, k, 0, run_v, run_i)
                continue
            cv, ci = (self._cv, self._ci) if kc == k else (self._tv, self._ti)
            ops.topk_chunk(lg, kc, c0, cv, ci)
            ops.topk_merge(run_v, run_i, cv, ci, k, out_v, out_i)
            run_v, out_v = out_v, run_v
            run_i, out_i = out_i, run_i
        if run_v.data_ptr() != self.value.data_ptr():
            self.value.copy_(run_v)
            self.indices.copy_(run_i)
        if self.want_lse:
            ops.row_logsumexp(lp, self.lse)


class LatentProductModel(object):
    def __init__(self, user_size, item_size, size, num_layers, batch_size, learning_rate,
                 learning_rate_decay_factor, user_attributes=None, item_attributes=None,
                 item_ind2logit_ind=None, logit_ind2item_ind=None, loss_function='ce', GPU=None,
                 logit_size_test=None, nonlinear=None, dropout=1.0, n_sampled=None,
                 indices_item=None, dtype='float32', top_N_items=100, hidden_size=500,
                 loss_func='log', loss_exp_p=1.005, params=None, use_graph=True, seed=0,
                 mw_eval_unmasked=True):
        self.user_size = user_size
        self.item_size = item_size
        self.top_N_items = top_N_items
        # 'mw' models evaluate with the full-vocabulary 'warp' loss (:130,144).  The reference's
        # step() only runs set_mask['mw'] (:209-210), so ITS eval loss sees an all-True warp mask
        # (the target column itself adds relu(0 + 1) = 1 inside the log).  That is the default
        # (run_hmf.py selects checkpoints, patience and learning-rate decay on this number);
        # mw_eval_unmasked=False masks the user's eval positives instead (what the masks are for).
        self.mw_eval_unmasked = bool(mw_eval_unmasked)
        if user_attributes is not None:
            user_attributes.set_model_size(size)             # hmf_model.py:34-36
            self.user_attributes = user_attributes
        if item_attributes is not None:
            item_attributes.set_model_size(size)
            self.item_attributes = item_attributes
        self.item_ind2logit_ind = item_ind2logit_ind
        self.logit_ind2item_ind = logit_ind2item_ind
        if logit_ind2item_ind is not None:
            self.logit_size = len(logit_ind2item_ind)
        self.indices_item = indices_item if indices_item is not None else range(self.logit_size)
        self.logit_size_test = logit_size_test
        self.nonlinear = nonlinear
        self.loss_function = loss_function
        self.n_sampled = n_sampled
        self.batch_size = batch_size
        self.dropout = dropout
        self.dtype = dtype
        self.data_length = None
        self.train_permutation = None
        self.start_index = None
        if loss_function in ('bpr', 'bpr-hinge'):
            raise NotImplementedError("bpr losses: the reference's pos/neg feeds are commented out "
                                      "(embed_attribute.py:704-706); not on the hot path")
        self.loss_func, self.loss_exp_p = loss_func, loss_exp_p

        self.rt = rt = G.Runtime(learning_rate=learning_rate, use_graph=use_graph)
        self._lr_decay = learning_rate_decay_factor
        self.learning_rate = _Var(lambda: rt.lr_host)
        self.learning_rate_decay_op = _Op(lambda: rt.set_learning_rate(rt.lr_host * self._lr_decay))
        self.global_step = _Var(lambda: rt.global_step)

        mb = batch_size
        # mapped item target (logit index) and raw item id target (hmf_model.py:69-70)
        self.item_id_target = G.IdsInput(rt, mb, 'item_id')
        self.item_target = None      # built below: the in-plan mapping of item_id_target (needs att_emb)

        m = embed_attribute.EmbeddingAttribute(user_attributes, item_attributes, mb, self.n_sampled,
                                               0, False, item_ind2logit_ind, logit_ind2item_ind,
                                               params=params, runtime=rt, seed=seed)
        self.att_emb = m
        self.item_target = embed_attribute.TargetMapping(rt, self.item_id_target, m, 'item')    # :69,173
        embedded_user, _ = m.get_batch_user(float(dropout), False)          # :78
        if self.nonlinear in ('relu', 'tanh'):
            ps = []
            for name, shape in (('w1', (size, hidden_size)), ('b1', (hidden_size,)),
                                ('w2', (hidden_size, size)), ('b2', (size,))):
                w = m._new_var(name, shape, params or {})
                p = G.DenseParam(name, w)
                rt.dense[name] = p
                ps.append(p)
            embedded_user, _ = m.get_batch_user(1.0, False)                  # :87
            rt.keep_prob = float(dropout)                                    # :88-94 dropout inside the MLP
            embedded_user = MLP(rt, embedded_user, ps, self.nonlinear)
        self.embedded_user = embedded_user

        loss = self.loss_function
        self._plans = {}
        self.set_mask, self.reset_mask = {}, {}
        sampled_logits = target_score = None
        if self.n_sampled is not None:
            sampled_logits = m.get_prediction(embedded_user, 'sampled')       # :112
            target_score = m.get_target_score(embedded_user, self.item_id_target)  # :115
        logits = m.get_prediction(embedded_user)                              # :118
        self.output = logits
        batch_loss_eval = None
        # [mb, V] logits of the evaluation loss are streamed, not materialised, past this size
        stream_eval = batch_size * self.logit_size * 4 > int(os.environ.get('ARX_STREAM_TOPK_BYTES', str(1 << 30)))
        if loss in ('warp', 'ce', 'rs', 'rs-sig', 'rs-sig2', 'bbpr'):            # :121-122
            batch_loss = m.compute_loss(logits, self.item_target, loss, loss_func=self.loss_func,
                                        exp_p=self.loss_exp_p)
        elif loss == 'warp_eval':
            batch_loss, _ = m.compute_loss(logits, self.item_target, loss)
        elif loss == 'mw' and stream_eval:
            batch_loss = m.compute_loss(sampled_logits, target_score, loss)
            ms = None if self.mw_eval_unmasked else m._mask_state('warp', batch_size)
            batch_loss_eval = G.StreamEvalLoss(rt, 'warp', embedded_user, m._pool_embed('full', 1),
                                               self.item_target, mask=ms, mask_rows=batch_size)
        elif loss == 'mce' and stream_eval:
            batch_loss = m.compute_loss(sampled_logits, target_score, loss)
            batch_loss_eval = G.StreamEvalLoss(rt, 'ce', embedded_user, m._pool_embed('full', 1), self.item_target)
        elif loss == 'mw':
            batch_loss = m.compute_loss(sampled_logits, target_score, loss)
            if self.mw_eval_unmasked:
                batch_loss_eval = G.BatchLoss(rt, 'warp', logits, self.item_target, mask=None,
                                              mask_rows=batch_size)
            else:
                batch_loss_eval = m.compute_loss(logits, self.item_target, 'warp')  # :130
        elif loss == 'mce':
            # build-defined sampled softmax (the reference has no arithmetic for 'mce', see arx.h):
            # trains like 'mw' on the sampled pool; evaluates with the full softmax 'ce', the loss
            # run_hmf.py:255,304 groups it with
            batch_loss = m.compute_loss(sampled_logits, target_score, loss)
            batch_loss_eval = m.compute_loss(logits, self.item_target, 'ce')
        else:
            raise NotImplementedError("not implemented!")
        if loss in ('warp', 'warp_eval', 'mw', 'mce', 'rs', 'rs-sig', 'rs-sig2', 'bbpr'):   # :137
            self.set_mask, self.reset_mask = m.get_warp_mask()
        self.batch_loss = batch_loss
        self.loss = G.MeanLoss(rt, batch_loss)                                # :140
        self.loss.lazy = True
        self.loss_eval = G.MeanLoss(rt, batch_loss_eval) if loss in ('mw', 'mce') else self.loss  # :144
        kk = min(self.top_N_items, self.logit_size)
        stream_min = int(os.environ.get('ARX_STREAM_TOPK_BYTES', str(1 << 30)))
        lat_w = logits.inputs[0].shape[1] if hasattr(logits, 'inputs') and logits.inputs else 0
        fused_ok = self.logit_size > 65536 and lat_w in (32, 64, 128) and isinstance(logits, G.Prediction)
        if kk <= 1024 and (batch_size * self.logit_size * 4 > stream_min or fused_ok):
            # [mb, V] is not worth materialising: streaming scorer + top-k.  Past 65 536 items the fused form beats the
            # materialising one at every batch size (V = 1 M: 1.0 against 8.1 ms at mb = 64, 3.3 against 13.8 at 1 024)
            self.topk = StreamTopK(rt, logits.inputs[0], logits.inputs[1], kk)
        else:
            self.topk = TopK(rt, logits, kk)                                     # :154
        self.indices = self.topk
        self.saver = Saver(self)

    # ------------------------------------------------------------------
    def prepare_warp(self, pos_item_set, pos_item_set_eval):
        self.att_emb.prepare_warp(pos_item_set, pos_item_set_eval)

    def _plan(self, key):
        if key in self._plans:
            return self._plans[key]
        rt, m = self.rt, self.att_emb
        loss = self.loss_function
        if key == 'train':
            masks = [m.mask[loss]] if loss in m.mask else []
            p = G.Plan(rt, [self.loss], True, masks)
        elif key == 'eval':
            l = 'warp' if loss == 'mw' else ('ce' if loss == 'mce' else loss)
            masks = [m.mask[l]] if l in m.mask else []
            if isinstance(self.loss_eval.inputs[0], G.StreamEvalLoss):
                masks = []                 # the streaming loss reads the positives CSR itself
            p = G.Plan(rt, [self.loss_eval], False, masks)
        elif key == 'recommend':
            p = G.Plan(rt, [self.topk], False, [])
        elif key == 'warp_eval':
            p = G.Plan(rt, [self.batch_loss], False, [m.mask['warp_eval']])
        else:
            raise KeyError(key)
        self._plans[key] = p
        return p

    def _feed(self, user_input, item_input, recommend, loss, item_sampled, item_sampled_id2idx,
              forward_only):
        m = self.att_emb
        if not recommend:
            if not isinstance(item_input, torch.Tensor) and (self.loss_function not in ('mw', 'mce') or forward_only):
                # host ids: mapped here so that an item without a logit fails like the reference's
                # dict lookup (:173); the plan maps item_id_target again on device (same values)
                self.item_target.feed(m.target_mapping([item_input])[0])
            self.item_id_target.feed(item_input)                              # :176
        update_sampled, _, _ = m.add_input({}, user_input, item_input, item_sampled=item_sampled,
                                           item_sampled_id2idx=item_sampled_id2idx,
                                           forward_only=forward_only, recommend=recommend, loss=loss)
        for op in update_sampled:                                             # :206-207
            op()

    def prepare_next(self, user_input, item_input, item_sampled=None):
        """Announce the batch of the NEXT training step before calling step() for the current one (the host loop
        draws batches before it trains on them: hmf/run_hmf.py:234-242, so a loader can hand step t + 1's ids over
        one step early).  The half of the sparse update that needs the ids only -- contribution keys, their sort,
        the run records (K7, hmf_model.py:146-151) -- then runs for step t + 1 as a side branch of step t's graph,
        off the critical path.  item_sampled: the pool step t + 1 will be given, if it is a new one.  Optional:
        step() without it (or with other ids than announced) computes the same numbers, bit for bit."""
        # (advisor, round 4) identity alone does not see a loader that refills the announced buffers in place: torch
        # tensors are remembered with their version counters, anything else (numpy arrays, lists) is COPIED here
        self._next_batch = tuple(self._announce(x) for x in (user_input, item_input, item_sampled))

    @staticmethod
    def _announce(x):
        import torch
        if x is None or isinstance(x, torch.Tensor):
            return x
        return np.array(x, dtype=np.int32, copy=True)

    @staticmethod
    def _versions(t):
        import torch
        return tuple((x._version if isinstance(x, torch.Tensor) else None) for x in t)

    def _ring_feed(self, plan, item_sampled):
        """step t: queue the announced ids of step t + 1 into the next-step placeholders; True if the plan may run
        in ring mode."""
        nxt, self._next_batch = getattr(self, '_next_batch', None), None
        ann, self._announced = getattr(self, '_announced', None), None
        if nxt is None or not plan.ring_capable():
            return False
        m = self.att_emb
        nodes = {id(m.u_indices['input']): nxt[0], id(self.item_id_target): nxt[1]}
        pool = m.i_indices.get('sampled_pass') if hasattr(m.i_indices, 'get') else None
        known = set(nodes) | ({id(pool)} if pool is not None else set())
        if any(id(n) not in known for n in plan.ring_ids()):
            return False                      # a lookup this method does not know how to announce
        # what step t - 1 sorted ahead is only valid for the ids it was told (tensor identity; a pool given now
        # must be the one announced)
        same = lambda a, b: a is b or (isinstance(a, np.ndarray) and not hasattr(b, '_version')
                                       and np.array_equal(a, np.asarray(b)))
        if ann is None or not same(ann[0], self._cur[0]) or not same(ann[1], self._cur[1]) or \
                (item_sampled is not None and not same(ann[2], item_sampled)) or \
                self._versions(ann) != getattr(self, '_announced_versions', None):
            plan._ring_ready = False        # other ids than announced (or announced buffers refilled since): sort now
        if not plan._ring_ready:
            plan.ring_bootstrap()
        for n in plan.ring_ids():
            if id(n) in nodes:
                n.feed_next(nodes[id(n)])
            elif nxt[2] is not None:
                n.feed_next(nxt[2])
            else:
                n.feed_next(item_sampled if item_sampled is not None else n.value)   # same pool as this step
        self._announced = nxt
        self._announced_versions = self._versions(nxt)
        plan.ring_req = True
        return True

    def step_async(self, session, user_input, item_input, neg_item_input=None, item_sampled=None,
                   item_sampled_id2idx=None, forward_only=False, recommend=False,
                   recommend_new=False, loss=None, run_op=None, run_meta=None):
        """step() without the device->host read of the result: returns the MeanLoss
        node (call .read() for the device scalar) / the top-k index tensor.  recommend with the streaming top-k
        (StreamTopK, fused form): the result is complete only if `self.topk.overflowed()` is False afterwards
        (one device -> host read; step() checks it and re-runs the request on the chunked path)."""
        if loss is None:
            loss = self.loss_function
        self._cur = (user_input, item_input)
        self._feed(user_input, item_input, recommend, loss, item_sampled, item_sampled_id2idx,
                   forward_only)
        if recommend:
            self._plan('recommend').run()
            return self.topk.indices
        if loss == 'warp_eval':
            self._plan('warp_eval').run()
            return [self.batch_loss.value, self.batch_loss.rank_value]
        if forward_only:
            self._plan('eval').run()
            return self.loss_eval
        plan = self._plan('train')
        if getattr(self, '_next_batch', None) is not None:
            self._ring_feed(plan, item_sampled)
        else:
            self._announced = None
        plan.run()
        self.rt.global_step += 1
        return self.loss

    def step(self, session, user_input, item_input, neg_item_input=None, item_sampled=None,
             item_sampled_id2idx=None, forward_only=False, recommend=False, recommend_new=False,
             loss=None, run_op=None, run_meta=None):
        """hmf_model.py:162-228.  Returns: train -> mean loss (float); forward_only ->
        loss_eval (float); recommend -> int32 [mb, top_N]; warp_eval -> [loss, rank]."""
        out = self.step_async(session, user_input, item_input, neg_item_input, item_sampled,
                              item_sampled_id2idx, forward_only, recommend, recommend_new, loss,
                              run_op, run_meta)
        if recommend:
            if isinstance(self.topk, StreamTopK) and self.topk.overflowed():
                # a candidate list of the fused top-k was too short for this batch: once more on the chunked path
                self.topk.fused = False
                self._plans.pop('recommend', None)
                try:
                    out = self.step_async(session, user_input, item_input, neg_item_input, item_sampled,
                                          item_sampled_id2idx, forward_only, recommend, recommend_new, loss, run_op,
                                          run_meta)
                    return out.cpu().numpy()
                finally:
                    self.topk.fused = True
                    self._plans.pop('recommend', None)
            return out.cpu().numpy()
        if isinstance(out, list):
            return [o.cpu().numpy() for o in out]
        return float(out.read().item())

    # ---- batch drawing (hmf_model.py:230-260) ----
    def get_batch(self, data, loss='ce', hist=None):
        batch_user_input, batch_item_input = [], []
        for _ in range(self.batch_size):
            u, i, _t = random.choice(data)
            batch_user_input.append(u)
            batch_item_input.append(i)
        return batch_user_input, batch_item_input, []

    def get_permuted_batch(self, data):
        if self.data_length is None:
            self.data_length = len(data)
            self.start_index = 0
            self.train_permutation = np.random.permutation(self.data_length)
        if self.start_index + self.batch_size >= self.data_length:
            self.start_index = 0
            self.train_permutation = np.random.permutation(self.data_length)
        idx = self.train_permutation[self.start_index:self.start_index + self.batch_size]
        self.start_index += self.batch_size
        users = [data[j][0] for j in idx]
        items = [data[j][1] for j in idx]
        return users, items, None
