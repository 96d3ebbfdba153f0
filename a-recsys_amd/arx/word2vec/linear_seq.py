"""LinearSeq -- shared base of the skip-gram / CBOW recommenders (word2vec/linear_seq.py:11-121)
and the graph both build (word2vec/skipgram_model.py:12-137, cbow_model.py:12-140).

The two models differ only in the item half of the input embedding:
    skip-gram  train: mean([user, item_0])          test: mean([user, mean_i(item_i)])
    CBOW       train: mean([user, mean_i(item_i)])  test: the same
(item_i = mean over the attribute features of the i-th context item; `user` = mean over the
user's features).  With use_sep_item the context items read the 'item' tables and the scorer
the separate 'item_output' tables (embed_attribute.py:96-108).  All n context lookups are ONE
gather launch over the time-major id list; the mean over context positions is a column sum.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import graph as G
from .. import ops
from ..attributes import embed_attribute
from ..attributes.embed_attribute import Dropout
from ..hmf.hmf_model import TopK, _Op, _Var
from ..utils.checkpoint import Saver


class ContextMean(G.Node):
    """x = 0.5 * user + sum_t rows_t, rows = context lookups pre-scaled by 0.5 / n ([n*mb, d])."""

    requires_grad = True

    def __init__(self, rt, items, user, n, mb):
        super().__init__(rt, (mb, items.shape[1]), (items, user))
        self.n, self.mb = n, mb

    def forward(self, train):
        items, user = self.inputs
        x = self.alloc_value()
        d = self.shape[1]
        if self.n == 1:
            ops.axpby(1.0, items.value, 0.0, x)
        else:
            ops.col_sum(items.value.view(self.n, self.mb * d), x.view(-1), self.rt.ws)
        ops.add_rows_bcast(0.5, user.value, 1.0, x)

    def backward(self):
        items, user = self.inputs
        g = self.grad
        if items.requires_grad:
            ops.add_rows_bcast(1.0, g, items.grad_beta(), items.alloc_grad())
        if user.requires_grad:
            ops.add_rows_bcast(0.5, g, user.grad_beta(), user.alloc_grad())


class _FirstRows(G.Node):
    """First mb entries of the time-major context id list (= placeholder 'input0')."""

    def __init__(self, rt, parent, n):
        super().__init__(rt, (n,), (parent,))
        self.value = parent.value[:n]

    def forward(self, train):
        pass


class LinearSeq(object):
    def __init__(self, user_size, item_size, size, batch_size, learning_rate,
                 learning_rate_decay_factor, user_attributes=None, item_attributes=None,
                 item_ind2logit_ind=None, logit_ind2item_ind=None, n_input_items=0,
                 loss_function='ce', logit_size_test=None, dropout=1.0, top_N_items=100,
                 use_sep_item=True, n_sampled=None, output_feat=1, indices_item=None,
                 dtype='float32', params=None, use_graph=True, seed=0, cbow=False):
        self.user_size = user_size
        self.item_size = item_size
        self.top_N_items = top_N_items
        if user_attributes is not None:
            user_attributes.set_model_size(size)
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
        self.loss_function = loss = loss_function
        self.n_input_items = n_input_items
        self.n_sampled = n_sampled
        self.batch_size = mb = batch_size
        self.dropout = dropout
        self.dtype = dtype
        if loss not in ('warp', 'ce', 'bbpr'):
            # 'mw' is accepted by the reference's constructor but `batch_loss_test` is never
            # defined on that branch (skipgram_model.py:113-125): the graph cannot be built
            raise NotImplementedError("loss %r (the reference builds 'warp', 'ce', 'bbpr' here)" % loss)

        self.rt = rt = G.Runtime(learning_rate=learning_rate, use_graph=use_graph)
        self._lr_decay = learning_rate_decay_factor
        self.learning_rate = _Var(lambda: rt.lr_host)
        self.learning_rate_decay_op = _Op(lambda: rt.set_learning_rate(rt.lr_host * self._lr_decay))
        self.global_step = _Var(lambda: rt.global_step)
        self.item_target = G.IdsInput(rt, mb, 'item')               # mapped (logit) target
        self.item_id_target = G.IdsInput(rt, mb, 'item_id')

        n_input = max(n_input_items, 1)
        self._n_input = n_input
        m = embed_attribute.EmbeddingAttribute(user_attributes, item_attributes, mb, n_sampled, n_input,
                                               use_sep_item, item_ind2logit_ind, logit_ind2item_ind,
                                               params=params, runtime=rt, seed=seed)
        self.att_emb = m
        user, _ = m.get_batch_user(1.0, False)                                    # :80
        feats = m._select_feats(m.item_feats, m.item_attributes)
        ctx_all = G.EntityEmbed(rt, m.input_all, feats, with_bias=False, out_scale=0.5 / n_input)
        mean_all = ContextMean(rt, ctx_all, user, n_input, mb)
        if cbow or n_input == 1:
            x_train = mean_all                                                    # cbow_model.py:87-90
        else:
            first = G.EntityEmbed(rt, _FirstRows(rt, m.input_all, mb), feats, with_bias=False, out_scale=0.5)
            x_train = ContextMean(rt, first, user, 1, mb)                         # skipgram_model.py:87
        if float(dropout) != 1.0:
            rt.keep_prob = float(dropout)
            x_train = Dropout(rt, x_train)                                      # :88
        x_test = user if n_input_items == 0 else mean_all                         # :91-99
        logits = m.get_prediction(x_train, output_feat=output_feat)
        logits_test = m.get_prediction(x_test, output_feat=output_feat)
        batch_loss = m.compute_loss(logits, self.item_target, loss)
        batch_loss_test = m.compute_loss(logits_test, self.item_target, loss)
        self.set_mask, self.reset_mask = {}, {}
        if loss in ('warp', 'bbpr'):
            self.set_mask, self.reset_mask = m.get_warp_mask()
        self.loss = G.MeanLoss(rt, batch_loss)
        self.loss_test = G.MeanLoss(rt, batch_loss_test)
        self.output = logits_test
        self.topk = TopK(rt, logits_test, min(top_N_items, self.logit_size))       # :135
        self.indices = self.topk
        self._plans = {}
        self.saver = Saver(self)

    def prepare_warp(self, pos_item_set, pos_item_set_eval):
        self.att_emb.prepare_warp(pos_item_set, pos_item_set_eval)

    def _plan(self, key):
        if key not in self._plans:
            rt, m, loss = self.rt, self.att_emb, self.loss_function
            masks = [m.mask[loss]] if loss in m.mask else []
            if key == 'train':
                self._plans[key] = G.Plan(rt, [self.loss], True, masks)
            elif key == 'eval':
                self._plans[key] = G.Plan(rt, [self.loss_test], False, masks)
            else:
                self._plans[key] = G.Plan(rt, [self.topk], False, [])
        return self._plans[key]

    def step(self, session, user_input, item_input=None, item_output=None, item_sampled=None,
             item_sampled_id2idx=None, forward_only=False, recommend=False, recommend_new=False,
             loss=None, run_op=None, run_meta=None):
        """linear_seq.py:67-121.  item_input: [n_input][mb] context items (time-major);
        item_output: [mb] target items.  Returns the mean loss (train / forward_only) or the
        top-N logit indices [mb, top_N] (recommend)."""
        m = self.att_emb
        if recommend_new:
            raise NotImplementedError("indices_test is never built by the reference (linear_seq.py:98)")
        if not recommend:
            if isinstance(item_output, torch.Tensor):
                self.item_id_target.feed(item_output)
                m.target_mapping_device(self.item_id_target.value, self.item_target.value)
            else:
                self.item_target.feed(m.target_mapping([item_output])[0])          # :76-77
        m.add_input({}, user_input, item_input, neg_item_input=None, item_sampled=item_sampled,
                    item_sampled_id2idx=item_sampled_id2idx, forward_only=forward_only,
                    recommend=recommend, loss=loss or self.loss_function)
        if recommend:
            self._plan('recommend').run()
            return self.topk.indices.cpu().numpy()
        if forward_only:
            self._plan('eval').run()
            return float(self.loss_test.read().item())
        self._plan('train').run()
        self.rt.global_step += 1
        return float(self.loss.read().item())
