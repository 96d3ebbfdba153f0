"""SeqModel -- the reference's LSTM sequence recommender (lstm/seqModel.py:24-604) on MI355X.

Same constructor / step / get_batch signatures.  What differs from the TF graph:
  * every time step is batched: ONE lookup launch gathers all L*mb input items,
    ONE persistent kernel runs the L LSTM steps, ONE GEMM scores all L*mb outputs
    against the (shared) pool, ONE fused kernel evaluates the loss of every step --
    the reference unrolls L copies of each op (seqModel.py:477-493);
  * tf.clip_by_global_norm (seqModel.py:180) is reproduced including TF-1.0's
    aggregation rule (a matmul'd table contributes one dense gradient PER unrolled
    step, IndexedSlices contribute their un-merged values) -- see _clip_hook;
  * evaluating an 'mw' model (`forward_only`) uses the 'warp' loss over the full
    vocabulary, as hmf_model.py:130 does: the reference's own losses_full graph
    for 'mw' (seqModel.py:510) mixes a [mb, n_sampled] mask with [mb, V] logits
    and cannot be built unless n_sampled == V.
Not implemented (raise): beam search (dead code in the reference).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import graph as G
from .. import ops
from ..attributes.embed_attribute import Dropout, ZeroEmbed
from ..hmf.hmf_model import _Op, _Var
from ..utils.checkpoint import Saver
from .batching import SeqBatching


class SeqInputMean(G.Node):
    """seqModel.py:148-156: x_t = reduce_mean([user_embed, reduce_mean(item_embed_t)]).
    `item_half` already holds 0.5*mean_f(item) for all L*mb rows; the user half is
    broadcast-added in place."""

    requires_grad = True

    def __init__(self, rt, item_half, user, L, B):
        super().__init__(rt, item_half.shape, (item_half, user))
        self.L, self.B = L, B
        self._tmp = None

    def forward(self, train):
        item_half, user = self.inputs
        self.value = item_half.value
        if not isinstance(user, ZeroEmbed):
            ops.add_rows_bcast(0.5, user.value, 1.0, self.value)

    def alloc_grad(self):
        self.grad = self.inputs[0].alloc_grad()
        return self.grad

    def grad_beta(self):
        self._grad_written = True
        return self.inputs[0].grad_beta()

    def backward(self):
        item_half, user = self.inputs
        if not user.requires_grad:
            return
        d = self.shape[1]
        if self._tmp is None:
            self._tmp = torch.empty((self.B, d), dtype=torch.float32, device=self.rt.device)
        ops.col_sum(self.grad.view(self.L, self.B * d), self._tmp.view(-1), self.rt.ws)
        ops.add_rows_bcast(0.5, self._tmp, user.grad_beta(), user.alloc_grad())


class InputProject(G.Node):
    """seqModel.py:130-146 (use_concat): x_t = concat_f(user) . w_input_user + concat_f(item_t) . w_input_item.
    The concatenation is never materialised: every feature keeps its own lookup node (width d_f,
    so its gradient rows stay in the arena of its table's width) and contributes
    e_f . W[rows of f]; the weight gradient is written block-row by block-row."""

    requires_grad = True

    def __init__(self, rt, item_feats, user_feats, Wi, Wu, L, B):
        size = Wi.w.shape[1]
        super().__init__(rt, (L * B, size), tuple(item_feats) + tuple(user_feats))
        self.item_feats, self.user_feats = list(item_feats), list(user_feats)
        self.Wi, self.Wu, self.L, self.B, self.size = Wi, Wu, L, B, size
        self._xu = self._dxu = None
        if user_feats:
            self._xu = torch.empty((B, size), dtype=torch.float32, device=rt.device)
            self._dxu = torch.empty((B, size), dtype=torch.float32, device=rt.device)

    @staticmethod
    def _blocks(nodes):
        off = 0
        for n in nodes:
            yield n, off, off + n.shape[1]
            off += n.shape[1]

    def forward(self, train):
        rt = self.rt
        x = self.alloc_value()
        for k, (n, a, b) in enumerate(self._blocks(self.item_feats)):
            ops.gemm(n.value, self.Wi.w[a:b], x, rt.ws, beta=1.0 if k else 0.0)           # :144
        if self.user_feats:
            for k, (n, a, b) in enumerate(self._blocks(self.user_feats)):
                ops.gemm(n.value, self.Wu.w[a:b], self._xu, rt.ws, beta=1.0 if k else 0.0)   # :138
            ops.add_rows_bcast(1.0, self._xu, 1.0, x)                                      # :145

    def backward(self):
        rt = self.rt
        dx = self.grad
        for n, a, b in self._blocks(self.item_feats):
            if n.requires_grad:
                ops.gemm(dx, self.Wi.w[a:b], n.alloc_grad(), rt.ws, transB=True, beta=n.grad_beta())
            ops.gemm(n.value, dx, self.Wi.grad[a:b], rt.ws, transA=True)
        self.Wi.touched = True
        if self.user_feats:
            ops.col_sum(dx.view(self.L, self.B * self.size), self._dxu.view(-1), rt.ws)
            for n, a, b in self._blocks(self.user_feats):
                if n.requires_grad:
                    ops.gemm(self._dxu, self.Wu.w[a:b], n.alloc_grad(), rt.ws, transB=True,
                             beta=n.grad_beta())
                ops.gemm(n.value, self._dxu, self.Wu.grad[a:b], rt.ws, transA=True)
            self.Wu.touched = True


class LSTM(G.Node):
    """seqModel.py:99-103,477: LSTMCell(size) under static_rnn from the zero state."""

    requires_grad = True

    def __init__(self, rt, x, W, b, L, B):
        h = W.w.shape[1] // 4
        super().__init__(rt, (L * B, h), (x,))
        self.W, self.b, self.L, self.B, self.h = W, b, L, B, h
        self.din = x.shape[1]
        dev = rt.device
        self.cs = torch.empty((L * B, h), dtype=torch.float32, device=dev)
        self.gates = torch.empty((L * B, 4 * h), dtype=torch.float32, device=dev)
        self.dz = None
        self._wt = self._dwt = None

    def forward(self, train):
        x = self.inputs[0]
        ops.lstm_fwd(x.value, self.W.w, self.b.w, self.L, self.B, self.din, self.h, 1.0,
                     self.alloc_value(), self.cs, self.gates)

    def backward(self):
        x = self.inputs[0]
        rt, L, B, din, h = self.rt, self.L, self.B, self.din, self.h
        if self.dz is None:
            self.dz = torch.empty((L * B, 4 * h), dtype=torch.float32, device=rt.device)
        dz = self.dz
        if self._wt is None:
            self._wt = torch.empty((4 * h, din), dtype=torch.float32, device=rt.device)
        # round 5: dx on the six-term bf16 tiles with W_x read as it lies (arx_gemm_bt_bx6; no W_x^T needed)
        dx_bx6 = x.requires_grad and ops.gemm_bt_bx6_supported(L * B, din, 4 * h)
        ops.lstm_bwd(self.W.w, self.value, self.cs, self.gates, self.grad, L, B, din, h, dz,
                     wxt=self._wt if (x.requires_grad and not dx_bx6) else None)
        # dx = dz . W_x^T streams dz through the LDS-DMA GEMM with W_x^T as a plain [4h, din] operand (its
        # output tile is at most 128 columns wide; the transpose rode along in the backward kernel).
        # The weight gradient, both halves, and db come from ONE more pass over dz:
        # (dz^T . [x | h_prev])^T with h_prev = the cell outputs one step (B rows) up, written in W's own
        # [din + h, 4h] layout by the split-K reduce (arx_gemm_f32_tn_pair) -- two TN products, two
        # reduces and a transpose before.
        if dx_bx6:
            ops.gemm_bt_bx6(dz, self.W.w[:din], x.alloc_grad(), beta=x.grad_beta())
        elif x.requires_grad:
            ops.gemm(dz, self._wt, x.alloc_grad(), rt.ws, beta=x.grad_beta())
        if L > 1 and ops.gemm_tn_pair_supported(4 * h, din, h, L * B):
            ops.gemm_tn_pair(dz, x.value, self.value, B, self.W.grad, rt.ws, a_rowsum=self.b.grad)
        else:
            if self._dwt is None:
                self._dwt = torch.empty((4 * h, din + h), dtype=torch.float32, device=rt.device)   # [dW_x^T | dW_h^T]
            ops.gemm(dz, x.value, self._dwt[:, :din], rt.ws, transA=True, a_rowsum=self.b.grad)
            if L > 1:
                ops.gemm(dz[B:], self.value[:(L - 1) * B], self._dwt[:, din:], rt.ws, transA=True)
            else:
                self._dwt[:, din:].zero_()                        # (strided view: torch fill)
            ops.transpose(self._dwt, self.W.grad)                 # both halves flipped by one launch
        self.W.touched = self.b.touched = True


class SeqPrediction(G.Prediction):
    """get_prediction applied to every LSTM output (seqModel.py:480-493).  Backward
    keeps the per-step pool gradients (the dense gradient each unrolled matmul
    produces) because clip_by_global_norm needs their separate norms."""

    def __init__(self, rt, latent, pool_embed, L, B):
        super().__init__(rt, latent, pool_embed)
        self.L, self.B = L, B
        self.C_steps = None

    def fusable(self, rows):
        # the fused 'mw' scorer keeps the per-step products when a step's rows are whole 128-row K slices
        return type(self) is SeqPrediction and rows == self.L * self.B and self.B % 128 == 0

    def _steps(self, S, d):
        if self.C_steps is None:
            dev = self.rt.device
            self.C_steps = torch.empty((self.L, S, d), dtype=torch.float32, device=dev)
            self.rs_steps = torch.empty((self.L, S), dtype=torch.float32, device=dev)

    def _backward_bits(self):
        """The fused scorer's backward (graph.Prediction._backward_bits) with the per-time-step products kept:
        arx_mw_scorer_bwd_di slices K = L * B by time step, its partial products ARE C_steps."""
        latent, pool = self.inputs
        S, d = pool.shape
        if latent.requires_grad:
            self.scorer.bwd_dU(latent.alloc_grad(), beta=latent.grad_beta())
        if pool.train_tables:
            self._steps(S, d)
            gp = pool.alloc_grad()
            self.scorer.bwd_dI(gp, db=pool.bias_grad, beta=pool.grad_beta(), step_rows=self.B,
                               dI_steps=self.C_steps, db_steps=self.rs_steps, loss=self.loss_out)
            pool.bias_grad_used = True

    def backward(self):
        if self.fused_into_loss:
            return self._backward_bits()
        latent, pool = self.inputs
        dl = self.grad
        S, d = pool.shape
        if latent.requires_grad:
            ops.gemm(dl, pool.value, latent.alloc_grad(), self.rt.ws, beta=latent.grad_beta())
        if pool.train_tables:
            self._steps(S, d)
            gp = pool.alloc_grad()
            ops.gemm_steps_tn(dl, latent.value, self.C_steps, self.rs_steps, self.L, self.B,
                              C_sum=gp, beta=pool.grad_beta(), rowsum_sum=pool.bias_grad)
            pool.bias_grad_used = True


class SeqWeights(G.Node):
    """seqModel.py:561-567: w_t / (sum_t w_t + 1e-12) per example (time-major rows)."""

    def __init__(self, rt, w, L, B):
        super().__init__(rt, (L * B,), (w,))
        self.L, self.B = L, B
        self.folded_into = None       # a BatchLoss whose fused scorer forms these weights in its first launch

    def forward(self, train):
        if train and self.folded_into is not None:
            self.alloc_value()        # written by arx_mw_scorer_fwd_seqw (graph._bl_forward_gemm_fused)
            return
        ops.seq_weights(self.inputs[0].value, self.L, self.B, self.alloc_value())


class SeqLoss(G.Node):
    """seqModel.py:571-604 sequence_loss(average_across_timesteps, not across batch)."""

    requires_grad = True

    def __init__(self, rt, batch_loss, wn):
        super().__init__(rt, (1,), (batch_loss, wn))
        batch_loss.gscale = 1.0
        batch_loss.row_w = wn
        batch_loss.extra_inputs = (wn,)
        batch_loss.loss_sink = self
        if getattr(batch_loss, 'gemm_fused', False) and isinstance(wn, SeqWeights):
            wn.folded_into = batch_loss

    def forward(self, train):
        bl, wn = self.inputs
        if train and getattr(bl, 'loss_in_scorer', False):
            self.alloc_value()        # sum_r wn_r * loss_r arrives with the scorer's backward (bwd_dI(loss_out=))
            return
        ops.dot_scaled(bl.value, wn.value, 1.0, self.alloc_value())

    def read(self):
        return self.value


class TopKSoftmax(G.Node):
    """seqModel.py:514-517  tf.nn.top_k(tf.nn.softmax(full_logits), topk_n, sorted=True) for every
    time-major row: top-k of the logits (same order as the softmax) + the row logsumexp;
    the softmax values of the k winners are exp(v - lse)."""

    def __init__(self, rt, logits, k):
        super().__init__(rt, (logits.shape[0], k), (logits,))
        self.k = k
        self.indices = torch.empty((logits.shape[0], k), dtype=torch.int32, device=rt.device)
        self.lse = torch.empty((logits.shape[0],), dtype=torch.float32, device=rt.device)

    def forward(self, train):
        x = self.inputs[0].value
        ops.topk(x, self.k, self.alloc_value(), self.indices)
        ops.row_logsumexp(x, self.lse)


class RowsAt(G.Node):
    """value[i] = x[rows[i]]: the LSTM outputs at one time position per sequence (step_recommend)."""

    def __init__(self, rt, x, rows):
        super().__init__(rt, (rows.shape[0], x.shape[1]), (x, rows))

    def forward(self, train):
        x, rows = self.inputs
        ops.gather_onehot(x.value, None, None, rows.value, self.alloc_value())


class _KeepProb(object):
    """model.dropoutRate: .eval() reads, .assign(v) returns an op that sets the keep probability."""

    def __init__(self, rt):
        self.rt = rt

    def eval(self, session=None):
        return self.rt.keep_prob

    def set(self, v):
        self.rt.keep_prob = float(v)

    def assign(self, v):
        return _Op(lambda: self.set(v))


class SeqModel(SeqBatching):
    def __init__(self, buckets, size, num_layers, max_gradient_norm, batch_size, learning_rate,
                 learning_rate_decay_factor, embeddingAttribute, withAdagrad=True, num_samples=512,
                 forward_only=False, dropoutRate=1.0, START_ID=0, loss="ce", devices="",
                 run_options=None, run_metadata=None, use_concat=True, output_feat=1,
                 no_input_item_feature=False, no_user_id=True, topk_n=30, dtype='float32',
                 params=None):
        if num_layers < 1:
            raise ValueError("num_layers must be >= 1")
        if not (0.0 < float(dropoutRate) <= 1.0):
            raise ValueError("dropoutRate (keep probability) must be in (0, 1]")
        if loss not in ('ce', 'warp', 'mw', 'mce'):
            raise NotImplementedError("loss %r" % loss)
        self.embeddingAttribute = m = embeddingAttribute
        m.rt.optimizer = 'adagrad' if withAdagrad else 'sgd'          # seqModel.py:173-176
        self.att_emb = m
        self.rt = rt = m.rt
        self.buckets = list(buckets)
        self.START_ID = START_ID
        self.PAD_ID = START_ID
        self.USER_PAD_ID = 0
        self.batch_size = B = batch_size
        self.loss = loss
        self.devices = devices
        self.output_feat = output_feat
        self.no_input_item_feature = no_input_item_feature
        self.no_user_id = no_user_id
        self.topk_n = topk_n
        self.size = size
        self.max_gradient_norm = float(max_gradient_norm)
        if B % 16 != 0:
            raise NotImplementedError("batch_size must be a multiple of 16 (per-step GEMM tiles)")
        Lmax = self.buckets[-1]
        if m.input_steps < Lmax:
            raise ValueError("EmbeddingAttribute was built with input_steps < longest bucket")

        rt.set_learning_rate(learning_rate)
        self._lr_decay = learning_rate_decay_factor
        self.learning_rate = _Var(lambda: rt.lr_host)
        self.learning_rate_decay_op = _Op(lambda: rt.set_learning_rate(rt.lr_host * self._lr_decay))
        self.global_step = _Var(lambda: rt.global_step)
        # seqModel.py:88-91: keep probability of the DropoutWrappers, switched to 1.0 by the runner
        # around evaluation (lstm/run.py:580,744).  Forward-only plans never drop.
        rt.keep_prob = float(dropoutRate)
        self.dropoutRate = _KeepProb(rt)
        self.dropoutAssign_op = _Op(lambda: self.dropoutRate.set(float(dropoutRate)))
        self.dropout10_op = _Op(lambda: self.dropoutRate.set(1.0))
        self.num_layers = num_layers

        # feeds (seqModel.py:118-124): time-major [L*mb]
        self.target_ids_all = G.IdsInput(rt, Lmax * B, 'target_id_all')
        self.targets_all = G.IdsInput(rt, Lmax * B, 'target_all')
        self.weights_all = G.FloatInput(rt, (Lmax * B,), 'target_weight_all')

        # LSTM weights: TF names of static_rnn(MultiRNNCell([LSTMCell])) variables
        params = params or {}
        din = size
        # MultiRNNCell([cell] * num_layers) (:99-103): every layer maps size -> size with its own
        # weights; initial values under 'lstm_w' / 'lstm_b' (layer 0) and 'lstm_w_<l>' / 'lstm_b_<l>'
        self.Ws, self.bs = [], []
        for l in range(num_layers):
            wname = 'rnn/multi_rnn_cell/cell_%d/lstm_cell/weights' % l
            bname = 'rnn/multi_rnn_cell/cell_%d/lstm_cell/biases' % l
            wk, bk_ = ('lstm_w', 'lstm_b') if l == 0 else ('lstm_w_%d' % l, 'lstm_b_%d' % l)
            if wk in params:
                W = rt.upload(np.asarray(params[wk], dtype=np.float32), torch.float32)
            else:
                W = m._new_var(wname, (din + size, 4 * size), params)
            if bk_ in params:
                b = rt.upload(np.asarray(params[bk_], dtype=np.float32), torch.float32)
            else:
                b = torch.zeros(4 * size, dtype=torch.float32, device=rt.device)    # zero-init biases
            Wp, bp = G.DenseParam(wname, W.contiguous()), G.DenseParam(bname, b.contiguous())
            rt.dense[wname], rt.dense[bname] = Wp, bp
            self.Ws.append(Wp)
            self.bs.append(bp)
        self.W, self.b = self.Ws[0], self.bs[0]

        rt.clip_coef_dev = torch.ones(1, dtype=torch.float32, device=rt.device)
        self._sq = torch.zeros(1, dtype=torch.float32, device=rt.device)
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=rt.device)
        # the "in the pool?" bitmap in front of item2slot pays when every row probes ITS OWN user's
        # positives (HMF); here the L steps of a sequence share one user, the map lines are L2-hot after
        # the first step and the extra dependent load only costs (C4: 690 vs 681 us/step) -- detached
        if getattr(self.att_emb, 'item2slot', None) is not None:
            ops.slot_map_attach_bitmap(self.att_emb.item2slot, None)
        rt.pre_apply_hooks.append(self._clip_hook)

        self.use_concat = bool(use_concat)
        if use_concat:                                                                # :130-146
            ua, ia = m.user_attributes, m.item_attributes
            self._ufeats = m._select_feats(m.user_feats, ua, no_id=no_user_id)
            self._ifeats = m._select_feats(m.item_feats, ia, no_attribute=no_input_item_feature)
            du, di = sum(f.d for f in self._ufeats), sum(f.d for f in self._ifeats)
            self.Wi = G.DenseParam('w_input_item', m._new_var('w_input_item', (di, size), params).contiguous())
            rt.dense['w_input_item'] = self.Wi
            self.Wu = None
            if du:
                self.Wu = G.DenseParam('w_input_user',
                                       m._new_var('w_input_user', (du, size), params).contiguous())
                rt.dense['w_input_user'] = self.Wu
            uid = m.u_indices['input']
            self.user_nodes = [G.EntityEmbed(rt, uid, [f], with_bias=False) for f in self._ufeats]
            if no_user_id and ua.num_features_cat == 1:
                # embed_attribute.py:356-366: the lookup short-circuits to zeros (other user
                # features included), so w_input_user sees a zero input and never moves
                self.user_nodes = []
        else:
            self.user_embed, _ = m.get_batch_user(1.0, concat=False, no_id=no_user_id)   # :148
        self._bk = {}
        self._pool_scale = {}
        self.saver = Saver(self)
        self.gradient_norms = _Var(lambda: float(self._gnorm.item()))

    # ------------------------------------------------------------- graph per bucket
    def _bucket(self, bucket_id):
        if bucket_id in self._bk:
            return self._bk[bucket_id]
        m, rt, B = self.att_emb, self.rt, self.batch_size
        L = self.buckets[bucket_id]
        n = L * B
        Lmax = self.buckets[-1]

        def view(node, name):
            return node if L == Lmax else G.IdsSlice(rt, node, 0, n, name)

        ids_in = view(m.input_all, 'item_input_%d' % L)
        feats = m._select_feats(m.item_feats, m.item_attributes, no_attribute=self.no_input_item_feature)
        if self.use_concat:
            item_nodes = [G.EntityEmbed(rt, ids_in, [f], with_bias=False) for f in feats]   # :142
            x = InputProject(rt, item_nodes, self.user_nodes, self.Wi, self.Wu, L, B)        # :144-145
        else:
            item_half = G.EntityEmbed(rt, ids_in, feats, with_bias=False, out_scale=0.5)   # :150-154
            x = SeqInputMean(rt, item_half, self.user_embed, L, B)                           # :155
        # DropoutWrapper(input_keep_prob) inside every layer, DropoutWrapper(output_keep_prob) on
        # the stack (:100-103); the nodes are the identity while rt.keep_prob == 1
        hs = x
        bk_drop = []
        for l in range(self.num_layers):
            hs = Dropout(rt, hs)
            bk_drop.append(hs)
            hs = LSTM(rt, hs, self.Ws[l], self.bs[l], L, B)                                # :477
        hs = Dropout(rt, hs)
        bk_drop.append(hs)
        wn = SeqWeights(rt, self.weights_all if L == Lmax else _FloatView(rt, self.weights_all, n), L, B)
        tid = view(self.target_ids_all, 'target_id_%d' % L)
        tgt = view(self.targets_all, 'target_%d' % L)
        bk = {'L': L, 'dropouts': bk_drop}
        seq_pred = lambda lat, pe: SeqPrediction(rt, lat, pe, L, B)     # scorer of all L time steps at once
        if self.loss in ('mw', 'mce'):       # ('mce': build-defined sampled softmax, see arx.h)
            logits = m.get_prediction(hs, 'sampled', output_feat=self.output_feat, pred_cls=seq_pred, steps=(L, B))  # :492
            tscore = m.get_target_score(hs, tid)                                            # :493
            bl = m.compute_loss(logits, tscore, self.loss)
        else:
            logits = m.get_prediction(hs, 'full', output_feat=self.output_feat, pred_cls=seq_pred, steps=(L, B))   # :484
            bl = m.compute_loss(logits, tgt, self.loss)
        bk['train'] = SeqLoss(rt, bl, wn)
        bk['train_logits'] = logits
        # losses_full (:510): full-vocabulary loss for evaluation
        if self.loss in ('mw', 'mce'):
            wn2 = SeqWeights(rt, wn.inputs[0], L, B)
            full = m.get_prediction(hs, 'full', output_feat=self.output_feat, steps=(L, B))
            kind_full = 'warp' if self.loss == 'mw' else 'ce'
            import os as _os
            big = n * m.logit_size * 4 > int(_os.environ.get('ARX_STREAM_TOPK_BYTES', str(1 << 30)))
            if big and self.output_feat in (0, 1):     # [L*mb, V] logits streamed, not materialised
                ms = m._mask_state('warp', n) if kind_full == 'warp' else None
                bl_full = G.StreamEvalLoss(rt, kind_full, hs, m._pool_embed('full', self.output_feat), tgt,
                                           mask=ms, mask_rows=B)
                bk['eval_streamed'] = True
            else:
                bl_full = m.compute_loss(full, tgt, kind_full)
            bk['eval'] = SeqLoss(rt, bl_full, wn2)
        else:
            bk['eval'] = bk['train']
            full = logits
        bk['recommend'] = TopKSoftmax(rt, full, min(self.topk_n, full.shape[1]))      # :514-517
        import os as _os2
        if self.output_feat in (0, 1) and n * m.logit_size * 4 > int(_os2.environ.get('ARX_STREAM_TOPK_BYTES',
                                                                                      str(1 << 30))):
            # [L*mb, V] logits are not worth materialising for ONE position per sequence: the rows asked for are
            # gathered first, then the fused full-vocabulary top-k (+ the softmax normaliser) runs on [mb, d]
            from ..hmf.hmf_model import StreamTopK
            bk['rec_rows'] = G.IdsInput(rt, B, 'recommend_rows_%d' % L)
            sel = RowsAt(rt, hs, bk['rec_rows'])
            bk['recommend_stream'] = StreamTopK(rt, sel, m._pool_embed('full', self.output_feat),
                                                min(self.topk_n, m.logit_size), want_lse=True)
        bk['plans'] = {}
        self._bk[bucket_id] = bk
        return bk

    def _plan(self, bucket_id, key):
        bk = self._bucket(bucket_id)
        if key not in bk['plans']:
            m = self.att_emb
            if key == 'train':
                masks = [m.mask[self.loss]] if self.loss in m.mask else []
                bk['plans'][key] = G.Plan(self.rt, [bk['train']], True, masks)
            elif key == 'recommend':
                bk['plans'][key] = G.Plan(self.rt, [bk.get('recommend_stream', bk['recommend'])], False, [])
            else:
                l = 'warp' if self.loss == 'mw' else ('ce' if self.loss == 'mce' else self.loss)
                masks = [m.mask[l]] if (l in m.mask and not bk.get('eval_streamed')) else []
                bk['plans'][key] = G.Plan(self.rt, [bk['eval']], False, masks)
        return bk['plans'][key]

    # ------------------------------------------------ clip_by_global_norm (:180)
    def _row_scale(self, node, for_bias=False, subset=None, tag='all'):
        """Per-row weight of ||grad row||^2 in the IndexedSlices norm of one lookup:
        sum over (a subset of) its features of coef^2 (one-hot) or coef^2/len (bag)."""
        feats = node.feats if subset is None else subset
        key = (id(node), for_bias, tag)
        cache = self.__dict__.setdefault('_rs_cache', {})
        F = len(node.feats)
        coef = (1.0 / F) if for_bias else node.out_scale / F
        static = all(f.kind == 'cat' for f in feats)
        if key in cache and static:
            return cache[key]
        n = node.shape[0]
        dev = self.rt.device
        if key not in cache:
            cache[key] = torch.empty(n, dtype=torch.float32, device=dev)
            cache[(key, 'tmp')] = torch.empty(n, dtype=torch.float32, device=dev)
        out, tmp = cache[key], cache[(key, 'tmp')]
        ncat = sum(1 for f in feats if f.kind == 'cat')
        ops.fill_f32(out, ncat * coef * coef)
        for f in feats:
            if f.kind == 'mulhot':
                ops.inv_len_scale(f.maps[2], node.inputs[0].value, coef * coef, tmp)
                ops.axpby(1.0, tmp, 1.0, out)
        return out

    def _injective(self, f):
        """True when distinct pool items always hit distinct rows of the feature's table (the id
        feature): the per-row merge of the reference's dense matmul gradient is then a no-op."""
        if not hasattr(f, '_inj'):
            if f.kind != 'cat':
                f._inj = False
            elif f.maps[0] is None:
                f._inj = True
            else:
                f._inj = bool(torch.unique(f.maps[0]).numel() == f.maps[0].numel())
        return f._inj

    def _shared_rows_norm(self, n, sp, f, sites_of, sq):
        """Pool feature whose table rows are shared between pool items (multi-hot tokens, a
        categorical attribute): embed_attribute.py:171,188 score the WHOLE table
        (innerp = E . u^T + b) and gather afterwards, so the gradient of each unrolled step is a
        dense [rows, d] matrix in which items sharing a row are already summed."""
        rt = self.rt
        L, S, d = sp.C_steps.shape
        F = len(n.feats)
        cache = self.__dict__.setdefault('_rs_cache', {})
        key = ('shared', id(n), id(f))
        if key not in cache:
            cap = S if f.kind == 'cat' else S * f.max_len
            dev = rt.device
            cache[key] = (torch.empty(cap, dtype=torch.int32, device=dev),
                          torch.empty(cap, dtype=torch.int32, device=dev),
                          torch.empty(cap, dtype=torch.float32, device=dev),
                          torch.zeros(S + 1, dtype=torch.int32, device=dev),
                          torch.zeros(1, dtype=torch.int32, device=dev))
        ks, ss, cs, offs, tot = cache[key]
        ids = n.inputs[0].value
        if f.kind == 'cat':
            ops.sparse_site_onehot(f.maps[0], ids, 0, 1.0 / F, ks, ss, cs)
        else:
            ops.csr_expand(f.maps[0], f.maps[1], f.maps[2], ids, ks.shape[0], rt.ws, pad_token=G.KEY_NONE,
                           pad_seg=0, seg_base=0, coef_scale=1.0 / F, want_coef=True,
                           out=(ks, ss, offs, tot, cs))
        others = [s for s in sites_of.get(id(f.table), [])
                  if s.node is not n and not getattr(s.node, '_is_gmax_vstar', False)]
        others_b = [s for s in others if s.node.with_bias]
        # any IndexedSlices contribution to the variable -> every step's dense gradient is kept
        # apart (un-merged concat); otherwise the steps are add_n'ed first
        X, Lx = (sp.C_steps, L) if others else (n.grad, 1)
        Xb, Lb = (sp.rs_steps, L) if others_b else (n.bias_grad, 1)
        ops.merged_sq_norm(ks, ss, cs, f.table.E.shape[0], sq, rt.ws, scratch=rt.scratch, X=X, d=d, L=Lx, step_stride=S * d,
                           Xb=Xb, Lb=Lb, stepb_stride=S)
        gmax = getattr(n, '_gmax', None)
        if gmax is not None and gmax.vstar._grad_written:
            # output_feat 3: each step's reduce_max sends its residual to ONE element of that step's dense matmul
            # gradient (embed_attribute.py:197) -- a rank-one row on table row v*_t that the merged norm above has not
            # seen (it travels as the gradient of the `vstar` lookup): arx_gmax_norm_corr adds what it changes
            vs = gmax.vstar
            ckey = ('gmaxcorr', id(n))
            if ckey not in cache:
                cache[ckey] = (torch.zeros(gmax.L, dtype=torch.float32, device=rt.device),
                               torch.zeros(1, dtype=torch.float32, device=rt.device))
            corr, tot = cache[ckey]
            ops.gmax_norm_corr(ks, ss, cs, ks.shape[0], X, d, bool(others), S * d, Xb, bool(others_b), S, gmax.vrows,
                               vs.grad, vs.bias_grad if vs.bias_grad_used else None, corr)
            ops.sum_scaled(corr, 1.0, tot)
            ops.axpby(1.0, tot, 1.0, sq)

    def _tiled(self, rs, L, tag, static=False):
        """rs repeated for each of the L unrolled steps; static: rs never changes (one-hot features
        only, see _row_scale) -- tiled once, not in every step."""
        S = rs.shape[0]
        if S % 4 != 0:
            raise NotImplementedError("pool size must be a multiple of 4")
        cache = self._rs_cache
        key = ('tile', tag, L)              # (the pool lookup is shared by buckets of different L)
        fresh = key not in cache
        if fresh:
            cache[key] = torch.empty(L * S, dtype=torch.float32, device=self.rt.device)
        if fresh or not static:
            ops.add_rows_bcast(1.0, rs.view(1, S), 0.0, cache[key].view(L, S))
        return cache[key]

    def _clip_hook(self, plan):
        """tf.clip_by_global_norm over tf.gradients' aggregated list (seqModel.py:179-180).
        TF-1.0 aggregates per variable: all-dense contributions are add_n'ed (norm of the
        sum); if ANY contribution is an IndexedSlices (an embedding_lookup of the same
        variable) everything is concatenated as IndexedSlices and the norm runs over the
        un-merged values -- i.e. each unrolled step's dense matmul gradient separately."""
        rt = self.rt
        sq = self._sq
        zeroed = [False]

        def zero_sq():                    # (only paths that ACCUMULATE onto sq before the last launch need it)
            if not zeroed[0]:
                ops.fill_f32(sq, 0.0)
                zeroed[0] = True
        # Data-parallel replicas (rt.dp): the dense and the pool gradients were all-reduced before this
        # hook -- their norms are the global batch's on every replica; the IndexedSlices of the batch
        # lookups stay un-merged in TF's norm, i.e. their squared norms ADD over the replicas: those
        # are accumulated first and all-reduced as one scalar.
        dp = rt.dp if (rt.dp is not None and rt.dp.world > 1) else None
        norms = []
        local = []                    # indices into norms: batch lookups (additive over replicas)
        deferred = []                 # _shared_rows_norm calls (pool gradients: replicated)
        for p in rt.dense.values():
            if getattr(p, 'touched', False):
                norms.append((p.grad, 1, None, None))
        seq_pools = {}
        for n in plan.order:
            if isinstance(n, SeqPrediction) and n.inputs[1].train_tables and n._grad_written:
                seq_pools[id(n.inputs[1])] = n
        sites_of = {id(t): sites for t, sites, _, _ in plan.tables}
        for n in plan.order:
            if not (isinstance(n, G.EntityEmbed) and n.train_tables and n._grad_written):
                continue
            if getattr(n, '_is_gmax_vstar', False):
                continue                  # (counted with the pool feature it belongs to: _shared_rows_norm)
            if id(n) in seq_pools:
                sp = seq_pools[id(n)]
                L, S, d = sp.C_steps.shape
                shared = [f for f in n.feats if not self._injective(f)]
                for f in shared:
                    if dp is None:
                        zero_sq()
                        self._shared_rows_norm(n, sp, f, sites_of, sq)
                    elif hasattr(dp, 'step_slices'):
                        raise NotImplementedError("striped tables: pool features that share rows between slots "
                                                  "(attribute features) are not served; one-hot id features only")
                    else:
                        deferred.append((n, sp, f))
                for for_bias in (False, True):
                    per_step, merged = [], []
                    for f in n.feats:
                        if f in shared:
                            continue
                        others = [s for s in sites_of.get(id(f.table), [])
                                  if s.node is not n and not getattr(s.node, '_is_gmax_vstar', False)]
                        if for_bias:
                            others = [s for s in others if s.node.with_bias]
                        (per_step if others else merged).append(f)
                    steps_buf = sp.rs_steps if for_bias else sp.C_steps
                    sum_buf = n.bias_grad if for_bias else n.grad
                    dd = 1 if for_bias else d
                    if per_step:
                        rs = self._row_scale(n, True, per_step, 'ps%d' % for_bias)
                        tiled = self._tiled(rs, L, (id(n), for_bias), static=all(f.kind == 'cat' for f in per_step))
                        sl = dp.step_slices(steps_buf) if (dp is not None and hasattr(dp, 'step_slices')) else None
                        if sl is not None:
                            # striped tables (arx.dist.SeqHybridParallel): the per-step gradients were
                            # reduce-scattered -- this rank squares ITS slice of the global sums; the squares add
                            # over the ranks with the batch lookups' (the slice starts on a row: a multiple of dd)
                            mine, lo, cnt = sl
                            if cnt > 0:
                                local.append(len(norms))
                                norms.append((mine[:cnt].view(-1, dd) if dd > 1 else mine[:cnt], dd,
                                              tiled[lo // dd:(lo + cnt) // dd], None))
                        else:
                            norms.append((steps_buf, dd, tiled, None))
                    if merged:
                        rs = self._row_scale(n, True, merged, 'mg%d' % for_bias)
                        norms.append((sum_buf, dd, rs, S * dd))
            else:
                local.append(len(norms))
                norms.append((n.grad, n.shape[1], self._row_scale(n), n.grad.numel()))
                if n.bias_grad_used:
                    local.append(len(norms))
                    norms.append((n.bias_grad, 1, self._row_scale(n, True), None))
        if dp is None:
            # every plain tensor norm of the step + the clip coefficient: one launch (no fill, no clip_coef)
            ops.sq_norm_clip_multi(norms, sq, self.max_gradient_norm, rt.clip_coef_dev, self._gnorm,
                                   init=not zeroed[0], scratch=rt.scratch)
            return
        else:
            zero_sq()
            mine = [norms[i] for i in local]
            if mine:
                ops.sq_norm_accum_multi(mine, sq, scratch=rt.scratch)
            dp.all_reduce_sum(sq)
            for n, sp, f in deferred:
                self._shared_rows_norm(n, sp, f, sites_of, sq)
            rest = [e for i, e in enumerate(norms) if i not in set(local)]
            if rest:
                ops.sq_norm_accum_multi(rest, sq, scratch=rt.scratch)
        ops.clip_coef(sq, self.max_gradient_norm, rt.clip_coef_dev, self._gnorm)

    # ---------------------------------------------------------------------- step
    def _feed(self, user_input, item_inputs, targets, target_weights, L, item_sampled,
              item_sampled_id2idx, forward_only):
        m, B = self.att_emb, self.batch_size

        def flat_i(x):
            if isinstance(x, torch.Tensor):
                return x.reshape(-1)
            return np.asarray(x, dtype=np.int32)[:L].reshape(-1)

        n = L * B
        rt = self.rt

        def put(dst, src, dtype):
            """device tensors are queued (all feeds of the step leave as one copy launch when the
            plan runs); host data is packed into the step's pinned slab and goes up with ONE H2D copy"""
            if isinstance(src, torch.Tensor) and src.is_cuda and src.dtype == dtype and src.is_contiguous():
                rt.queue_feed(src, dst)
            else:
                if not isinstance(src, torch.Tensor):
                    src = torch.from_numpy(np.ascontiguousarray(src))
                if not src.is_cuda and src.dtype == dtype and rt.host_feed(dst, src.numpy()):
                    return                        # host arrays: the step's pinned slab (Runtime.host_feed)
                rt.drop_feed(dst)
                dst.copy_(src.reshape(-1), non_blocking=True)

        put(self.target_ids_all.value[:n], flat_i(targets), torch.int32)
        w = target_weights
        if not isinstance(w, torch.Tensor):
            w = np.asarray(w, dtype=np.float32)[:L].reshape(-1)
        else:
            w = w.reshape(-1)
        put(self.weights_all.value[:n], w, torch.float32)
        put(m.input_all.value[:n], flat_i(item_inputs), torch.int32)
        if self.loss not in ('mw', 'mce') or forward_only:
            m.target_mapping_device(self.target_ids_all.value[:n], self.targets_all.value[:n])
        update_sampled, _, _ = m.add_input({}, user_input, None, item_sampled=item_sampled,
                                           item_sampled_id2idx=item_sampled_id2idx,
                                           forward_only=forward_only, recommend=False, loss=self.loss)
        for op in update_sampled:
            op()

    def step_async(self, session, user_input, item_inputs, targets, target_weights, bucket_id,
                   item_sampled=None, item_sampled_id2idx=None, forward_only=False, recommend=False):
        L = self.buckets[bucket_id]
        self._feed(user_input, item_inputs, targets, target_weights, L, item_sampled,
                   item_sampled_id2idx, forward_only)
        if forward_only:
            self._plan(bucket_id, 'eval').run()
            return self._bucket(bucket_id)['eval']
        self._plan(bucket_id, 'train').run()
        self.rt.global_step += 1
        return self._bucket(bucket_id)['train']

    def step(self, session, user_input, item_inputs, targets, target_weights, bucket_id,
             item_sampled=None, item_sampled_id2idx=None, forward_only=False, recommend=False):
        """seqModel.py:289-324 -> summed sequence loss of the batch (float)."""
        node = self.step_async(session, user_input, item_inputs, targets, target_weights, bucket_id,
                               item_sampled, item_sampled_id2idx, forward_only, recommend)
        return float(node.read().item())

    def step_recommend(self, session, user_input, item_inputs, positions, bucket_id):
        """seqModel.py:326-353 -> [(uid, values[topk_n], indexes[topk_n])]: the top-k softmax
        values / logit indexes at time position positions[i] of sequence i.  Small vocabularies: the full
        [L*mb, V] logits are materialised; past ARX_STREAM_TOPK_BYTES (1 GB) the mb rows asked for are gathered
        and the fused full-vocabulary top-k + log-sum-exp of hmf_model.StreamTopK runs on them (round 5)."""
        L = self.buckets[bucket_id]
        m, B = self.att_emb, self.batch_size
        it = item_inputs
        if not isinstance(it, torch.Tensor):
            it = torch.from_numpy(np.ascontiguousarray(np.asarray(it, dtype=np.int32)[:L].reshape(-1)))
        m.input_all.value[:L * B].copy_(it.reshape(-1), non_blocking=True)
        m.add_input({}, user_input, None, forward_only=True, recommend=True, loss=self.loss)
        bk = self._bucket(bucket_id)
        users = user_input.cpu().numpy() if isinstance(user_input, torch.Tensor) else user_input
        if 'recommend_stream' in bk:
            node = bk['recommend_stream']
            bk['rec_rows'].feed(np.asarray([int(pos) * B + i for i, pos in enumerate(positions)], dtype=np.int32))
            self._plan(bucket_id, 'recommend').run()
            if node.overflowed():             # a candidate list of the fused top-k was too short: the chunked path
                node.fused = False
                bk['plans'].pop('recommend', None)
                try:
                    self._plan(bucket_id, 'recommend').run()
                finally:
                    node.fused = True
                    bk['plans'].pop('recommend', None)
            vals, idx, lse = node.value.cpu().numpy(), node.indices.cpu().numpy(), node.lse.cpu().numpy()
            return [(users[i], np.exp(vals[i] - lse[i]), idx[i]) for i in range(len(positions))]
        self._plan(bucket_id, 'recommend').run()
        node = bk['recommend']
        vals = node.value.cpu().numpy()
        idx = node.indices.cpu().numpy()
        lse = node.lse.cpu().numpy()
        results = []
        for i, pos in enumerate(positions):
            r = int(pos) * B + i
            results.append((users[i], np.exp(vals[r] - lse[r]), idx[r]))
        return results

    # get_batch / get_batch_recommend (:356-452): arx.lstm.batching.SeqBatching


class _FloatView(G.Node):
    def __init__(self, rt, parent, n):
        super().__init__(rt, (n,), (parent,))
        self.value = parent.value[:n]

    def forward(self, train):
        pass
