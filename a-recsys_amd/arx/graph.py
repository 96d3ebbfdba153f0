"""Deferred op graph + executor that replaces the reference's TF-1.0 graph/session.

The reference's classes build a symbolic graph in __init__ and feed it in
step().  Here the same calls build a tiny graph of `Node`s; `Runtime.run`
executes it eagerly as a fixed sequence of libarx.so kernel launches (forward,
backward, optimiser) and -- because every buffer, shape and pointer is static --
captures the whole step into ONE hipGraph that is replayed on later steps
(MI355X: a step at B=64 is launch-bound, not bandwidth-bound).

Only what the hot path needs is implemented; anything else
raises NotImplementedError.  torch tensors are device-memory holders only.
"""
from __future__ import annotations

import os
import weakref

import numpy as np
import torch

from . import ops

KEY_NONE = ops.KEY_NONE


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
_ABL_SKIP_SORT = bool(os.environ.get('ARX_ABL_SKIP_SORT'))
_ABL_SKIP_APPLY = bool(os.environ.get('ARX_ABL_SKIP_APPLY'))

class Table(object):
    """One attribute embedding table [Vf,d] (+ optional bias [Vf]) with its
    Adagrad accumulators (TF initial_accumulator_value = 0.1)."""

    def __init__(self, name, bias_name, E, bias, acc0=0.1):
        self.name = name
        self.bias_name = bias_name
        self.E = E
        self.acc = torch.full_like(E, acc0)
        self.bias = bias
        self.bias_acc = torch.full_like(bias, acc0) if bias is not None else None
        self.sites = []      # SparseSite list (rebuilt per plan)


class DenseParam(object):
    def __init__(self, name, w, acc0=0.1):
        self.name = name
        self.w = w
        self.acc = torch.full_like(w, acc0)
        self.grad = torch.zeros_like(w)


class SparseSite(object):
    """One embedding_lookup call site whose gradient lands in `table`."""

    def __init__(self, table, kind, ids_node, maps, n, max_len, coef, node):
        self.table = table
        self.kind = kind          # 'cat' | 'mulhot'
        self.ids_node = ids_node
        self.maps = maps          # cat: (cat_map,) ; mulhot: (vals, starts, lens)
        self.n = n                # lookups (rows of the gradient source)
        self.cap = n if kind == 'cat' else n * max_len
        self.max_len = max_len
        self.coef = coef
        self.node = node          # EntityEmbed node owning G rows (arena slice)
        self.col_off = 0          # column offset inside the node's grad (concat)
        self.key_off = 0          # offset inside the table's key/src/coef buffers


# --------------------------------------------------------------------------
# nodes
# --------------------------------------------------------------------------
class Node(object):
    requires_grad = False

    def __init__(self, rt, shape, inputs=()):
        self.rt = rt
        self.shape = tuple(int(s) for s in shape)
        self.inputs = tuple(inputs)
        self.value = None
        self.grad = None
        self._grad_written = False
        rt.nodes.append(self)

    # allocation ------------------------------------------------------------
    def alloc_value(self):
        if self.value is None:
            self.value = torch.empty(self.shape, dtype=torch.float32, device=self.rt.device)
        return self.value

    def alloc_grad(self):
        if self.grad is None:
            self.grad = torch.empty(self.shape, dtype=torch.float32, device=self.rt.device)
        return self.grad

    def grad_beta(self):
        """0.0 for the first writer of this node's grad in a backward pass, 1.0 after."""
        b = 1.0 if self._grad_written else 0.0
        self._grad_written = True
        return b

    def forward(self, train):
        raise NotImplementedError

    def backward(self):
        pass

    def get_shape(self):
        return self.shape


class IdsInput(Node):
    """int32 placeholder (embed_attribute.py:142-146 _placeholders)."""

    def __init__(self, rt, n, name):
        super().__init__(rt, (n,))
        self.name = name
        self.value = torch.zeros(n, dtype=torch.int32, device=rt.device)

    def feed(self, arr):
        if isinstance(arr, torch.Tensor):
            src = arr.to(dtype=torch.int32)
        else:
            src = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.int32)))
        if src.numel() != self.value.numel():
            raise ValueError("placeholder %s expects %d ids, got %d" % (self.name, self.value.numel(),
                                                                          src.numel()))
        if src.is_cuda and src.is_contiguous():
            # device-resident batch: queued, all feeds of a step go out as ONE copy launch when the
            # plan runs (Runtime.flush_feeds; eager readers of .value flush first)
            self.rt.queue_feed(src, self.value)
        elif not src.is_cuda and self.rt.host_feed(self.value, src.numpy()):
            pass                                 # host ids: packed into the step's pinned slab (Runtime.host_feed)
        else:
            self.rt.drop_feed(self.value)        # a queued older feed must not overwrite this one
            self.value.copy_(src.reshape(self.value.shape), non_blocking=True)

    def forward(self, train):
        pass

    def next_value(self):
        """Placeholder of the NEXT step's ids (Plan ring mode: that step's K7 sort half runs one step early)."""
        if getattr(self, '_next', None) is None:
            self._next = torch.zeros_like(self.value)
        return self._next

    def feed_next(self, arr):
        dst = self.next_value()
        if isinstance(arr, torch.Tensor):
            src = arr.to(dtype=torch.int32)
        else:
            src = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.int32)))
        if src.numel() != dst.numel():
            raise ValueError("placeholder %s expects %d ids, got %d" % (self.name, dst.numel(), src.numel()))
        if src.is_cuda and src.is_contiguous():
            self.rt.queue_feed(src, dst)
        elif not src.is_cuda and self.rt.host_feed(dst, src.numpy()):
            pass
        else:
            self.rt.drop_feed(dst)
            dst.copy_(src.reshape(dst.shape), non_blocking=True)


class IdsSlice(Node):
    """Rows [start, start + n) of an int placeholder: a bucket shorter than the longest one
    (start = 0), or one time step 'input{t}' of the sequence buffer
    (embed_attribute.py:87-90 keeps one placeholder per step; here they are slices of ONE
    time-major [L*mb] buffer so that the whole sequence can also be looked up by one launch)."""

    def __init__(self, rt, parent, start, n, name):
        super().__init__(rt, (n,), (parent,))
        self.name = name
        self.value = parent.value[start:start + n]

    def feed(self, arr):
        IdsInput.feed(self, arr)

    def next_value(self):
        if getattr(self, '_next', None) is None:
            a = self.value.data_ptr() - self.inputs[0].value.data_ptr()
            start = a // 4
            self._next = self.inputs[0].next_value()[start:start + self.value.numel()]
        return self._next

    def feed_next(self, arr):
        IdsInput.feed_next(self, arr)

    def forward(self, train):
        pass


class FloatInput(Node):
    def __init__(self, rt, shape, name):
        super().__init__(rt, shape)
        self.name = name
        self.value = torch.zeros(self.shape, dtype=torch.float32, device=rt.device)

    def feed(self, arr):
        if isinstance(arr, torch.Tensor):
            src = arr.to(dtype=torch.float32)
        else:
            src = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.float32)))
        if not src.is_cuda and self.value.is_contiguous() and self.rt.host_feed(self.value, src.numpy()):
            return
        self.rt.drop_feed(self.value)
        self.value.copy_(src.reshape(self.value.shape), non_blocking=True)

    def forward(self, train):
        pass


class Feature(object):
    """Static description of one attribute feature of an entity."""

    def __init__(self, kind, table, maps, max_len=1):
        self.kind = kind          # 'cat' | 'mulhot'
        self.table = table
        self.maps = maps          # cat: (cat_map,) ; mulhot: (vals, starts, lens)
        self.max_len = max_len
        self.d = int(table.E.shape[1])


class EntityEmbed(Node):
    """embed_attribute.py:350-417 _get_embedded + the reduce_mean / concat the
    callers apply (:219, :235, :415): value [n, D], optional bias [n].
    mean:   value = out_scale * mean_f e_f(ids)          (D = d)
    concat: value = [e_0 | e_1 | ...]                     (D = sum d_f)"""

    requires_grad = True

    def __init__(self, rt, ids_node, feats, with_bias, concat=False, out_scale=1.0,
                 train_tables=True):
        n = ids_node.shape[0]
        self.feats = feats
        self.concat = concat
        D = sum(f.d for f in feats) if concat else feats[0].d
        if not concat and any(f.d != D for f in feats):
            raise ValueError("mean over features needs equal embedding sizes")
        super().__init__(rt, (n, D), (ids_node,))
        self.with_bias = with_bias
        self.out_scale = float(out_scale)
        if with_bias and (self.out_scale != 1.0 or concat):
            raise NotImplementedError("bias is only produced for mean-combined lookups with out_scale 1")
        self.bias_value = None
        self.bias_grad = None
        self.train_tables = train_tables
        self.bias_grad_used = False

    def alloc_value(self):
        super().alloc_value()
        if self.with_bias and self.bias_value is None:
            self.bias_value = torch.empty(self.shape[0], dtype=torch.float32, device=self.rt.device)
        return self.value

    def forward(self, train):
        out = self.alloc_value()
        ids = self.inputs[0].value
        F = len(self.feats)
        col = 0
        if (F == 2 and not self.concat and self.feats[0].kind == 'cat' and self.feats[1].kind == 'mulhot'
                and self.feats[0].d == self.feats[1].d):
            # id + one multi-hot attribute (HET items): both lookups in one launch
            f0, f1 = self.feats
            wb = self.with_bias
            ops.gather_id_plus_bag(f0.table.E, f0.table.bias if wb else None, f0.maps[0], f1.table.E,
                                   f1.table.bias if wb else None, f1.maps[0], f1.maps[1], f1.maps[2], ids, out,
                                   scale=self.out_scale / F, bias_out=self.bias_value if wb else None)
            return
        for k, f in enumerate(self.feats):
            if self.concat:
                dst, scale, acc = out[:, col:col + f.d], 1.0, False
                col += f.d
            else:
                dst, scale, acc = out, self.out_scale / F, k > 0
            # bias = mean over features of the per-feature bias (:412); it shares the
            # 1/F scale of the mean-combined embedding, so it rides in the same launch.
            bias = f.table.bias if self.with_bias else None
            bout = self.bias_value if self.with_bias else None
            if f.kind == 'cat':
                ops.gather_onehot(f.table.E, bias, f.maps[0], ids, dst, scale=scale,
                                  accumulate=acc, bias_out=bout)
            else:
                ops.gather_mulhot_mean(f.table.E, bias, f.maps[0], f.maps[1], f.maps[2], ids, dst,
                                       scale=scale, accumulate=acc, bias_out=bout)

    def sites(self):
        """Sparse-gradient call sites of this lookup (one per feature)."""
        res = []
        F = len(self.feats)
        col = 0
        for f in self.feats:
            coef = 1.0 if self.concat else self.out_scale / F
            s = SparseSite(f.table, f.kind, self.inputs[0], f.maps, self.shape[0], f.max_len, coef, self)
            s.col_off = col if self.concat else 0
            s.bias_coef = 1.0 / F
            if self.concat:
                col += f.d
            res.append(s)
        return res


class Prediction(Node):
    """embed_attribute.py:148-206 get_prediction in embedding-space form:
    logits[r, j] = latent_r . Ibar_j + bbar_j with (Ibar, bbar) = EntityEmbed of
    the pool's items (mean over output features, output_feat 0/1)."""

    requires_grad = True

    def __init__(self, rt, latent, pool_embed):
        super().__init__(rt, (latent.shape[0], pool_embed.shape[0]), (latent, pool_embed))
        # set by BatchLoss('mw'): in train plans the loss node runs the scorer GEMM itself with the
        # hinge in its epilogue (no [B, S] logits / dlogits in HBM); it leaves the 0/1 activity
        # matrix as bits, the row factors g and g * U here for the two backward products
        self.fused_into_loss = False
        self.scorer = None            # ops.MwScorer (its state holds the act bits, g and the operand planes)
        self.loss_out = None          # (batch_loss, gscale, row_w, out [1]): the step's scalar loss, left by the fused scorer's dI reduce

    def fusable(self, rows):
        """May BatchLoss('mw') take the scorer GEMM in (csrc/scorer.hip)?  Subclasses with their own backward say."""
        return type(self) is Prediction

    def forward(self, train):
        if self.fused_into_loss and train:
            return
        latent, pool = self.inputs
        ops.gemm(latent.value, pool.value, self.alloc_value(), self.rt.ws, transB=True,
                 col_bias=pool.bias_value)

    def _backward_bits(self):
        """dlogits = g_r * act[r, s]: dU += g * (act . Ibar), dIbar = act^T . (g * U), dbbar = act^T . g."""
        latent, pool = self.inputs
        if latent.requires_grad:
            g = latent.alloc_grad()
            self.scorer.bwd_dU(g, beta=latent.grad_beta())
        if pool.train_tables:
            gp = pool.alloc_grad()
            beta = pool.grad_beta()
            hook = getattr(self.rt, '_between_backward_gemms', None)
            if hook is not None:
                hook()
            self.scorer.bwd_dI(gp, db=pool.bias_grad, beta=beta, loss=self.loss_out)
            pool.bias_grad_used = True

    def backward(self):
        if self.fused_into_loss:
            return self._backward_bits()
        latent, pool = self.inputs
        dl = self.grad
        if latent.requires_grad:
            g = latent.alloc_grad()
            ops.gemm(dl, pool.value, g, self.rt.ws, beta=latent.grad_beta())     # dU = dL . Ibar
        if pool.train_tables:
            gp = pool.alloc_grad()
            beta = pool.grad_beta()
            hook = getattr(self.rt, '_between_backward_gemms', None)
            if hook is not None:
                hook()
            # dIbar = dL^T . U ; dbbar = rowsum(dL^T) rides in the same kernel
            ops.gemm(dl, latent.value, gp, self.rt.ws, transA=True, beta=beta,
                     a_rowsum=pool.bias_grad)
            pool.bias_grad_used = True


class BagTokens(Node):
    """Packed bag tokens of a pool's items for ONE multi-hot feature (mulhot_index.py:48-67
    batch_slice2 / batch_segids2 on device): value = token rows int32 [cap] (pads = row 0, their
    scores are never pooled and their gradients are zero), offs int32 [W + 1]."""

    def __init__(self, rt, ids_node, maps, cap, static=False):
        super().__init__(rt, (cap,), (ids_node,))
        self.name = 'bag_tokens'
        self.maps, self.static, self._done = maps, static, False
        W = ids_node.shape[0]
        dev = rt.device
        self.value = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.seg = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.offs = torch.zeros(W + 1, dtype=torch.int32, device=dev)
        self.tot = torch.zeros(1, dtype=torch.int32, device=dev)

    def forward(self, train):
        if self.static and self._done:
            return
        vals, starts, lens = self.maps
        ops.csr_expand(vals, starts, lens, self.inputs[0].value, self.shape[0], self.rt.ws, pad_token=0,
                       pad_seg=0, out=(self.value, self.seg, self.offs, self.tot, None))
        self._done = True


class MeanOf(Node):
    """tf.reduce_mean(innerps, 0) over the per-feature score matrices (embed_attribute.py:205)."""

    requires_grad = True

    def __init__(self, rt, parts):
        super().__init__(rt, parts[0].shape, tuple(parts))

    def forward(self, train):
        out = self.alloc_value()
        F = len(self.inputs)
        for k, p in enumerate(self.inputs):
            ops.axpby(1.0 / F, p.value, 1.0 if k else 0.0, out)

    def backward(self):
        F = len(self.inputs)
        for p in self.inputs:
            if p.requires_grad:
                g = p.alloc_grad()
                ops.axpby(1.0 / F, self.grad, p.grad_beta(), g)


class GlobalMax(Node):
    """score_max = tf.reduce_max(innerp) over the WHOLE feature table's score matrix
    (embed_attribute.py:197): chunked scorer GEMM + running arg-max; nothing of size [B, Vf] is
    kept.  backward: the gradient that flows through the max (collected by SegmentPool) goes to
    its arg-max element -- table row v*, batch row r* -- through `vstar` (a one-row lookup).
    steps = (L, B): the latent holds L unrolled steps of B rows and get_prediction ran once per
    step (lstm/seqModel.py:480-493): one maximum, arg-max and residual PER STEP."""

    requires_grad = True

    def __init__(self, rt, latent, table, chunk=65536, steps=None):
        super().__init__(rt, (1,), (latent,))
        self.table = table
        self.steps = steps
        dev = rt.device
        L, B = steps if steps else (1, latent.shape[0])
        if L * B != latent.shape[0]:
            raise ValueError("GlobalMax: steps do not cover the latent's rows")
        self.L, self.B = L, B
        self.best = torch.zeros(L, dtype=torch.float32, device=dev)
        self.best_idx = torch.zeros((L, 2), dtype=torch.int32, device=dev)      # (row inside the step, table row)
        self.resid = torch.zeros(L, dtype=torch.float32, device=dev)
        self.vrows = torch.zeros(L, dtype=torch.int32, device=dev)              # table rows, contiguous (vstar's ids)
        V = int(table.E.shape[0])
        self.chunk = min(V, int(chunk))
        self._tmp = torch.empty((B, self.chunk), dtype=torch.float32, device=dev)
        self.vstar = None            # EntityEmbed over the arg-max rows, set by the builder

    def forward(self, train):
        lat = self.inputs[0]
        E, b = self.table.E, self.table.bias
        V = int(E.shape[0])
        for t in range(self.L):
            rows = lat.value[t * self.B:(t + 1) * self.B]
            for c0 in range(0, V, self.chunk):
                n = min(self.chunk, V - c0)
                tmp = self._tmp[:, :n]
                ops.gemm(rows, E[c0:c0 + n], tmp, self.rt.ws, transB=True,
                         col_bias=b[c0:c0 + n] if b is not None else None)
                ops.max_argmax(tmp, c0, c0 == 0, self.best[t:t + 1], self.best_idx[t], scratch=self.rt.scratch)
        self.vrows.copy_(self.best_idx[:, 1])

    def backward(self):
        lat, vs = self.inputs[0], self.vstar
        dU = None
        if lat.requires_grad:
            dU = lat.alloc_grad()
            if lat.grad_beta() == 0.0:
                raise RuntimeError("the max residual must not be the first writer of the latent gradient")
        rg = vs.alloc_grad()
        vs.grad_beta()
        vs.bias_grad_used = True
        for t in range(self.L):
            sl = slice(t * self.B, (t + 1) * self.B)
            ops.gmax_residual_bwd(self.resid[t:t + 1], self.best_idx[t], lat.value[sl], vs.value[t:t + 1], rg[t:t + 1],
                                  vs.bias_grad[t:t + 1], dU[sl] if dU is not None else None)


class _IdsOf(Node):
    """int32 view of another node's device index tensor (the arg-max row of GlobalMax)."""

    def __init__(self, rt, parent, tensor, name):
        super().__init__(rt, (tensor.shape[0],), (parent,))
        self.name = name
        self.value = tensor

    def forward(self, train):
        pass


class SegmentPool(Node):
    """embed_attribute.py:194-200: per-bag pooling of the token scores, output_feat 2 (segment_max)
    or 3 (score_max + log(1 + segment_sum(exp(score - score_max))))."""

    requires_grad = True

    def __init__(self, rt, scores, bag, W, mode, gmax=None):
        super().__init__(rt, (scores.shape[0], W), (scores, bag) + ((gmax,) if gmax is not None else ()))
        self.mode, self.W, self.gmax = mode, W, gmax
        self.extra_inputs = (gmax.vstar,) if gmax is not None else ()
        self._resid_rows = None

    def _step_rows(self):
        g = self.gmax
        if g is None or g.L == 1:
            return [(slice(0, self.shape[0]), 0)]
        return [(slice(t * g.B, (t + 1) * g.B), t) for t in range(g.L)]

    def forward(self, train):
        scores, bag = self.inputs[0], self.inputs[1]
        out = self.alloc_value()
        for sl, t in self._step_rows():
            ops.segment_pool_fwd(scores.value[sl], bag.offs, self.W, self.mode, out[sl],
                                 gmax=self.gmax.best[t:t + 1] if self.gmax is not None else None)

    def backward(self):
        scores, bag = self.inputs[0], self.inputs[1]
        ds = scores.alloc_grad()
        if scores.grad_beta() != 0.0:
            raise NotImplementedError("pooled token scores with a second consumer")
        rr = None
        if self.mode == 3:
            if self._resid_rows is None:
                self._resid_rows = torch.empty(self.shape[0], dtype=torch.float32, device=self.rt.device)
            rr = self._resid_rows
        for sl, t in self._step_rows():
            ops.segment_pool_bwd(scores.value[sl], bag.offs, self.W, self.mode, self.value[sl], self.grad[sl], ds[sl],
                                 gmax=self.gmax.best[t:t + 1] if self.gmax is not None else None,
                                 resid_rows=rr[sl] if rr is not None else None)
            if self.mode == 3:
                ops.sum_scaled(rr[sl], 1.0, self.gmax.resid[t:t + 1])
        if self.mode == 3:
            self.gmax._grad_written = True            # its backward applies the residual (runs later)


class TargetScore(Node):
    """embed_attribute.py:208-220 get_target_score."""

    requires_grad = True

    def __init__(self, rt, latent, target_embed):
        super().__init__(rt, (latent.shape[0],), (latent, target_embed))
        self.fused_into_loss = False      # set by BatchLoss('mw'): the loss kernel forms the score
                                          # and both rank-one gradients itself

    def forward(self, train):
        latent, te = self.inputs
        if self.fused_into_loss:
            self.alloc_value()
            return
        ops.dot_score(latent.value, te.value, te.bias_value, self.alloc_value())

    def alloc_grad(self):
        # d(score)/d(bias) = 1: the bias-gradient rows of the target lookup ARE this
        # node's gradient, so the loss kernel writes dt straight into that slice.
        te = self.inputs[1]
        if te.train_tables and te.bias_grad is not None:
            self.grad = te.bias_grad
            return self.grad
        return super().alloc_grad()

    def backward(self):
        latent, te = self.inputs
        if self.fused_into_loss:
            return
        ds = self.grad
        g = latent.alloc_grad()
        dT = te.alloc_grad() if te.train_tables else None
        ops.dot_score_bwd(latent.value, te.value, ds, g, latent.grad_beta() != 0.0, dT)
        if te.train_tables:
            te.grad_beta()
            if ds.data_ptr() != te.bias_grad.data_ptr():
                ops.axpby(1.0, ds, 0.0, te.bias_grad)
            te.bias_grad_used = True


class BatchLoss(Node):
    """embed_attribute.py:525-649 compute_loss ('ce', 'warp', 'mw'); forward and
    backward are one fused kernel, so the downstream reduction tells this node
    its d(total)/d(batch_loss) up front (gscale, row_w)."""

    requires_grad = True

    def __init__(self, rt, kind, logits, target, mask=None, mask_rows=0, loss_func='log', exp_p=1.005):
        if kind not in ('ce', 'warp', 'mw', 'mce', 'warp_eval', 'rs', 'rs-sig', 'rs-sig2', 'bbpr'):
            raise NotImplementedError("loss %r is not implemented on the HIP path" % kind)
        if loss_func not in ops.RS_FUNCS:
            raise ValueError("unknown loss_func %r" % (loss_func,))
        self.loss_func, self.exp_p = loss_func, float(exp_p)
        super().__init__(rt, (logits.shape[0],), (logits, target))
        self.kind = kind
        self.mask = mask              # MaskState or None
        self.mask_rows = mask_rows
        self.gscale = 1.0
        self.row_w = None             # FloatInput-like node
        self.loss_sink = None         # the node that reduces this loss to the step's scalar (MeanLoss / SeqLoss)
        self.loss_in_scorer = False   # this execution: the fused scorer's backward leaves that scalar in the sink
        self.rank_value = None
        # 'mw' over <= 2048 sampled columns with the in-kernel positive mask: the wave that owns
        # a row also forms its target score and the two rank-one gradients (2 launches less)
        self.fuse_ts = False
        self.gemm_fused = False
        if kind in ('mw', 'mce') and isinstance(target, TargetScore) and mask is not None and mask.fused:
            lat, te = target.inputs
            d, W = lat.shape[1], logits.shape[1]
            if W <= 2048 and W % 4 == 0 and d % 4 == 0 and d <= 256 and te.shape[1] == d:
                self.fuse_ts = True
                target.fused_into_loss = True
                # ... and, for the plain pool scorer, the GEMM itself moves in (hinge epilogue)
                B_ = logits.shape[0]
                if (isinstance(logits, Prediction) and logits.fusable(B_) and logits.inputs[0] is lat
                        and (ops.mw_scorer_supported(B_, W, d) if kind == 'mw' else ops.mce_scorer_supported(B_, W, d))):
                    # DEFAULT since round 4 (ARX_SCORER_F32=1: logits GEMM + loss kernel + two f32 GEMMs): no
                    # [B, S] logits / dlogits in HBM, 2 MB of activity bits instead, and all three products on
                    # the bf16 matrix pipe, f32-exact (csrc/gemm_bx6.hip; C3 312 -> 260 us/step in round 3)
                    # 'mce' (round 5): the same move with exp in the epilogue and the weight tiles recomputed in the
                    # backward (ops.MceScorer, d in {64, 128})
                    self.gemm_fused = True
                    logits.fused_into_loss = True

    def forward(self, train):
        logits, target = self.inputs
        bl = self.alloc_value()
        self.loss_in_scorer = False
        in_gemm = self.gemm_fused and train           # no [B, S] logits / dlogits at all
        dl = logits.alloc_grad() if (train and not in_gemm) else None
        if train and not in_gemm:
            logits.grad_beta()
        ms = self.mask
        fused = ms is not None and ms.fused
        m = ms.buf if (ms is not None and not fused) else None
        rw = self.row_w.value if self.row_w is not None else None
        if fused:
            ptr, items = ms.pos_getter()
            uid, i2s = ms.user_ids.value, ms.slot_map_getter()
        if self.gemm_fused and train:
            return self._forward_gemm_fused(bl, rw, uid, ptr, items, i2s)
        if self.gemm_fused and logits.value is None:
            raise RuntimeError("the 'mw' loss of a train-fused scorer cannot run forward-only")
        if self.kind in ('mw', 'mce'):        # sampled pool + separate target score ('mce': build-defined)
            kind = self.kind
            dt = target.alloc_grad() if train else None
            if train:
                target.grad_beta()
            if self.fuse_ts:
                lat, te = target.inputs
                dU = dT = None
                if train:
                    if lat.requires_grad:
                        dU = lat.alloc_grad()
                        if lat.grad_beta() != 0.0:
                            raise RuntimeError("fused target score must be the first writer of the latent gradient")
                    if te.train_tables:
                        dT = te.alloc_grad()
                        te.grad_beta()
                        te.bias_grad_used = True
                        if dt.data_ptr() != te.bias_grad.data_ptr():
                            raise RuntimeError("target-score gradient is expected to alias the bias gradient rows")
                ops.loss_mw_fused_pos(logits.value, lat.value, te.value, te.bias_value, uid, ptr, items, i2s,
                                      bl, dl, target.value, dt, dU, dT, self.gscale, rw, self.mask_rows,
                                      kind=kind)
            elif fused:
                ops.loss_mw_pos(logits.value, target.value, uid, ptr, items, i2s, bl, dl, dt,
                                self.gscale, rw, self.mask_rows, kind=kind)
            else:
                ops.loss_mw(logits.value, target.value, m, bl, dl, dt, self.gscale, rw, self.mask_rows,
                            kind=kind)
        elif self.kind == 'warp':
            if fused:
                ops.loss_warp_pos(logits.value, target.value, uid, ptr, items, i2s, bl, dl,
                                  self.gscale, rw, self.mask_rows)
            else:
                ops.loss_warp(logits.value, target.value, m, bl, dl, self.gscale, rw, self.mask_rows)
        elif self.kind == 'ce':
            ops.loss_ce(logits.value, target.value, bl, dl, self.gscale, rw)
        elif self.kind in ops.RS_KINDS:               # embed_attribute.py:551-603
            ops.loss_rs(logits.value, target.value, self.kind, self.loss_func, self.exp_p, bl, dl,
                        self.gscale, mask=m, pos=(uid, ptr, items, i2s) if fused else None, row_w=rw,
                        mask_rows=self.mask_rows)
        else:  # warp_eval -> [margin_rank, true_rank]
            if self.rank_value is None:
                self.rank_value = torch.empty(self.shape[0], dtype=torch.int32, device=self.rt.device)
            if fused:          # eval-only path: use the array form of the mask
                ms.scatter(0)
            ops.loss_warp_eval(logits.value, target.value, ms.buf if ms is not None else None, bl,
                               self.rank_value, self.mask_rows)
            if fused:
                ms.scatter(1)


def _bl_forward_gemm_fused(self, bl, rw, uid, ptr, items, i2s):
    """scorer GEMM + target score + 'mw' loss fwd/bwd in three launches (arx_mw_scorer_fwd)."""
    logits, target = self.inputs
    lat, te = target.inputs
    pool = logits.inputs[1]
    rt = self.rt
    B, S, d = logits.shape[0], logits.shape[1], lat.shape[1]
    if logits.scorer is None:
        logits.scorer = (ops.MwScorer if self.kind == 'mw' else ops.MceScorer)(B, S, d, rt.device)
    dt = target.alloc_grad()
    target.grad_beta()
    logits._grad_written = True                    # its backward consumes the bits
    dU = dT = None
    if lat.requires_grad:
        dU = lat.alloc_grad()
        if lat.grad_beta() != 0.0:
            raise RuntimeError("fused target score must be the first writer of the latent gradient")
    if te.train_tables:
        dT = te.alloc_grad()
        te.grad_beta()
        te.bias_grad_used = True
        if dt.data_ptr() != te.bias_grad.data_ptr():
            raise RuntimeError("target-score gradient is expected to alias the bias gradient rows")
    # the scalar the step reports, gscale * sum_r row_w_r * loss_r, comes out of the backward's reduce launch
    # (arx_mw_scorer_bwd_di_loss) whenever that launch exists, i.e. the pool trains
    sink = self.loss_sink
    self.loss_in_scorer = sink is not None and bool(pool.train_tables)
    # the sequence model's example weights, normalised over time by the scorer's first launch (SeqWeights.forward
    # left them to it: folded_into)
    seq_w, seq_rows = None, 0
    wn = self.row_w
    if wn is not None and getattr(wn, 'folded_into', None) is self:
        seq_w, seq_rows, rw = wn.inputs[0].value, wn.B, wn.alloc_value()
    logits.loss_out = (bl, self.gscale, rw, sink.alloc_value()) if self.loss_in_scorer else None
    logits.scorer.fwd(lat.value, pool.value, pool.bias_value, te.value, te.bias_value, uid, ptr, items, i2s,
                      bl, target.value, dt, dU, dT, self.gscale, row_w=rw, mask_rows=self.mask_rows,
                      seq_w=seq_w, seq_rows=seq_rows)


BatchLoss._forward_gemm_fused = _bl_forward_gemm_fused


class StreamEvalLoss(Node):
    """Full-vocabulary 'ce' / 'warp' batch loss WITHOUT the [rows, V] logits (forward only:
    hmf_model.py:130,144, seqModel.py:510 -- the evaluation loss of a sampled-loss model): chunked
    scorer GEMM + running per-row reductions (arx_eval_chunk_accum), target logit from the
    target's pool row, positives taken out afterwards (arx_eval_warp_unmask)."""

    def __init__(self, rt, kind, latent, pool, target, mask=None, mask_rows=0, chunk=65536):
        if kind not in ('ce', 'warp'):
            raise NotImplementedError("streaming evaluation loss %r" % kind)
        super().__init__(rt, (latent.shape[0],), (latent, pool, target))
        self.kind, self.mask, self.mask_rows = kind, mask, mask_rows
        self.gscale, self.row_w = 1.0, None
        B, V, dev = latent.shape[0], pool.shape[0], rt.device
        self.chunk = min(int(os.environ.get('ARX_STREAM_EVAL_CHUNK', chunk)), V)
        f32 = torch.float32
        self._buf = torch.empty((B, self.chunk), dtype=f32, device=dev)
        self._t = torch.empty((B,), dtype=f32, device=dev)
        self._T = torch.empty((B, pool.shape[1]), dtype=f32, device=dev)
        self._tb = torch.empty((B,), dtype=f32, device=dev)
        self._a0, self._a1 = torch.empty((B,), dtype=f32, device=dev), torch.empty((B,), dtype=f32, device=dev)
        self.fused = os.environ.get('ARX_EVAL_FUSED', '1') != '0'
        self._parts = None

    def forward(self, train):
        if train:
            raise RuntimeError("StreamEvalLoss is an evaluation node")
        latent, pool, target = self.inputs
        rt, V = self.rt, pool.shape[0]
        mode = 0 if self.kind == 'ce' else 1
        # target logit = latent . pool_row(target) + bias (the target is a logit index = a pool row)
        ops.gather_onehot(pool.value, pool.bias_value, None, target.value, self._T, bias_out=self._tb)
        ops.dot_score(latent.value, self._T, self._tb, self._t)
        if (self.fused and latent.shape[1] in (32, 64, 128) and pool.shape[1] == latent.shape[1]
                and latent.value.stride(0) % 4 == 0 and pool.value.stride(0) % 4 == 0):
            # round 5: ONE pass of the scorer GEMM whose epilogue keeps the per-row sums (arx_gemm_nt_eval_parts)
            # instead of chunks of [rows, 65536] logits and an accumulate launch per chunk
            if self._parts is None:
                npart = ops.gemm_nt_topk_parts(latent.shape[0], V)
                self._parts = torch.empty((latent.shape[0], npart), dtype=torch.float32, device=rt.device)
            if mode == 0:
                ops.gemm_nt_eval_parts(latent.value, pool.value, pool.bias_value, None, self._parts, None)
                ops.row_logsumexp(self._parts, self._a0)
                ops.fill_f32(self._a1, 1.0)                    # (eval_finish: a0 + log(a1) - t)
            else:
                ops.gemm_nt_eval_parts(latent.value, pool.value, pool.bias_value, self._t, None, self._parts)
                ops.row_sum(self._parts, self._a0)
        else:
            for c0 in range(0, V, self.chunk):
                c1 = min(V, c0 + self.chunk)
                lg = self._buf[:, :c1 - c0]
                bias = pool.bias_value[c0:c1] if pool.bias_value is not None else None
                ops.gemm(latent.value, pool.value[c0:c1], lg, rt.ws, transB=True, col_bias=bias)
                ops.eval_chunk_accum(lg, self._t, mode, c0 == 0, self._a0, self._a1)
        ms = self.mask
        if mode == 1 and ms is not None:
            ptr, items = ms.pos_getter()
            ops.eval_warp_unmask(latent.value, pool.value, pool.bias_value, self._t, ms.user_ids.value, ptr, items,
                                 ms.slot_map_getter(), self._a0, mask_rows=self.mask_rows)
        ops.eval_finish(mode, self._a0, self._a1, self._t, self.alloc_value())


class MeanLoss(Node):
    """hmf_model.py:140 tf.reduce_mean(batch_loss)."""

    requires_grad = True

    def __init__(self, rt, batch_loss):
        super().__init__(rt, (1,), (batch_loss,))
        batch_loss.gscale = 1.0 / batch_loss.shape[0]
        batch_loss.loss_sink = self

    lazy = False   # train plans set this: the scalar is reduced only when somebody reads it

    def in_scorer(self):
        """This execution's mean came out of the fused scorer's backward (BatchLoss.loss_in_scorer)."""
        return bool(getattr(self.inputs[0], 'loss_in_scorer', False))

    def forward(self, train):
        if train and self.in_scorer():
            self.alloc_value()
            self._stale = False
            return
        if self.lazy and train:
            self.alloc_value()      # Plan.run marks it stale after every (replayed) step
            return
        self.reduce_now()

    def reduce_now(self):
        bl = self.inputs[0]
        ops.sum_scaled(bl.value, 1.0 / bl.shape[0], self.alloc_value())
        self._stale = False

    def read(self):
        """Device scalar holding the mean loss of the last step."""
        if getattr(self, '_stale', False):
            self.reduce_now()
        return self.value


class MaskState(object):
    """embed_attribute.py:651-672: persistent bool mask [rows, W] (initially all
    True) plus the set / reset scatter ops; positives come from a device CSR."""

    def __init__(self, rt, rows, W, user_ids, slot_map_getter, pos_getter):
        self.rt = rt
        self.rows, self.W = rows, W
        self._buf = None            # allocated on first use: [rows, V] can be GBs at V = 1M
        # fused: the loss kernel derives the mask bits itself (LDS, <= 2^20 columns), so
        # the persistent array and its set/reset launches are not needed by the plans.
        self.fused = W <= ops.POS_MASK_MAX_COLS
        self.user_ids = user_ids
        self.slot_map_getter = slot_map_getter
        self.pos_getter = pos_getter

    @property
    def buf(self):
        if self._buf is None:
            self._buf = torch.ones((self.rows, self.W), dtype=torch.uint8, device=self.rt.device)
        return self._buf

    def scatter(self, value):
        ptr, items = self.pos_getter()
        ops.pos_mask_scatter(self.user_ids.value, ptr, items, self.slot_map_getter(), self.buf, value)


# --------------------------------------------------------------------------
# runtime
# --------------------------------------------------------------------------
class Plan(object):
    """A fixed kernel sequence (forward [+ backward + optimiser]) over the graph."""

    def __init__(self, rt, fetch, train, masks=()):
        self.rt = rt
        self.fetch = fetch
        self.train = train
        self.masks = list(masks)
        self.order = rt.topo(fetch)
        self.graph = None
        self.warm = 0
        self._kp = rt.keep_prob
        self._pregather = None
        self._early_jobs, self._k7_early, self._k7_stream, self._k7_done_keys = [], None, None, None
        self._k7_fork = None
        self._jobs, self._n_passes = [], 0
        # ring mode (prepare_next): the K7 sort half of step t + 1 runs as the side branch of step t's graph
        self.ring_req = False          # set by the model for the coming run()
        self._ring = False             # inside a ring execution / capture
        self._ring_par = 0             # slot the coming step APPLIES from; its branch sorts into 1 - _ring_par
        self._ring_ready = False       # slot _ring_par holds the sorted lookups of the coming step
        self._ring_graphs = {}         # parity -> captured executable
        self._ring_warm = {}
        self.has_dropout = any(getattr(n, 'uses_dropout', False) for n in self.order)
        self.tables = []
        self.arenas = []
        self._bindings = []      # (node, grad, bias_grad, arena, arena_b, row0): see _bind
        if train:
            self._plan_sparse()

    # ---- sparse-gradient bookkeeping (static) ----
    def _plan_sparse(self):
        rt = self.rt
        embeds = [n for n in self.order if isinstance(n, EntityEmbed) and n.train_tables]
        # gradient arena: one buffer per row width so rows are contiguous
        by_width = {}
        for n in embeds:
            by_width.setdefault(n.shape[1], []).append(n)
        for D, nodes in by_width.items():
            rows = sum(n.shape[0] for n in nodes)
            arena = torch.zeros((rows, D), dtype=torch.float32, device=rt.device)
            arena_b = torch.zeros((rows,), dtype=torch.float32, device=rt.device)
            r0 = 0
            for n in nodes:
                self._bindings.append((n, arena[r0:r0 + n.shape[0]], arena_b[r0:r0 + n.shape[0]], arena,
                                       arena_b, r0))
                r0 += n.shape[0]
            self.arenas.append((arena, arena_b))
        self._bind()
        tables = {}
        for n in embeds:
            for s in n.sites():
                tables.setdefault(id(s.table), (s.table, []))[1].append(s)
        for table, sites in tables.values():
            # all sites of a table must share one arena (same width) -- true for
            # mean-combined features; concat slices keep the node's width too.
            total = 0
            for s in sites:
                s.key_off = total
                total += s.cap
            bufs = {
                'keys': torch.full((total,), KEY_NONE, dtype=torch.int32, device=rt.device),
                'src': torch.zeros((total,), dtype=torch.int32, device=rt.device),
                'coef': torch.zeros((total,), dtype=torch.float32, device=rt.device),
                'hot': torch.zeros((total // 16 + 4,), dtype=torch.int32, device=rt.device),
            }
            widths = set(s.node.shape[1] for s in sites)
            if len(widths) != 1:
                raise NotImplementedError("a table used by lookups of different output widths")
            self.tables.append((table, sites, bufs, total))

    def _bind(self):
        """Point the lookup nodes at THIS plan's gradient arena.  Lookup nodes can be shared between
        plans (SeqModel: the cached pool lookup and the user lookups serve every bucket's train
        plan, each with an arena sized for its own L), so the slices are plan state that is
        re-attached before every execution / capture, never node state."""
        for n, g, gb, arena, arena_b, r0 in self._bindings:
            n.grad, n.bias_grad = g, gb
            n.arena, n.arena_b, n.row0 = arena, arena_b, r0

    # ---- execution ----
    def _execute(self):
        rt = self.rt
        self._bind()
        if self.train and self.has_dropout and rt.keep_prob < 1.0:
            ops.counter_add(rt.step_dev, 1)      # fresh dropout masks on every (replayed) step
        for m in self.masks:
            if not m.fused:
                m.scatter(0)                               # set_mask (hmf_model.py:209-210)
        for n in self.order:
            n._grad_written = False
            if isinstance(n, EntityEmbed):
                n.bias_grad_used = False
        # Row-striped tables (arx.dist.SeqHybridParallel): the lookups of this step are exchanges -- ids to the owners
        # of the rows, rows back -- done by the wrapper; the nodes it served are skipped below
        fetched = rt.dp.fetch(self) if (rt.dp is not None and hasattr(rt.dp, 'fetch')) else ()
        # single one-hot lookups are independent leaves: all of them leave in one launch
        if self._pregather is None:
            self._pregather = []
            groups = {}
            for n in self.order:
                kinds = tuple(f.kind for f in n.feats) if isinstance(n, EntityEmbed) else ()
                if id(n) in fetched:
                    continue
                if (kinds in (('cat',), ('mulhot',), ('cat', 'mulhot')) and not n.concat
                        and all(f.d == n.feats[0].d for f in n.feats)
                        and type(n.inputs[0]).__name__ in ('IdsInput', 'IdsSlice')
                        and n.inputs[0].value.dtype == torch.int32):
                    groups.setdefault(n.shape[1], []).append(n)
            for d_, nodes in groups.items():
                for k in range(0, len(nodes), 8):
                    grp = nodes[k:k + 8]
                    if len(grp) < 2:
                        continue
                    for n in grp:
                        n.alloc_value()
                    if all(len(n.feats) == 1 and n.feats[0].kind == 'cat' for n in grp):
                        sites = [(n.feats[0].table.E, n.feats[0].table.bias if n.with_bias else None,
                                  n.feats[0].maps[0], n.inputs[0].value, n.value, n.out_scale,
                                  n.bias_value if n.with_bias else None) for n in grp]
                        self._pregather.append((ops.GatherSet(sites), grp))
                        continue
                    # one-hot and / or multi-hot features per lookup (users, HET / MIX items): arx_lookup_multi
                    sites = []
                    for n in grp:
                        fc = next((f for f in n.feats if f.kind == 'cat'), None)
                        fm = next((f for f in n.feats if f.kind == 'mulhot'), None)
                        wb = n.with_bias
                        sites.append((fc.table.E if fc else None, fc.table.bias if (fc and wb) else None,
                                      fc.maps[0] if fc else None, fm.table.E if fm else None,
                                      fm.table.bias if (fm and wb) else None,
                                      fm.maps[0] if fm else None, fm.maps[1] if fm else None,
                                      fm.maps[2] if fm else None, n.inputs[0].value, n.value,
                                      n.out_scale / len(n.feats), n.bias_value if wb else None))
                    self._pregather.append((ops.LookupSet(sites), grp))
        if self.train:
            self._early_sort()
        pre = set()
        for gs, grp in self._pregather:
            if isinstance(gs, ops.LookupSet):
                ops.lookup_multi(gs)
            else:
                ops.gather_onehot_multi(gs)
            pre.update(id(n) for n in grp)
        # where the sort branch is issued decides which chain the graph gives the first queue to.
        # Measured (B=16384): with the bag table riding on the one-hot pass (C3 HET, ~100 us of
        # sorts) the branch issued BEFORE the lookups ran ahead of them -- the first gather started
        # ~50 us after the feed: 367 us/step; issued after the first lookup: 352 us.  With a
        # two-stage bag pass of its own (C3-MIX, ~135 us of sorts) the early start wins (414 vs 423
        # us); one-hot only (C2): no difference.
        rider = any(j[0] == 'multi' and j[1][1] is not None for j in self._early_jobs)
        early_after = (1 if rider else 0) if self.train else -1
        if early_after == 0:
            self._early_launch()
        k_fwd = 0
        pre.update(fetched)
        for n in self.order:
            if id(n) in pre:
                continue
            n.forward(self.train)
            k_fwd += 1
            if k_fwd == early_after:
                self._early_launch()
        if self.train:
            self._early_launch()          # (no-op when already issued)
        if self.train:
            # The one-hot sort of the K7 branch is long done when the backward GEMMs start: the main
            # chain takes that dependency THERE (between dU and dI), where it costs nothing, instead
            # of in front of the one-hot apply, where a two-parent node starts ~10 us late (measured)
            self._mid_waited = False
            # (measured: with a VIRTUAL entity table -- MIX -- 350 -> 340 us/step; with a real id table -- HET
            # -- 313 -> 315, so there the dependency stays at the apply)
            mid_at = 'dI' if getattr(self, '_k7_virtual', False) else 'apply'
            if (self._k7_early is not None and self._k7_early[2] is not None and self._k7_early[2] != 'ring'
                    and mid_at == 'dI'):
                def _mid_wait(ev=self._k7_early[2]):
                    if not self._mid_waited:
                        torch.cuda.current_stream().wait_event(ev)
                        self._mid_waited = True
                rt._between_backward_gemms = _mid_wait
            try:
                for n in reversed(self.order):
                    if n.requires_grad and n._grad_written:
                        n.backward()
            finally:
                rt._between_backward_gemms = None     # (never leave a closure over this step's event behind)
            dp = rt.dp
            if dp is not None:
                # data-parallel replicas (arx.dist.SeqDataParallel): dense and pool gradients are
                # all-reduced, the rows of the batch lookups gathered, so that every replica applies
                # the update of the GLOBAL batch
                dp.exchange(self)
            rt.pre_apply(self)
            if dp is not None:
                local_tables, self.tables = self.tables, dp.gathered_tables(self)
                try:
                    self._apply_sparse()
                finally:
                    self.tables = local_tables
            else:
                self._apply_sparse()
            rt.apply_dense(self)
        for m in self.masks:
            if not m.fused:
                m.scatter(1)                               # reset_mask (:217-218)

    def _fusable(self, entry):
        """(one-hot sites, multi-hot sites) of a table that can join the fused K7 pass."""
        rt = self.rt
        table, sites, bufs, total = entry
        live = [s for s in sites if s.node._grad_written]
        if not live or rt.force_sort_path or rt.cat_mode != 0:
            return None
        if any(s.col_off != 0 for s in live):
            return None
        if self._bags_ok(live):
            return None                   # multi-hot table with its own two-stage pass (_bag_pass)
        return [s for s in live if s.kind == 'cat'], [s for s in live if s.kind != 'cat']

    def _bags_ok(self, live):
        """Multi-hot lookups of one table take the two-stage pass (merge per entity, then per token:
        arx_sparse_adagrad_bags) once their padded expansion is past the one-launch rank sort --
        below that the step is launch-bound and the contribution-level chain has fewer launches."""
        rt = self.rt
        if rt.no_bags or rt.force_sort_path or not live or len(live) > 8:
            return False
        s0 = live[0]
        return (all(s.kind == 'mulhot' and s.col_off == 0 and s.maps[0] is s0.maps[0]
                    and s.node.arena is s0.node.arena for s in live)
                and sum(s.cap for s in live) > 8192 and sum(s.cap for s in live) < (1 << 31) - 1)

    def _apply_sparse(self):
        """One K7 pass per table -- except that tables of equal width that read the same
        gradient arena share ONE pass (arx_sparse_adagrad_cat_multi: the sort key carries the
        table index; multi-hot lookups join as pre-expanded segments): the kernel chains are
        launch-bound, and parallel hipGraph branches did not overlap them (measured)."""
        rt = self.rt
        fused = []
        if not rt.no_multi:
            cand = []
            for entry in self.tables:
                f = self._fusable(entry)
                if f is not None:
                    cand.append((entry, f[0], f[1]))
            virt = None
            if len(cand) == 1 and not cand[0][2]:
                virt = self._find_rider([cand[0]], {id(cand[0][0])})     # a bag table that can ride on it?
            if len(cand) >= 2 or virt is not None:
                e0, c0, m0 = cand[0]
                n0 = (c0 + m0)[0].node
                d0 = e0[0].E.shape[1]
                group = [(e, c, m) for e, c, m in cand
                         if e[0].E.shape[1] == d0 and all(x.node.arena is n0.arena for x in c + m)
                         and all(x.node.arena_b is n0.arena_b for x in c + m)]
                # multi-hot segments are padded to their worst-case capacity: past ~0.5 M entries
                # the shared sort/apply pays more for the pads than the saved launches (measured:
                # C3 B=16384 fused 627 us vs 543 us) -- then only the one-hot tables share a pass
                if sum(x.cap for _, c, m in group for x in c + m) > (1 << 19):
                    group = [(e, c, m) for e, c, m in group if not m]
                rows_bits = max(int(e[0].E.shape[0] - 1).bit_length() for e, _, _ in group) if group else 0
                n_tot = sum(x.cap for _, c, m in group for x in c + m)
                # below the LDS rank-sort limit a table's own chain is already 3 launches; fusing
                # would push the union into the 8-launch radix path (measured slower at B=4096)
                big = any(sum(x.cap for x in c + m) > 8192 for _, c, m in group)
                # ... unless the UNION still fits the rank sort: then one 3-launch chain serves all
                # tables (C1, B=64: two chains of 3 launches -> one)
                if virt is not None and len(group) == 1:
                    fused = group            # one one-hot table + the entity ids of a bag table (below)
                elif ((big or (n_tot <= 8192 and not any(m for _, _, m in group))) and 2 <= len(group) <= 4 and sum(len(c) for _, c, _ in group) <= 8
                        and sum(len(m) for _, _, m in group) <= 8 and rows_bits + 2 <= 30
                        and n_tot <= (1 << 22)):
                    fused = group
        early = self._k7_early
        self._k7_early = None
        self._k7_done_keys = None
        split_join = None
        ring_done = None
        if early is not None and early[2] == 'ring':
            ring_done = early[1]
            self._k7_done_keys = early[0]           # this step's sort half ran one step ago (slot _ring_par)
        elif early is not None:
            if early[2] is not None:
                if not getattr(self, '_mid_waited', False):
                    torch.cuda.current_stream().wait_event(early[2])  # the one-hot sort is done ...
                split_join = early[1]                                 # ... the token chain is joined before its apply
            else:
                torch.cuda.current_stream().wait_event(early[1])      # join the sort branch
            self._k7_done_keys = early[0]
        self._jobs, self._n_passes = [], 0
        done = set(id(e) for e, _, _ in fused)
        rider = self._find_rider(fused, done) if fused else None
        if rider is not None:
            # a multi-hot table looked up with exactly the lookups of one fused one-hot table (HET: an
            # item's id row and its bag): it rides on the fused pass -- that table goes first (its
            # keys sort first), the bag needs no entity sort / merge pass of its own.  Without such a
            # table (MIX: the id token lives in the bag) the entity ids join the pass as a VIRTUAL
            # table 0: sorted with the others, runs merged, no rows updated.
            gi, bag_entry, bag_live = rider
            if gi is not None:
                fused = [fused[gi]] + fused[:gi] + fused[gi + 1:]
            done.add(id(bag_entry))
        elif len(fused) == 1:
            fused = []                       # (a lone one-hot table keeps its own pass)
            done = set()
        if fused:
            key = self._multi_key(fused)
            bag = None
            if rider is not None:
                use_bias = bag_entry[0].bias is not None and any(x.node.bias_grad_used for x in bag_live)
                key = key + ('rider', id(bag_entry), use_bias, gi is None)
                bag = (bag_entry, bag_live, use_bias, gi is None)
            phase = 2 if (self._k7_done_keys is not None and key in self._k7_done_keys) else 3
            if self._ring and phase != 2:
                raise RuntimeError("ring mode: the step's K7 pass differs from the one sorted ahead")
            if phase == 2 and ring_done is not None and bag is not None:
                self._apply_multi(fused, phase=7, key=key, bag=bag)
                self._apply_multi(fused, phase=8, key=key, bag=bag)
            elif phase == 2 and split_join is not None and bag is not None:
                self._apply_multi(fused, phase=7, key=key, bag=bag)
                torch.cuda.current_stream().wait_event(split_join)
                split_join = None
                self._apply_multi(fused, phase=8, key=key, bag=bag)
            else:
                if split_join is not None:
                    torch.cuda.current_stream().wait_event(split_join)
                    split_join = None
                self._apply_multi(fused, phase=phase, key=key, bag=bag)
            # size estimate for _plan_early: one-hot sites are all live, a multi-hot site's padded
            # capacity (max_len slots per bag) holds about a third of that in live tokens
            job = ('multi', (fused, bag), key, sum(x.cap for _, c, _m in fused for x in c)
                   + sum(x.cap for _, _c, m in fused for x in m) // 3)
            if bag is not None:
                job = job + (sum(x.cap for x in bag_live) // 6,)
            self._jobs.append(job)
            self._n_passes += 1
        if split_join is not None:
            torch.cuda.current_stream().wait_event(split_join)
        for entry in self.tables:            # the tables that are not part of the fused pass: one pass each
            if id(entry) not in done:
                self._apply_one(entry)
        if ring_done is not None:
            torch.cuda.current_stream().wait_event(ring_done)      # rejoin the branch (next step's sorts)
        self._plan_early(self._jobs, self._n_passes)

    def _find_rider(self, fused, done):
        """(index in the fused group, table entry, live sites) of a two-stage multi-hot table whose
        lookups are exactly the lookups of one fused one-hot table: same id tensors, gradient rows,
        coefficients and arena (arx_sparse_adagrad_cat_multi_bags), or None.  Index None: no such
        table, but the bag table's lookups read the group's arena and fit the pass as a virtual
        table (its entity ids only)."""
        rt = self.rt
        if rt.no_rider or any(m for _, _, m in fused):
            return None
        virtual = None
        n0 = fused[0][1][0].node
        for entry in self.tables:
            if id(entry) in done:
                continue
            live = [x for x in entry[1] if x.node._grad_written]
            if not self._bags_ok(live):
                continue
            for gi, (e, c, _m) in enumerate(fused):
                if len(c) != len(live) or e[0].E.shape[1] != entry[0].E.shape[1]:
                    continue
                if any(x.maps[0] is not c[0].maps[0] for x in c):
                    continue
                if self._rider_csr(c[0].maps[0], live[0].maps[1], live[0].maps[2], int(e[0].E.shape[0])) is None:
                    continue                    # (entity -> id-table row must be one-to-one)
                if all(a.ids_node.value.data_ptr() == b.ids_node.value.data_ptr()
                       and a.ids_node.value.shape == b.ids_node.value.shape and a.node.row0 == b.node.row0
                       and a.coef == b.coef and a.node.arena is b.node.arena and a.node.arena_b is b.node.arena_b
                       for a, b in zip(c, live)):
                    return gi, entry, live
            if (virtual is None and len(fused) <= 3 and sum(len(c) for _, c, _ in fused) + len(live) <= 8
                    and entry[0].E.shape[1] == fused[0][0][0].E.shape[1]
                    and all(x.node.arena is n0.arena and x.node.arena_b is n0.arena_b for x in live)
                    and int(live[0].maps[2].shape[0]).bit_length() + 2 <= 30 and not rt.no_virtual):
                virtual = (None, entry, live)
        return virtual

    def _rider_csr(self, m, starts, lens, rows):
        """The bag index (starts, lens) of a multi-hot feature re-indexed by the ROW of the one-hot
        table whose lookups it shares (m: entity -> row, None = identity): the sorted keys of the
        fused pass are rows.  None unless m is one-to-one.  Built once per (map, bag index)."""
        cache = self.__dict__.setdefault('_rider_cache', {})
        k = (0 if m is None else m.data_ptr(), starts.data_ptr(), lens.data_ptr(), rows)
        if k not in cache:
            out = None
            n = int(lens.shape[0]) if m is None else min(int(m.shape[0]), int(lens.shape[0]))
            if m is None:
                out = (starts, lens) if rows <= n else None
            elif n > 0:
                mm = m[:n].long()
                if int(mm.min()) >= 0 and int(mm.max()) < rows and int(torch.bincount(mm, minlength=rows).max()) <= 1:
                    st_r = torch.zeros(rows, dtype=torch.int32, device=m.device)
                    ln_r = torch.zeros(rows, dtype=torch.int32, device=m.device)
                    st_r[mm] = starts[:n]
                    ln_r[mm] = lens[:n]
                    out = (st_r, ln_r)
            cache[k] = out
        return cache[k]

    def _multi_key(self, group):
        return tuple(id(x) for _, c, m in group for x in c + m) + tuple(
            bool(e[0].bias is not None and any(x.node.bias_grad_used for x in c + m)) for e, c, m in group)

    def _apply_multi(self, group, phase=3, key=None, bag=None, slot=None):
        """phase 1: contributions + sort (ids only), 2: apply, 3: both -- see _early_sort.
        bag: (table entry, live sites, use_bias) of a multi-hot table riding on table 0 of the group.
        slot (ring mode): one of two buffer sets; its sort half reads the NEXT step's id placeholders."""
        rt = self.rt
        if key is None:
            key = self._multi_key(group)
        if slot is None and self._ring:
            if phase in (1, 5, 6):
                slot = 1 - self._ring_par        # the branch: the NEXT step's sort half
            elif phase in (2, 7, 8):
                slot = self._ring_par            # this step's apply, from what the previous step's branch left
            else:
                raise RuntimeError("ring mode runs the two halves of a K7 pass in different steps (phase %d)" % phase)
        ids_of = (lambda x: x.ids_node.value) if slot is None else (lambda x: x.ids_node.next_value())
        cache = self.__dict__.setdefault('_multi_cache', {})
        ck = key if slot is None else (key, 'slot', slot)
        ent = cache.get(ck)
        if ent is None:
            tables, sites, extra, xsites = [], [], [], []
            t_off = 0
            if bag is not None and bag[3]:
                # virtual table 0: the bag table's entity lookups (key = entity id)
                n_ent = int(bag[1][0].maps[2].shape[0])
                vmap = torch.zeros(n_ent, dtype=torch.int32, device=rt.device)    # per-entity map of the grouped K7 path
                tables.append((None, None, None, None, vmap, n_ent))
                for x in bag[1]:
                    sites.append((0, None, ids_of(x), x.node.row0, x.coef))
                t_off = 1
            # which tables update their bias is part of the pass key (_multi_key, taken at apply time): the
            # nodes' bias_grad_used flags are only set once the step's backward ran -- a sort branch that builds
            # this entry runs before it
            n_ids = sum(len(c) + len(m) for _, c, m in group)
            key_bias = key[n_ids:n_ids + len(group)]
            for ti, (e, c, m) in enumerate(group, start=t_off):
                table = e[0]
                use_bias = bool(key_bias[ti - t_off])
                sgd = rt.optimizer == 'sgd'          # no slots: the kernels do plain gradient descent
                tables.append((table.E, None if sgd else table.acc, table.bias if use_bias else None,
                               table.bias_acc if (use_bias and not sgd) else None, self._aux_cnt(table)))
                for x in c:
                    sites.append((ti, x.maps[0], ids_of(x), x.node.row0, x.coef))
                for x in m:
                    extra.append((ti, x.cap))
                    xsites.append(x)
            args = ops.MultiCatArgs(tables, sites, extra)
            n = args.total
            dev = rt.device
            ent = dict(args=args, xsites=xsites,
                       keys=torch.empty(n, dtype=torch.int32, device=dev),
                       src=torch.empty(n, dtype=torch.int32, device=dev),
                       coef=torch.empty(n, dtype=torch.float32, device=dev),
                       any_bias=any(t[2] is not None for t in tables),
                       ws=ops.Workspace(dev))       # own workspace: the sorted arrays live in it
            cache[ck] = ent                         # between the two phases
        args = ent['args']
        # timing-only ablations (WRONG results; tools/r06_abl.sh): ARX_ABL_SKIP_SORT leaves the ids-only half out once it
        # ran eagerly (the applies then walk the first batch's lists), ARX_ABL_SKIP_APPLY leaves the apply half out
        if _ABL_SKIP_SORT and phase in (1, 5, 6) and self.warm >= 1:
            return
        if _ABL_SKIP_APPLY and phase in (2, 7, 8) and self.warm >= 1:
            return
        if phase & 1:
            for x, off in zip(ent['xsites'], args.extra_off):   # multi-hot lookups: padded slots, the sort drops the pads
                ops.bag_expand_padded(x.maps[0], x.maps[1], x.maps[2], ids_of(x), x.max_len,
                                      x.node.row0, x.coef, ent['keys'][off:off + x.cap],
                                      ent['src'][off:off + x.cap], ent['coef'][off:off + x.cap])
        node0 = (group[0][1] + group[0][2])[0].node
        if bag is not None:
            bag_entry, bag_live, bag_bias, bag_virtual = bag
            bt, s0 = bag_entry[0], bag_live[0]
            if 'bag_ws' not in ent:
                ent['bag_ws'] = ops.Workspace(rt.device)     # token lists + merged rows, between the phases
            sgd = rt.optimizer == 'sgd'
            if bag_virtual:
                starts_r, lens_r = s0.maps[1], s0.maps[2]
            else:
                starts_r, lens_r = self._rider_csr(group[0][1][0].maps[0], s0.maps[1], s0.maps[2],
                                                   int(group[0][0][0].E.shape[0]))
            max_len = max(x.max_len for x in bag_live)
            if 'csc' not in ent:
                # static token order of the bags (ops.BagCSC: built once per bag index), flags / slots per pass
                ent['csc'] = None
                if rt.csc_mode == '1' or (rt.csc_mode != '0' and bag_virtual):
                    cache = self.__dict__.setdefault('_csc_cache', {})
                    ck2 = (s0.maps[0].data_ptr(), starts_r.data_ptr(), lens_r.data_ptr(), max_len, int(bt.E.shape[0]))
                    if ck2 not in cache:
                        cache[ck2] = ops.BagCSC(s0.maps[0], starts_r, lens_r, max_len, int(bt.E.shape[0]))
                    if cache[ck2].ok:
                        ent['csc'] = (cache[ck2],) + cache[ck2].scratch()
            if ent['csc'] is not None:
                # the flag bytes are zero between steps because every sort half that marks (phase 1 / 5) is followed
                # by the sweep that clears (1 / 6); an eager step that raised in between leaves marks behind, which
                # the next sweep would take for live pairs -- cleared here (never on the replay path: no Python there)
                if phase in (1, 3, 5):
                    if ent.get('csc_marked'):
                        ent['csc'][1].zero_()
                    ent['csc_marked'] = phase == 5
                elif phase == 6:
                    ent['csc_marked'] = False
            ops.sparse_adagrad_cat_multi_bags(
                args, node0.arena, node0.arena_b if (ent['any_bias'] or bag_bias) else None, rt.lr,
                ent['keys'], ent['src'], ent['coef'], ent['ws'], bt.E, None if sgd else bt.acc,
                bt.bias if bag_bias else None, bt.bias_acc if (bag_bias and not sgd) else None,
                s0.maps[0], starts_r, lens_r, max_len, ent['bag_ws'],
                gscale_dev=rt.clip_coef_dev, phase=phase, bag_aux_cnt=self._aux_cnt(bt), csc=ent['csc'],
                split=rt.rider_split(ent['csc'] is not None, bool(bag_virtual)))
            return
        ops.sparse_adagrad_cat_multi(args, node0.arena, node0.arena_b if ent['any_bias'] else None,
                                     rt.lr, ent['keys'], ent['src'], ent['coef'], ent['ws'],
                                     gscale_dev=rt.clip_coef_dev, phase=phase)

    def _early_sort(self):
        """The K7 contributions and their sort depend on the lookup ids only: run them on a side
        stream (a parallel branch of the captured graph) under the forward / backward kernels;
        _apply_sparse joins the branch and only applies.  The jobs are the passes the previous
        execution of this plan ran (static in steady state; verified again at apply time)."""
        self._k7_early = None
        self._k7_fork = None
        jobs = self._early_jobs
        if not jobs or self.rt.dp is not None:
            return                            # (data-parallel: the apply sorts the GATHERED lookups)
        rt = self.rt
        if self._k7_stream is None:
            # (ARX_K7_STREAM_PRIO=1, experiment: the sort branch captured from a high-priority stream)
            self._k7_stream = torch.cuda.Stream(device=rt.device,
                                                priority=-1 if os.environ.get('ARX_K7_STREAM_PRIO') else 0)
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self._k7_fork = (ev, jobs)

    def _early_launch(self):
        """Second half of _early_sort: the branch's kernels.  Issued AFTER the main stream's first
        lookups so that, in the captured graph, the forward chain is the first child of the feed
        node and stays on its queue -- as the first child the branch took that place and the
        step's first gather started ~17 us after the feed instead of right behind it."""
        if self._k7_fork is None:
            return
        ev, jobs = self._k7_fork
        self._k7_fork = None
        self._k7_stream.wait_event(ev)
        with torch.cuda.stream(self._k7_stream):
            mid = None
            for kind, what, key in jobs:
                if kind == 'multi' and what[1] is not None and len(jobs) == 1:
                    # a bag table rides on the pass: the one-hot apply only needs the one-hot sort --
                    # the token chain behind it may still run while that apply does (quarter phases)
                    self._apply_multi(what[0], phase=5, key=key, bag=what[1])
                    self._k7_virtual = bool(what[1][3])
                    mid = torch.cuda.Event()
                    mid.record(self._k7_stream)
                    self._apply_multi(what[0], phase=6, key=key, bag=what[1])
                elif kind == 'multi':
                    self._apply_multi(what[0], phase=1, key=key, bag=what[1])
                elif kind == 'bags':
                    self._bag_pass(what, key, phase=1)
                else:
                    self._cat_pass(what, key, phase=1)
            done = torch.cuda.Event()
            done.record(self._k7_stream)
        if self._ring:
            # the branch sorted the NEXT step's lookups (slot 1 - _ring_par): nothing in this step waits for it
            # but the end of the step (a captured branch has to rejoin its origin)
            self._k7_early = (set(k for _, _, k in jobs), done, 'ring')
            return
        self._k7_early = (set(k for _, _, k in jobs), done, mid)

    def _plan_early(self, jobs, n_passes):
        """Which of this execution's passes may be sorted ahead next time: all of them or none
        (with a separate heavy pass behind the branch -- the multi-hot table of C3 -- it measured
        slower: 469 -> 482 us), and only while the sorts are launch-bound little kernels (beyond
        ~1.25 * 10^5 live contributions they take CUs from the GEMMs: C2 B=65536, 132 k: 791 -> 827 us;
        C4, 103 k: 909 -> 895 us; C3 B=4096, ~118 k: 215 -> 192 us) on a step that is GPU-bound at
        all."""
        cap = int(os.environ.get('ARX_K7_EARLY_MAX', '125000'))
        # a two-stage multi-hot pass has ~100 us of sorts per step (two sorts, the second over the
        # padded bag slots) that would otherwise sit on the critical path: worth the CUs they take
        # from the GEMMs far beyond that size (C3 B=16384, ~220 k: 474 -> 404 us)
        cap_bags = int(os.environ.get('ARX_K7_EARLY_MAX_BAGS', '1500000'))
        # tiny steps are bound by the host-side cost of a graph launch, and a graph with a second
        # branch costs more to launch (C1, B=64: 75 -> 103 us per step with the branch)
        lo = int(os.environ.get('ARX_K7_EARLY_MIN', '8192'))
        n_bags = sum(j[3] for j in jobs if j[0] == 'bags') + sum(j[4] for j in jobs if len(j) > 4)
        n_hot = sum(j[3] for j in jobs if j[0] != 'bags')
        ok = (jobs and len(jobs) == n_passes and lo <= n_bags + n_hot and n_hot <= cap
              and n_bags <= cap_bags)
        self._early_jobs = [j[:3] for j in jobs] if ok else []

    def _cat_pass(self, entry, key, phase):
        """One-hot table, its own pass (arx_sparse_adagrad_cat); key = (live sites, use_bias)."""
        rt = self.rt
        table, sites, bufs, total = entry
        live_ids, use_bias = key
        if bufs.get('cat_key') != live_ids:
            live_sites = [s for s in sites if id(s) in live_ids]
            bufs['cat_key'] = live_ids
            bufs['cat_args'] = ops.CatSiteArgs(
                [(s.maps[0], None, s.ids_node.value, s.node.row0, s.coef) for s in live_sites])
            bufs['cat_node0'] = live_sites[0].node
            bufs['cat_ws'] = ops.Workspace(rt.device)      # the sorted arrays live here between the phases
        if getattr(table, 'aux_first', None) is None:
            table.aux_first = torch.full((table.E.shape[0],), 2 ** 31 - 1, dtype=torch.int32,
                                         device=rt.device)
            table.aux_cnt = torch.zeros((table.E.shape[0],), dtype=torch.int32, device=rt.device)
        node0 = bufs['cat_node0']
        mode = rt.cat_mode if rt.cat_mode else {3: 0, 1: 0x10, 2: 0x20}[phase]
        sgd = rt.optimizer == 'sgd'
        ops.sparse_adagrad_cat(table.E, None if sgd else table.acc, table.bias if use_bias else None,
                               table.bias_acc if (use_bias and not sgd) else None, bufs['cat_args'],
                               node0.arena, node0.arena_b if use_bias else None, rt.lr,
                               table.aux_first, table.aux_cnt, bufs['hot'], bufs['keys'],
                               bufs['src'], bufs['coef'], bufs['cat_ws'], gscale_dev=rt.clip_coef_dev,
                               mode=mode)

    def _bag_pass(self, entry, key, phase):
        """Multi-hot table, two-stage pass (arx_sparse_adagrad_bags); key = (live sites, use_bias)."""
        rt = self.rt
        table, sites, bufs, total = entry
        live_ids, use_bias = key[1], key[2]
        if bufs.get('bag_key') != key:
            live_sites = [s for s in sites if id(s) in live_ids]
            bufs['bag_key'] = key
            bufs['bag_args'] = ops.BagSiteArgs([(s.ids_node.value, s.node.row0, s.coef) for s in live_sites],
                                               max(s.max_len for s in live_sites))
            bufs['bag_site0'] = live_sites[0]
            bufs['bag_ws'] = ops.Workspace(rt.device)      # sorted arrays + merged rows live here between the phases
        s0 = bufs['bag_site0']
        sgd = rt.optimizer == 'sgd'
        ops.sparse_adagrad_bags(table.E, None if sgd else table.acc, table.bias if use_bias else None,
                                table.bias_acc if (use_bias and not sgd) else None, s0.maps[0], s0.maps[1],
                                s0.maps[2], bufs['bag_args'], s0.node.arena,
                                s0.node.arena_b if use_bias else None, rt.lr, bufs['bag_ws'],
                                gscale_dev=rt.clip_coef_dev, phase=phase, aux_cnt=None)

    def _apply_one(self, entry):
        rt = self.rt
        for table, sites, bufs, total in (entry,):
            live_sites = [s for s in sites if s.node._grad_written]
            if not live_sites:
                continue
            n_live = sum(s.n for s in live_sites)
            if self._bags_ok(live_sites):
                use_bias = table.bias is not None and any(s.node.bias_grad_used for s in live_sites)
                key = ('bags', tuple(id(s) for s in live_sites), use_bias)
                early = self._k7_done_keys
                self._bag_pass(entry, key, 2 if (early is not None and key in early) else 3)
                # live tokens after the per-entity merge: about a third of the padded slots hold a
                # token, and Zipf-popular batches repeat entities about twice
                self._jobs.append(('bags', entry, key, sum(s.cap for s in live_sites) // 6))
                self._n_passes += 1
                continue
            if all(s.kind == 'cat' and s.col_off == 0 for s in live_sites) and n_live <= (1 << 22) \
                    and len(live_sites) <= 8 and not rt.force_sort_path:
                # one-hot lookups: own pass (optim_cat.hip); sorted ahead when _early_sort ran it
                use_bias = table.bias is not None and any(s.node.bias_grad_used for s in live_sites)
                key = (tuple(id(s) for s in live_sites), use_bias)
                early = self._k7_done_keys
                self._cat_pass(entry, key, 2 if (early is not None and key in early and not rt.cat_mode) else 3)
                if not rt.cat_mode:
                    self._jobs.append(('cat', entry, key, n_live))
                self._n_passes += 1
                continue
            live = False
            for s in sites:
                node = s.node
                if not node._grad_written:
                    # this lookup received no gradient this step: leave its keys at NONE
                    ops.fill_i32(bufs['keys'][s.key_off:s.key_off + s.cap], KEY_NONE)
                    continue
                live = True
                ks = bufs['keys'][s.key_off:s.key_off + s.cap]
                ss = bufs['src'][s.key_off:s.key_off + s.cap]
                cs = bufs['coef'][s.key_off:s.key_off + s.cap]
                if s.kind == 'cat':
                    ops.sparse_site_onehot(s.maps[0], s.ids_node.value, node.row0, s.coef, ks, ss, cs)
                else:
                    ops.bag_expand_padded(s.maps[0], s.maps[1], s.maps[2], s.ids_node.value, s.max_len,
                                          node.row0, s.coef, ks, ss, cs)
            if not live:
                continue
            node0 = sites[0].node
            G = node0.arena[:, sites[0].col_off:] if sites[0].col_off else node0.arena
            use_bias = table.bias is not None and any(s.node.bias_grad_used for s in sites)
            sgd = rt.optimizer == 'sgd'
            ops.sparse_adagrad(table.E, None if sgd else table.acc, table.bias if use_bias else None,
                               table.bias_acc if (use_bias and not sgd) else None, bufs['keys'], bufs['src'],
                               bufs['coef'], G, node0.arena_b if use_bias else None, rt.lr, rt.ws,
                               gscale_dev=rt.clip_coef_dev, n=total, aux_cnt=self._aux_cnt(table))
            self._n_passes += 1           # a pass that is not sorted ahead: no side branch next time

    def _aux_cnt(self, table):
        """Per-row arrival counters of the window apply (zero between launches)."""
        if getattr(table, 'aux_cnt', None) is None:
            table.aux_first = torch.full((table.E.shape[0],), 2 ** 31 - 1, dtype=torch.int32,
                                         device=self.rt.device)
            table.aux_cnt = torch.zeros((table.E.shape[0],), dtype=torch.int32, device=self.rt.device)
        return table.aux_cnt

    # ---- ring mode: the NEXT step's K7 sort half as this step's side branch -------------------------------------
    def ring_capable(self):
        """The passes of this plan are known (it has run before) and all of the fused kind whose sort half can read
        another id placeholder; hipGraph replay on (ring mode is a property of the captured graph)."""
        rt = self.rt
        return bool(self.train and rt.use_graph and rt.dp is None and self.warm >= 1 and self._early_jobs
                    and all(j[0] == 'multi' for j in self._early_jobs) and len(self._early_jobs) == 1
                    and not rt.no_multi)

    def ring_ids(self):
        """The id placeholders the K7 sites read (each once)."""
        seen, out = set(), []
        for _t, sites, _b, _n in self.tables:
            for s_ in sites:
                n = s_.ids_node
                if id(n) not in seen:
                    seen.add(id(n))
                    out.append(n)
        return out

    def ring_bootstrap(self):
        """First ring step (or the first after a step without prepare_next): nobody sorted THIS step's lookups ahead.
        Do it now, eagerly: the current ids go through the next-step placeholders into slot _ring_par."""
        rt = self.rt
        rt.flush_feeds()
        ops.copy_words([(n.value, n.next_value()) for n in self.ring_ids()])
        self._bind()
        self._ring, par = True, self._ring_par
        try:
            self._ring_par = 1 - par              # (_apply_multi sorts into 1 - _ring_par)
            for kind, what, key in self._early_jobs:
                if what[1] is not None:
                    self._apply_multi(what[0], phase=5, key=key, bag=what[1])
                    self._apply_multi(what[0], phase=6, key=key, bag=what[1])
                else:
                    self._apply_multi(what[0], phase=1, key=key, bag=what[1])
        finally:
            self._ring, self._ring_par = False, par
        self._ring_ready = True

    def _ring_reset(self):
        self._ring_graphs, self._ring_warm, self._ring_ready = {}, {}, False

    def _run_ring(self):
        rt = self.rt
        par = self._ring_par
        self._ring = True
        try:
            if self._ring_warm.get(par, 0) >= 1:
                # one captured executable per parity, launched in turn.  (Re-pointing ONE executable at the other
                # parity's graph with hipGraphExecUpdate saved the ~15 us that alternating executables cost per
                # launch, and segfaulted inside a later hipGraphLaunch of ANOTHER plan once in a full test run:
                # not worth an opt-in mode that does not pay anyway.)
                g = self._ring_graphs.get(par)
                if g is None:
                    g = ops.CapturedGraph()
                    side = torch.cuda.Stream(device=rt.device)
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        g.begin()
                        try:
                            self._execute()
                        finally:
                            g.end()
                    torch.cuda.current_stream().wait_stream(side)
                    self._ring_graphs[par] = g
                g.launch()
            else:
                self._execute()
                self._ring_warm[par] = self._ring_warm.get(par, 0) + 1
        finally:
            self._ring = False
        self._ring_par = 1 - par                   # the branch sorted slot 1 - par: the next step applies from it
        self._ring_ready = True

    def run(self):
        rt = self.rt
        if self.train and self._kp != rt.keep_prob:
            # keep_prob is baked into the kernel sequence (identity vs masked): rebuild
            self._kp = rt.keep_prob
            self.graph = None
            self.warm = 0
            self._ring_reset()
        ring_now = self.ring_req and self._ring_ready and self.ring_capable()
        in_graph = rt.use_graph and self.warm >= 1 and rt.feeds_in_graph and not ring_now
        if not in_graph:
            rt.flush_feeds()
        ring, self.ring_req = self.ring_req, False
        if ring and self._ring_ready and self.ring_capable():
            self._run_ring()
            if self.train:
                for n in self.fetch:
                    if isinstance(n, MeanLoss) and n.lazy and not n.in_scorer():
                        n._stale = True
            return
        self._ring_ready = False
        if rt.use_graph and self.warm >= 1:
            # the step's placeholder feeds travel as the graph's first node(s): their sources are swapped in before
            # the replay (CapturedGraph.set_feeds) -- one submission per step instead of an eager copy + the graph
            pf = rt.take_feeds() if in_graph else []
            if self.graph is None:
                g = ops.CapturedGraph()
                side = torch.cuda.Stream(device=rt.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g.begin()
                    try:
                        if pf:
                            ops.copy_words(pf)
                        self._execute()
                    except BaseException:
                        # (the capture may already be invalidated: ending it raises again -- the ORIGINAL error is the
                        # one that has to surface; advisor, round 4)
                        try:
                            g.end()
                        except BaseException:
                            pass
                        if pf:
                            rt.pending_feeds = list(pf) + list(getattr(rt, 'pending_feeds', []))
                        raise
                    try:
                        g.end(feeds=pf)
                    except BaseException:
                        # feed nodes not found as captured (node count mismatch): the feeds taken out of the queue
                        # go back in front of it, so that an eager retry of the step still feeds its placeholders
                        if pf:
                            rt.pending_feeds = list(pf) + list(getattr(rt, 'pending_feeds', []))
                        raise
                torch.cuda.current_stream().wait_stream(side)
                self.graph = g
            elif pf and self.graph.feeds_match(pf):
                self.graph.set_feeds(pf)
            else:
                if pf:
                    ops.copy_words(pf)            # other placeholders than the captured ones: fed eagerly
                self.graph.set_feeds(None)
            self.graph.launch()
        else:
            self._execute()
            self.warm += 1
        if self.train:
            for n in self.fetch:
                if isinstance(n, MeanLoss) and n.lazy and not n.in_scorer():
                    n._stale = True


_STAGE_POOL = []     # pinned slab sets of collected runtimes (Runtime.host_feed)


def _stage_release(pin):
    try:
        torch.cuda.synchronize()
    except Exception:
        pass
    _STAGE_POOL.append(pin)


class Runtime(object):
    def __init__(self, device=None, learning_rate=0.1, use_graph=True):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("arx needs an AMD GPU: the hot path is HIP-only (no CPU fallback)")
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.ws = ops.Workspace(self.device)
        self.scratch = ops.new_reduce_scratch(self.device)     # this runtime's own reduce scratch (re-entrancy)
        self.nodes = []
        self.tables = {}
        self.dense = {}
        self.lr = torch.tensor([float(learning_rate)], dtype=torch.float32, device=self.device)
        self.lr_host = float(learning_rate)
        self.clip_coef_dev = None
        self.max_gradient_norm = None
        self.use_graph = use_graph
        self.global_step = 0
        self.pre_apply_hooks = []
        self.optimizer = 'adagrad'      # or 'sgd' (tf.train.GradientDescentOptimizer, seqModel.py:176)
        self.seed = 0
        self.keep_prob = 1.0            # dropout keep probability of train plans (Dropout nodes)
        self.dropout_calls = 0
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)   # device step counter
        self.pending_feeds = []         # (src, placeholder buffer) device-to-device feeds not yet issued
        self.feeds_in_graph = True      # captured plans take them in as graph nodes (False: an eager launch per step)
        self._stage = None              # host-fed placeholders: pinned + device slabs (host_feed)
        self.stage_host_feeds = os.environ.get('ARX_STAGE_FEEDS', '1') != '0'
        # K7 pass selection overrides (attributes, not environment switches: set them on the runtime object to
        # force a pass shape that the sizes would not pick; all False / 0 in production)
        self.force_sort_path = False     # every table through arx_sparse_adagrad (explicit triples)
        self.cat_mode = 0                # 1: atomic-election one-hot pass (arx_sparse_adagrad_cat mode 1)
        self.no_multi = False            # one K7 pass per table
        self.no_bags = False             # contribution-level multi-hot pass
        self.no_rider = False            # the bag table keeps its own two-stage pass
        self.no_virtual = False          # ... unless an id table shares its lookups
        # round 6: the riding bag table's tokens in their STATIC order (csrc/csc.hip) instead of a sort per step.
        # Measured (profiles/r06_csc_ab.txt, alternating runs on one box): MIX layout (virtual entity table) 250 ->
        # 236 us per step, HET 231 -> 235..239 -- so: ARX_K7_CSC unset = where the entity table is virtual, 1 = every
        # riding bag table, 0 = the per-step expansion + radix sort of rounds 2-6 everywhere
        self.csc_mode = os.environ.get('ARX_K7_CSC', 'auto')
        self.dp = None                  # arx.dist.SeqDataParallel: gradient exchange between replicas

    def rider_split(self, csc, virtual):
        """The apply form of a pass with a riding bag table (arx.h, ARX_K7_RIDER): the environment's choice if it
        makes one; else `split` (entity runs, then ONE launch with the token runs and the other tables' runs) where the
        step's sort branch has room for the one-hot list's run records -- with the static token order -- and `win`
        (window + finish + token apply) otherwise.  Measured: profiles/r06_csc_ab.txt."""
        if os.environ.get('ARX_K7_RIDER') in ('win', 'split'):
            return None
        return bool(csc)

    def drop_feed(self, dst):
        """Forget queued feeds whose destination is `dst` (same start address: placeholders and
        their bucket views alias)."""
        if self.pending_feeds:
            p = dst.data_ptr()
            self.pending_feeds = [(s_, d_) for s_, d_ in self.pending_feeds if d_.data_ptr() != p]

    def queue_feed(self, src, dst):
        """Queue a device-to-device placeholder feed; the latest feed of a destination wins.  The
        source tensor is referenced, not copied: it must stay unmodified until the plan runs."""
        self.drop_feed(dst)
        self.pending_feeds.append((src, dst))

    # ---- host-fed placeholders (the reference's own hand-over: python lists / numpy arrays into feed_dict,
    # hmf_model.py:162-175, seqModel.py:289-324).  A pageable `copy_` per placeholder blocks the host for the whole
    # transfer (measured: C3 348-355 us per step against 231 with device-resident ids).  Here the step's host arrays
    # are packed into ONE pinned slab (hipHostMalloc memory is mapped into the GPU's address space) and queued like
    # device feeds with the slab as their source: the step's feed launch -- the captured graph's first node -- reads
    # the ids straight out of host memory over the link, no copy engine, no second stream, no event in the step's
    # stream.  (A first form -- one asynchronous copy per step on a side stream into a device slab, two events and a
    # wait in the main stream -- measured 255 us: the markers between the graph launches cost more than the copy.)
    # Eight slabs in two halves: one event per four steps says that the readers of a half are done before the host
    # writes it again (the host blocks there only when it is more than four steps ahead of the GPU).
    _STAGE_SLOTS = 8
    _STAGE_WORDS = 1 << 20              # 4 MB per slab; a larger step of host feeds falls back to direct copies

    def host_feed(self, dst, arr):
        """Feed the placeholder buffer `dst` (contiguous, 4-byte elements) from the host array `arr` (same element
        count, already of dst's dtype).  Returns False when the staged route is not available (caller copies directly)."""
        n = dst.numel()
        same = ((dst.dtype == torch.int32 and arr.dtype == np.int32)
                or (dst.dtype == torch.float32 and arr.dtype == np.float32))
        if not (self.stage_host_feeds and dst.is_cuda and dst.is_contiguous() and same and arr.size == n
                and n <= self._STAGE_WORDS):
            return False
        st = self._stage
        if st is None:
            S, W = self._STAGE_SLOTS, self._STAGE_WORDS
            # the slabs outlive this runtime's last submitted step: a collected runtime hands them back to a module
            # pool after a device synchronisation (the feed launch reads them in place -- the pinned allocator does
            # not know about that reader and would recycle the block under it)
            pin = _STAGE_POOL.pop() if _STAGE_POOL else [torch.empty(W, dtype=torch.int32).pin_memory()
                                                         for _ in range(S)]
            weakref.finalize(self, _stage_release, pin)
            st = self._stage = {'pin': pin, 'done': [None, None], 'k': 0, 'used': 0}
            st['np'] = [b.numpy() for b in st['pin']]
        k = st['k']
        if st['used'] + n > self._STAGE_WORDS:
            return False
        half = self._STAGE_SLOTS // 2
        if st['used'] == 0 and k % half == 0 and st['done'][k // half] is not None:
            st['done'][k // half].synchronize()          # the steps that read this half of the slabs have finished
        o = st['used']
        st['np'][k][o:o + n] = np.ascontiguousarray(arr).reshape(-1).view(np.int32)
        st['used'] = (o + n + 3) & ~3                    # (16-byte aligned pieces)
        src = st['pin'][k][o:o + n]
        self.queue_feed(src if dst.dtype == torch.int32 else src.view(dst.dtype), dst.reshape(-1))
        return True

    def _stage_commit(self):
        """The feeds queued so far are about to be issued (or handed to the graph): move on to the next slab."""
        st = self._stage
        if st is None or st['used'] == 0:
            return
        S = self._STAGE_SLOTS
        half = S // 2
        k = (st['k'] + 1) % S
        if k % half == 0:
            # entering a half: what has been submitted so far includes every reader of the OTHER half's previous
            # round except the one about to be issued from slab k - 1 -- so the event is recorded one commit later
            st['mark'] = 1 - k // half
        elif st.get('mark') is not None:
            h = st.pop('mark')
            ev = st['done'][h] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())       # all readers of half h's slabs are in front of this point
            st['done'][h] = ev
        st['k'], st['used'] = k, 0

    def take_feeds(self):
        """The queued placeholder feeds, for a caller that issues them itself (Plan.run: as graph nodes)."""
        self._stage_commit()
        pf, self.pending_feeds = self.pending_feeds, []
        return pf

    def flush_feeds(self):
        """Issue the queued placeholder feeds (one launch per eight buffers)."""
        self._stage_commit()
        if self.pending_feeds:
            pf, self.pending_feeds = self.pending_feeds, []
            ops.copy_words(pf)

    def set_learning_rate(self, v):
        self.lr_host = float(v)
        ops.fill_f32(self.lr, self.lr_host)

    def topo(self, fetch):
        seen, order = set(), []

        def visit(n):
            if id(n) in seen:
                return
            seen.add(id(n))
            for i in n.inputs:
                visit(i)
            for extra in getattr(n, 'extra_inputs', ()):
                visit(extra)
            order.append(n)
        for f in fetch:
            visit(f)
        return order

    def pre_apply(self, plan):
        for h in self.pre_apply_hooks:
            h(plan)

    def apply_dense(self, plan):
        todo = [p for p in self.dense.values() if getattr(p, 'touched', False)]
        sgd = self.optimizer == 'sgd'
        if len(todo) == 1:
            p = todo[0]
            ops.adagrad_dense(p.w, None if sgd else p.acc, p.grad, self.lr, gscale_dev=self.clip_coef_dev)
        elif todo:                          # all dense parameters of the step: one launch
            ops.adagrad_dense_multi([(p.w, None if sgd else p.acc, p.grad) for p in todo], self.lr,
                                    gscale_dev=self.clip_coef_dev)
        for p in todo:
            p.touched = False

    def upload(self, arr, dtype):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(arr))).to(dtype)
        return t.to(self.device)
