"""EmbeddingAttribute -- the reference's hot-path class on MI355X.

Same constructor and method surface as attributes/embed_attribute.py:19-747 of
the reference (get_batch_user / get_batch_item / get_prediction /
get_target_score / compute_loss / get_warp_mask / prepare_warp /
target_mapping / add_input / get_*_model_size); the bodies build arx.graph
nodes whose kernels live in libarx.so instead of TF ops.  Differences a caller
can observe:
  * returned objects are arx.graph.Node handles, not tf.Tensors;
  * the scorer is evaluated in embedding space (pool rows gathered, then one
    [mb,d]x[d,S] GEMM) instead of multiplying the whole attribute table
    (:171,188) -- algebraically identical, summation order differs;
  * errors raise ValueError / NotImplementedError instead of print+exit();
  * tables may be supplied through `params` (reference variable names) --
    otherwise glorot-uniform like tf.get_variable's default.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import graph as G
from .. import ops


def _np_i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


class FeatureList(list):
    """What _get_embedded(concatenation=False) returns in the reference: the list
    cat_list + mulhot_list.  The fused node computing their mean is attached, so
    arx.graph-level reduce_mean(list, 0) costs nothing extra."""

    def __init__(self, mean_node, n):
        super().__init__([mean_node] * n)
        self.mean_node = mean_node


class ZeroEmbed(G.Node):
    """tf.zeros([mb, dim]) of the no_id branch (embed_attribute.py:356-366)."""

    def __init__(self, rt, n, d):
        super().__init__(rt, (n, d))
        self.value = torch.zeros((n, d), dtype=torch.float32, device=rt.device)

    def forward(self, train):
        pass


class TargetMapping(G.Node):
    """item_target = item_ind2logit_ind[item_id_target] (embed_attribute.py:676-695 target_mapping)
    as a node of the step: the mapping runs inside the captured plan instead of as an eager launch
    in front of it (a B=64 step is bound by exactly those).  `feed` keeps the placeholder
    interface for host-mapped targets (the in-plan mapping rewrites the same values)."""

    def __init__(self, rt, ids_node, emb, name):
        super().__init__(rt, ids_node.shape, (ids_node,))
        self.emb, self.name = emb, name
        self.value = torch.zeros(ids_node.shape[0], dtype=torch.int32, device=rt.device)

    def feed(self, arr):
        src = arr if isinstance(arr, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(np.asarray(arr, dtype=np.int32)))
        self.value.copy_(src.reshape(self.value.shape), non_blocking=True)

    def forward(self, train):
        if self.emb._item2logit_dev is None:
            raise ValueError("target mapping needs item_ind2logit_ind")
        ops.sparse_site_onehot(self.emb._item2logit_dev, self.inputs[0].value, 0, 0.0, self.value, None, None)


class Dropout(G.Node):
    """tf.nn.dropout (embed_attribute.py:236; DropoutWrapper seqModel.py:100,103).  keep_prob is
    read from rt.keep_prob when the plan is built / captured (1.0 => identity; changing it makes
    the plans re-capture); RNG is counter-based (not TF's Philox): parity tests replay the
    drawn masks (`keep`) through the oracle."""

    requires_grad = True
    uses_dropout = True          # plans with such nodes bump the device step counter every step

    def __init__(self, rt, x):
        super().__init__(rt, x.shape, (x,))
        self.keep = None
        rt.dropout_calls += 1
        self.sid = rt.dropout_calls          # fixed stream id of this node; the step counter
                                             # (rt.step_dev, bumped once per train step inside the
                                             # plan / captured graph) varies the draw per step

    def forward(self, train):
        x = self.inputs[0]
        kp = self.rt.keep_prob if train else 1.0
        self.kp_used = kp
        if kp >= 1.0:
            self.value = x.value
            return
        if self.keep is None:
            self.keep = torch.empty(int(np.prod(self.shape)), dtype=torch.uint8, device=self.rt.device)
            self._own = torch.empty(self.shape, dtype=torch.float32, device=self.rt.device)
        self.value = self._own
        ops.dropout_fwd_step(x.value, kp, self.rt.seed * 1000003 + self.sid, self.rt.step_dev, self.value,
                             self.keep)

    def alloc_grad(self):
        x = self.inputs[0]
        if getattr(self, 'kp_used', 1.0) >= 1.0:
            self.grad = x.alloc_grad()          # identity: share the gradient buffer
            return self.grad
        return super().alloc_grad()

    def grad_beta(self):
        if getattr(self, 'kp_used', 1.0) >= 1.0:
            self._grad_written = True
            return self.inputs[0].grad_beta()
        return super().grad_beta()

    def backward(self):
        if self.kp_used >= 1.0:
            return
        x = self.inputs[0]
        if x.grad_beta() != 0.0:
            raise NotImplementedError("dropout input with several consumers")
        ops.dropout_bwd(self.grad, self.keep, self.kp_used, x.alloc_grad())


class EmbeddingAttribute(object):
    def __init__(self, user_attributes, item_attributes, mb, n_sampled, input_steps=0,
                 item_output=False, item_ind2logit_ind=None, logit_ind2item_ind=None,
                 indices_item=None, devices=['/gpu:0'], params=None, runtime=None, seed=0):
        self.user_attributes = user_attributes
        self.item_attributes = item_attributes
        self.batch_size = mb
        self.n_sampled = n_sampled
        self.input_steps = input_steps
        self.item_output = item_output
        self.num_item_features = item_attributes.num_features_cat + item_attributes.num_features_mulhot
        self.item_ind2logit_ind = item_ind2logit_ind
        self.logit_ind2item_ind = logit_ind2item_ind
        if logit_ind2item_ind is not None:
            self.logit_size = len(logit_ind2item_ind)
        self.indices_item = indices_item if indices_item is not None else range(self.logit_size)
        self.devices = devices
        self.rt = runtime if runtime is not None else G.Runtime()
        rt = self.rt
        rt.seed = seed
        self._rng = torch.Generator(device=rt.device)
        self._rng.manual_seed(seed)
        params = params or {}

        # ---- attribute maps -> device constants (embed_attribute.py:308-318) ----
        self.att = {'user': self._init_attributes(user_attributes),
                    'item': self._init_attributes(item_attributes)}
        if item_output:
            self.att['item_output'] = self.att['item']
        self.n_users = self._n_entities(user_attributes)
        self.n_items = self._n_entities(item_attributes)

        # ---- tables (:265-306) ----
        self.tables = {}
        self.user_feats = self._embedded(user_attributes, 'user', False, params)
        self.item_feats = self._embedded(item_attributes, 'item', True, params)
        if item_output:
            self.item_out_feats = self._embedded(item_attributes, 'item_output', True, params)
        else:
            self.item_out_feats = self.item_feats

        # ---- placeholders (:72-93) ----
        self.u_indices = {'input': G.IdsInput(rt, mb, 'user_input_ind')}
        self.i_indices = {}
        self.i_indices['pos'] = G.IdsInput(rt, mb, 'item_pos_ind')
        self.i_indices['neg'] = G.IdsInput(rt, mb, 'item_neg_ind')
        if n_sampled is not None:
            self.i_indices['sampled_pass'] = G.IdsInput(rt, n_sampled, 'item_sampled_ind')
        if input_steps > 0:
            # one [L*mb] time-major buffer; 'input{t}' are row slices of it so the
            # whole sequence is looked up by ONE launch
            self.input_all = G.IdsInput(rt, input_steps * mb, 'item_input_all')
            for step in range(input_steps):
                self.i_indices['input{}'.format(step)] = ('input_all', step)

        # ---- pools (:96-116) ----
        self._pool_nodes = {}
        self._old_pool = None
        self.item2slot = None
        if n_sampled is not None:
            self.item2slot = torch.full((self.n_items + 1,), -1, dtype=torch.int32, device=rt.device)
            # 1 bit per item "in the pool?" in front of the map (arx_slot_map_attach_bitmap)
            self._pool_bits = torch.zeros(((self.n_items + 1) + 32) // 32, dtype=torch.int32, device=rt.device)
            ops.slot_map_attach_bitmap(self.item2slot, self._pool_bits)
        self._item2logit_dev = None
        self._item2logit_np = None
        if item_ind2logit_ind is not None:
            m = np.full((self.n_items + 2,), -1, dtype=np.int32)
            if isinstance(item_ind2logit_ind, dict):
                keys = np.fromiter(item_ind2logit_ind.keys(), dtype=np.int64, count=len(item_ind2logit_ind))
                vals = np.fromiter(item_ind2logit_ind.values(), dtype=np.int64, count=len(item_ind2logit_ind))
                ok = keys < len(m)
                m[keys[ok]] = vals[ok]
            else:
                a = np.asarray(item_ind2logit_ind)
                m[:len(a)] = a
            self._item2logit_np = m
            self._item2logit_dev = rt.upload(m, torch.int32)

        self.mask = {}
        self.pos_item_set = None
        self.pos_item_set_eval = None
        self._pos_dev = {}
        self.set_mask, self.reset_mask = {}, {}

    # ------------------------------------------------------------------ setup
    @staticmethod
    def _n_entities(att):
        if att.num_features_cat > 0:
            return len(att.features_cat[0]) - 1
        return len(att.mulhot_lengths[0]) - 1

    def _init_attributes(self, att):
        rt = self.rt
        cat = [rt.upload(_np_i32(att.features_cat[i]), torch.int32) for i in range(att.num_features_cat)]
        mul = []
        for i in range(att.num_features_mulhot):
            lens = _np_i32(att.mulhot_lengths[i])
            mul.append((rt.upload(_np_i32(att.features_mulhot[i]), torch.int32),
                        rt.upload(_np_i32(att.mulhot_starts[i]), torch.int32),
                        rt.upload(lens, torch.int32), int(lens.max()) if len(lens) else 1))
        return cat, mul

    def _new_var(self, name, shape, params):
        rt = self.rt
        if name in params:
            a = np.asarray(params[name], dtype=np.float32).reshape(shape)
            return rt.upload(a, torch.float32).contiguous()
        # tf.get_variable default initializer: glorot_uniform
        fan_in, fan_out = shape[0], (shape[1] if len(shape) > 1 else 1)
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        t = torch.empty(shape, dtype=torch.float32, device=rt.device)
        t.uniform_(-lim, lim, generator=self._rng)
        return t

    def _embedded(self, attributes, prefix, with_bias, params):
        """embed_attribute.py:265-306 _embedded + _embedded_bias."""
        feats = []
        maps_cat, maps_mul = self.att['item' if prefix.startswith('item') else 'user']
        for i in range(attributes.num_features_cat):
            d = attributes._embedding_size_list_cat[i]
            V = attributes._embedding_classes_list_cat[i]
            ename = '%sembed_cat_%d' % (prefix, i)
            bname = '%s_bias_cat_%d' % (prefix, i) if with_bias else None
            E = self._new_var(ename, (V, d), params)
            b = self._new_var(bname, (V, 1), params).reshape(V).contiguous() if with_bias else None
            t = G.Table(ename, bname, E, b)
            self.tables[ename] = t
            feats.append(G.Feature('cat', t, (maps_cat[i],)))
        for i in range(attributes.num_features_mulhot):
            d = attributes._embedding_size_list_mulhot[i]
            V = attributes._embedding_classes_list_mulhot[i]
            ename = '%sembed_mulhot_%d' % (prefix, i)
            bname = '%s_bias_mulhot_%d' % (prefix, i) if with_bias else None
            E = self._new_var(ename, (V, d), params)
            b = self._new_var(bname, (V, 1), params).reshape(V).contiguous() if with_bias else None
            t = G.Table(ename, bname, E, b)
            self.tables[ename] = t
            vals, starts, lens, mx = maps_mul[i]
            feats.append(G.Feature('mulhot', t, (vals, starts, lens), max_len=mx))
        return feats

    # ------------------------------------------------------------- variables
    def get_params(self):
        """{reference variable name: numpy array} (tables [Vf,d], biases [Vf,1])."""
        out = {}
        for t in self.tables.values():
            out[t.name] = t.E.cpu().numpy()
            if t.bias is not None:
                out[t.bias_name] = t.bias.cpu().numpy().reshape(-1, 1)
        return out

    def get_slots(self):
        out = {}
        for t in self.tables.values():
            out[t.name] = t.acc.cpu().numpy()
            if t.bias is not None:
                out[t.bias_name] = t.bias_acc.cpu().numpy().reshape(-1, 1)
        return out

    # --------------------------------------------------------------- lookups
    def _select_feats(self, feats, attributes, no_id=False, no_attribute=False):
        ncat = attributes.num_features_cat
        cat, mul = feats[:ncat], feats[ncat:]
        if no_attribute:                                  # embed_attribute.py:368-369
            cat, mul = cat[:1], []
        if no_id:
            cat = cat[1:]                                 # :372-373
        return cat + mul

    def get_batch_user(self, keep_prob, concat=True, no_id=False, device='/gpu:0'):
        """embed_attribute.py:222-237 -> (embedded_user [mb, d or sum d], None)."""
        ua = self.user_attributes
        ids = self.u_indices['input']
        if no_id and ua.num_features_cat == 1:            # :356-366
            node = ZeroEmbed(self.rt, self.batch_size, ua._embedding_size_list_cat[0])
        else:
            feats = self._select_feats(self.user_feats, ua, no_id=no_id)
            node = G.EntityEmbed(self.rt, ids, feats, with_bias=False, concat=concat)
        if not (isinstance(keep_prob, (int, float)) and float(keep_prob) == 1.0):
            if isinstance(keep_prob, (int, float)):
                self.rt.keep_prob = float(keep_prob)
            node = Dropout(self.rt, node)
        return node, None

    def _ids_node(self, name):
        h = self.i_indices[name]
        if isinstance(h, tuple):          # 'input{t}': time step t of the sequence buffer (:87-90)
            step = h[1]
            mb = self.batch_size
            h = G.IdsSlice(self.rt, self.input_all, step * mb, mb, 'item_%s_ind' % name)
            self.i_indices[name] = h
        return h

    def get_batch_item(self, name, batch_size, concat=False, keep_prob=1.0, no_attribute=False,
                       device='/gpu:0'):
        """embed_attribute.py:239-254.  concat=False returns (feature list, bias);
        reduce the list with arx.graph-level `reduce_mean` (== tf.reduce_mean(.., 0))."""
        if name not in self.i_indices:
            raise ValueError("unknown item placeholder %r" % name)
        if keep_prob != 1.0:
            raise NotImplementedError('otherwise not implemented')     # :242
        ids = self._ids_node(name)
        feats = self._select_feats(self.item_feats, self.item_attributes, no_attribute=no_attribute)
        if concat:
            node = G.EntityEmbed(self.rt, ids, feats, with_bias=False, concat=True)
            return node, None
        node = G.EntityEmbed(self.rt, ids, feats, with_bias=True)
        return FeatureList(node, len(feats)), node

    def get_batch_item_seq(self, steps, concat=False, no_attribute=False, out_scale=1.0):
        """All `steps` input placeholders ('input0'..) looked up by one launch:
        value rows are time-major [t*mb + b] (the batched form of seqModel.py:150-156)."""
        feats = self._select_feats(self.item_feats, self.item_attributes, no_attribute=no_attribute)
        return G.EntityEmbed(self.rt, self.input_all, feats, with_bias=False, concat=concat,
                             out_scale=out_scale)

    # ----------------------------------------------------------------- pools
    def _pool_embed(self, pool, output_feat):
        key = (pool, output_feat)
        if key in self._pool_nodes:
            return self._pool_nodes[key]
        if output_feat not in (0, 1):
            raise ValueError("_pool_embed serves output_feat 0 / 1; 2 / 3 pool in score space (_pooled_prediction)")
        rt = self.rt
        ia = self.item_attributes
        if pool == 'sampled':
            if self.n_sampled is None:
                raise ValueError("sampled pool requested but n_sampled is None")
            ids = self.i_indices['sampled_pass']
            feats = self._select_feats(self.item_out_feats, ia, no_attribute=(output_feat == 0))
        elif pool == 'full':
            # logit-ordered copies (embed_attribute.py:100-108 / preprocess.py:240-326)
            V = self.logit_size
            ids = G.IdsInput(rt, V, 'full_pool_ids')
            ids.value.copy_(torch.arange(V, dtype=torch.int32, device=rt.device))
            feats = []
            ncat = ia.num_features_cat if output_feat else 1
            nmul = ia.num_features_mulhot if output_feat else 0
            for i in range(ncat):
                if len(ia.full_cat_tr) > i:
                    cmap = rt.upload(_np_i32(ia.full_cat_tr[i]), torch.int32)
                else:
                    cmap = rt.upload(_np_i32(ia.features_cat[i])[_np_i32(self.logit_ind2item_ind)], torch.int32)
                feats.append(G.Feature('cat', self.item_out_feats[i].table, (cmap,)))
            for i in range(nmul):
                lens = np.asarray(ia.full_lengths_tr[i], dtype=np.float64).reshape(-1).astype(np.int32)
                starts = np.zeros(V, dtype=np.int32)
                starts[1:] = np.cumsum(lens)[:-1]
                f = self.item_out_feats[ia.num_features_cat + i]
                feats.append(G.Feature('mulhot', f.table,
                                       (rt.upload(_np_i32(ia.full_values_tr[i]), torch.int32),
                                        rt.upload(starts, torch.int32), rt.upload(lens, torch.int32)),
                                       max_len=int(lens.max())))
        else:
            raise ValueError("pool must be 'full' or 'sampled'")
        node = G.EntityEmbed(rt, ids, feats, with_bias=True)
        self._pool_nodes[key] = node
        return node

    def get_prediction(self, latent, pool='full', device='/gpu:0', output_feat=1, pred_cls=None, steps=None):
        """embed_attribute.py:148-206 -> logits [rows(latent), V or n_sampled].
        output_feat 0 / 1: embedding-space scorer (pool rows averaged first, one GEMM);
        2 / 3: the multi-hot features pool their per-token SCORES (max / log-sum-exp, :194-200).
        pred_cls: scorer node class (SeqModel passes its per-time-step variant); steps = (L, mb): the
        latent's rows are L unrolled steps -- the reference calls get_prediction once per step
        (seqModel.py:480-493), so output_feat 3 takes ONE reduce_max per step (:197)."""
        mk = pred_cls or (lambda lat, pe: G.Prediction(self.rt, lat, pe))
        if isinstance(latent, list):          # one latent per output feature (:169, :178)
            if output_feat not in (1, 2, 3):
                raise NotImplementedError('Error: Attribute combination not implemented!')
            return self._pooled_prediction(latent, pool, output_feat, mk, steps)
        if output_feat in (0, 1):
            return mk(latent, self._pool_embed(pool, output_feat))
        if output_feat not in (2, 3):
            raise NotImplementedError('Error: Attribute combination not implemented!')    # :202
        return self._pooled_prediction(latent, pool, output_feat, mk, steps)

    def _pool_ids_and_maps(self, pool):
        """(ids node, per-feature maps in pool order, W): the sampled placeholder with the item
        maps, or 0..V-1 with the logit-ordered copies (embed_attribute.py:100-108)."""
        rt, ia = self.rt, self.item_attributes
        key = ('ids', pool)
        if key in self._pool_nodes:
            return self._pool_nodes[key]
        if pool == 'sampled':
            ids = self.i_indices['sampled_pass']
            feats = self._select_feats(self.item_out_feats, ia)
            res = (ids, [(f.kind, f.table, f.maps, f.max_len) for f in feats], ids.shape[0], False)
        else:
            V = self.logit_size
            full = self._pool_embed('full', 1)             # builds the logit-ordered maps once
            res = (full.inputs[0], [(f.kind, f.table, f.maps, f.max_len) for f in full.feats], V, True)
        self._pool_nodes[key] = res
        return res

    def _pooled_prediction(self, latent, pool, output_feat, mk, steps=None):
        """logits = mean over output features of: the plain score for a categorical feature, the
        pooled token scores of a multi-hot one."""
        rt = self.rt
        ids, fmaps, W, static = self._pool_ids_and_maps(pool)
        if isinstance(latent, list) and len(latent) < len(fmaps):
            raise ValueError("latent list shorter than the number of output features")
        lat_all = latent
        parts = []
        for k, (kind, table, maps, max_len) in enumerate(fmaps):
            latent = lat_all[k] if isinstance(lat_all, list) else lat_all
            if kind == 'cat' or output_feat == 1:      # plain score / mean over the bag (embedding space)
                pe = G.EntityEmbed(rt, ids, [G.Feature(kind, table, maps, max_len)], with_bias=True)
                parts.append(mk(latent, pe))
                continue
            if static:
                cap = int(maps[2].sum().item())
            else:
                cap = W * max_len
            cap = (cap + 3) // 4 * 4
            bag = G.BagTokens(rt, ids, maps, cap, static=static)
            tokf = G.Feature('cat', table, (None,))
            tokf._inj = False                   # bag tokens repeat: shared table rows (clip norm)
            te = G.EntityEmbed(rt, bag, [tokf], with_bias=True)
            scores = mk(latent, te)
            gmax = None
            if output_feat == 3:
                if type(scores) is not G.Prediction and steps is None:
                    raise ValueError("output_feat 3 under a per-time-step scorer needs steps = (L, mb)")
                gmax = G.GlobalMax(rt, latent, table, steps=steps)
                vf = G.Feature('cat', table, (None,))
                vf._inj = gmax.L == 1           # (the steps' arg-max rows may coincide)
                gmax.vstar = G.EntityEmbed(rt, G._IdsOf(rt, gmax, gmax.vrows, 'gmax_row'), [vf], with_bias=True)
                # clip_by_global_norm (seqModel.py:180): the residual rows are part of the steps' dense matmul
                # gradients of `table`, not a lookup of their own (SeqModel._clip_hook)
                gmax.vstar._is_gmax_vstar = True
                te._gmax = gmax
                for pnode in parts + [scores]:          # its backward must follow every scorer's
                    pnode.extra_inputs = tuple(getattr(pnode, 'extra_inputs', ())) + (gmax,)
            parts.append(G.SegmentPool(rt, scores, bag, W, output_feat, gmax))
        return parts[0] if len(parts) == 1 else G.MeanOf(rt, parts)

    def get_target_score(self, latent, inds, device='/gpu:0'):
        """embed_attribute.py:208-220; `inds` is an item-index placeholder."""
        feats = self._select_feats(self.item_out_feats, self.item_attributes)
        te = G.EntityEmbed(self.rt, inds, feats, with_bias=True)
        return G.TargetScore(self.rt, latent, te)

    def update_sampled_pool(self, item_sampled):
        """embed_attribute.py:320-348 update_sampled: stage the negative pool.  The
        packed token ids / segment ids the reference materialises are produced on
        the fly by the fused gather kernel; what persists is the id list and the
        item -> slot map (the device twin of item_sampled_id2idx)."""
        buf = self.i_indices['sampled_pass']
        if self._old_pool is not None:
            ops.slot_map_set(self.item2slot, buf.value, clear=True)     # buf still holds the previous pool
        self._old_pool = True
        buf.feed(item_sampled)
        self.rt.flush_feeds()
        ops.slot_map_set(self.item2slot, buf.value, clear=False)

    # ------------------------------------------------------------------ loss
    def _mask_state(self, loss, rows):
        if loss not in self.mask:
            W = self.n_sampled if loss in ('mw', 'mce') else self.logit_size       # :652
            if loss in ('mw', 'mce'):
                getter = lambda: self.item2slot
            else:
                getter = lambda: self._item2logit_dev
            self.mask[loss] = G.MaskState(self.rt, self.batch_size, W, self.u_indices['input'], getter,
                                          lambda: self._pos_dev[self._pos_mode])
        return self.mask[loss]

    def compute_loss(self, logits, item_target, loss='ce', true_rank=False, loss_func='log',
                     exp_p=1.005, device='/gpu:0'):
        """embed_attribute.py:525-549.  Implemented on device: 'ce', 'warp', 'mw', 'warp_eval',
        'rs', 'rs-sig', 'rs-sig2', 'bbpr' (loss_func log/exp/poly/poly2/linear/square), and 'mce'
        -- accepted by the reference's assert (:527) but without a branch there: BUILD-DEFINED as
        the sampled softmax log(1 + sum_s m_rs exp(x_rs - t_r)) in the shape of 'mw' (arx.h).
        bpr* needs feeds the reference commented out."""
        if loss not in ['ce', 'mce', 'warp', 'warp_eval', 'rs', 'rs-sig', 'rs-sig2', 'mw', 'bbpr',
                        'bpr', 'bpr-hinge']:
            raise ValueError("unknown loss %r" % loss)
        if loss in ('ce',):
            return G.BatchLoss(self.rt, 'ce', logits, item_target)
        if loss in ('warp', 'mw', 'mce', 'warp_eval'):
            ms = self._mask_state(loss, logits.shape[0])
            node = G.BatchLoss(self.rt, loss, logits, item_target, mask=ms, mask_rows=self.batch_size)
            if loss == 'warp_eval':
                return [node, node]
            return node
        if loss in ('rs', 'rs-sig', 'rs-sig2', 'bbpr'):                   # :532-547, :551-603
            ms = self._mask_state(loss, logits.shape[0])
            return G.BatchLoss(self.rt, loss, logits, item_target, mask=ms, mask_rows=self.batch_size,
                               loss_func=loss_func, exp_p=exp_p)
        raise NotImplementedError('Error: not implemented other loss!!')   # :548

    def get_warp_mask(self, device='/gpu:0'):
        """embed_attribute.py:662-672 -> (set_mask, reset_mask) callables per loss."""
        self.set_mask, self.reset_mask = {}, {}
        for loss, ms in self.mask.items():
            self.set_mask[loss] = (lambda m=ms: m.scatter(0))
            self.reset_mask[loss] = (lambda m=ms: m.scatter(1))
        return self.set_mask, self.reset_mask

    def prepare_warp(self, pos_item_set, pos_item_set_eval):
        """embed_attribute.py:674-677.  Accepts the reference's {user: [items]}
        dicts or prebuilt CSR pairs (ptr[n_users+1], items) for large synthetic sets."""
        self.pos_item_set = pos_item_set
        self.pos_item_set_eval = pos_item_set_eval
        self._pos_dev = {'train': self._pos_csr(pos_item_set), 'eval': self._pos_csr(pos_item_set_eval)}
        self._pos_mode = 'train'

    def _pos_csr(self, pos):
        rt = self.rt
        nu = self.n_users + 1
        if pos is None:
            ptr = np.zeros(nu + 1, dtype=np.int32)
            items = np.zeros(1, dtype=np.int32)
        elif isinstance(pos, tuple):
            ptr, items = _np_i32(pos[0]), _np_i32(pos[1])
            if len(ptr) < nu + 1:
                ptr = np.concatenate([ptr, np.full(nu + 1 - len(ptr), ptr[-1], dtype=np.int32)])
        else:
            counts = np.zeros(nu, dtype=np.int64)
            for u, its in pos.items():
                if 0 <= u < nu:
                    counts[u] = len(its)
            ptr = np.zeros(nu + 1, dtype=np.int32)
            ptr[1:] = np.cumsum(counts)
            items = np.zeros(max(1, int(ptr[-1])), dtype=np.int32)
            for u, its in pos.items():
                if 0 <= u < nu and len(its):
                    items[ptr[u]:ptr[u + 1]] = np.asarray(list(its), dtype=np.int32)
        if len(items) == 0:
            items = np.zeros(1, dtype=np.int32)
        return rt.upload(ptr, torch.int32), rt.upload(items, torch.int32)

    def set_pos_mode(self, forward_only):
        self._pos_mode = 'eval' if forward_only else 'train'      # :727

    # ----------------------------------------------------------------- feeds
    def target_mapping(self, item_target):
        """embed_attribute.py:679-684: item index -> logit index (host lists)."""
        m = self._item2logit_np
        out = []
        for items in item_target:
            a = np.asarray(items, dtype=np.int64)
            r = m[a]
            if (r < 0).any():
                raise KeyError(int(a[np.argmax(r < 0)]))
            out.append(r.tolist())
        return out

    def target_mapping_device(self, item_ids_dev, out_dev):
        """device twin of target_mapping: out[i] = item_ind2logit_ind[item[i]]."""
        self.rt.flush_feeds()
        ops.sparse_site_onehot(self._item2logit_dev, item_ids_dev, 0, 0.0, out_dev, None, None)
        return out_dev

    def add_input(self, input_feed, user_input, item_input, neg_item_input=None, item_sampled=None,
                  item_sampled_id2idx=None, forward_only=False, recommend=False, loss=None):
        """embed_attribute.py:697-747.  Feeds the placeholders directly (input_feed is
        kept for signature compatibility) and returns the reference's triple; the
        positive-mask index list the reference builds on the host (:721-745) is
        produced on device from the positives CSR, so input_feed_warp is empty."""
        if self.user_attributes is not None:
            self.u_indices['input'].feed(user_input)
        if self.item_attributes is not None and self.input_steps > 0 and item_input is not None:
            if isinstance(item_input, torch.Tensor):
                self.input_all.feed(item_input.reshape(-1))
            else:
                arr = np.asarray(item_input, dtype=np.int32)      # [steps][mb] time-major
                if arr.shape[0] < self.input_steps:
                    pad = np.zeros((self.input_steps - arr.shape[0], arr.shape[1]), dtype=np.int32)
                    arr = np.concatenate([arr, pad], 0)
                self.input_all.feed(arr.reshape(-1))
        update_sampled = []
        input_feed_sampled = {}
        if (self.item_attributes is not None and recommend is False and item_sampled is not None
                and loss in ['mw', 'mce']):
            input_feed_sampled['item_sampled_ind'] = item_sampled
            update_sampled = [lambda: self.update_sampled_pool(item_sampled)]
        self.set_pos_mode(forward_only)
        return update_sampled, input_feed_sampled, {}

    # ------------------------------------------------------------------ sizes
    def get_user_model_size(self, no_id=False, concat=True):
        ua = self.user_attributes
        if concat:
            s = 1 if no_id else 0
            return (sum(ua._embedding_size_list_cat[s:ua.num_features_cat]) +
                    sum(ua._embedding_size_list_mulhot[0:ua.num_features_mulhot]))
        return ua._embedding_size_list_cat[0]

    def get_item_model_size(self, concat=True):
        ia = self.item_attributes
        if concat:
            return (sum(ia._embedding_size_list_cat[0:ia.num_features_cat]) +
                    sum(ia._embedding_size_list_mulhot[0:ia.num_features_mulhot]))
        return ia._embedding_size_list_cat[0]


def reduce_mean(x, axis=0):
    """tf.reduce_mean over a feature list (hmf_model.py:97, seqModel.py:154)."""
    if isinstance(x, FeatureList):
        return x.mean_node
    raise NotImplementedError("reduce_mean over %r" % type(x))
