"""Row-sharded HMF step over RCCL / xGMI.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The
item embedding table (+ bias, + Adagrad slots) is striped row-wise over the
ranks (owner = item % world); the user table is striped the same way and every
rank trains on interactions of ITS users (pure data parallelism on that side).
One step with B = world * B_loc interactions and a pool of S shared negatives
(S/world owned by each rank) is algebraically the single-process step of
hmf_model.py on the global batch.  Exchanges per step and rank (SURVEY 8e's
"gather the pool rows" alternative -- the [B, S] logits never cross xGMI):

  all_gather      owned pool rows [S/N, d+4] -> pool [S, d+4]        (0.5 MB at S=1024, d=128)
  all_to_all(v)   target item ids  -> their owners                   (4 B per interaction; input routing,
                                                                      done with the batch in prepare_route())
  all_to_all(v)   target rows [*, d+4] back                          (B_loc rows, (N-1)/N cross)
  (local)         logits = U_loc . pool^T + b, WMRB loss, dU, user-shard Adagrad
  reduce_scatter  pool gradient partials [S, d+4] -> owner blocks    (0.5 MB)
  all_to_all(v)   target-row gradients -> owners                     (B_loc rows)
  (local)         item-shard scatter + Adagrad (pool block + received target gradients)

(column d of the packed rows carries the bias / bias gradient.)  Per rank ~2 x B_loc x (d+4) x 4
bytes move per step instead of 2 x B x S x 4 / N for an all-to-all of the logits (16.5 MB vs
134 MB at B_loc=16384, S=1024).  The variable-size exchanges need per-destination counts: they
are a property of the batch (owner = item % N), computed by the data loader on the host
(`prepare_route`), which also orders the batch by owner so the device never permutes rows.
No table gradient ever crosses xGMI and there is no ring all-reduce on the path (the dense
LSTM weights of the sequence model would use one, 128 KB, latency-bound).

The compute stages go through a `backend` object; the product backend is
HipBackend (libarx.so).  tests/ injects a numpy backend to check the sharded
algorithm against the single-process oracle with gloo on CPU -- the package
itself contains no CPU implementation.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist


class _Done(object):
    def wait(self):
        return True


def _all_to_all(out, inp, out_splits=None, in_splits=None, group=None, async_op=False):
    """dist.all_to_all_single -- RCCL in production.  Test rigs run several ranks on ONE GPU over gloo,
    which has no all-to-all for device tensors: there the blocks are staged through the host."""
    if inp.is_cuda and dist.get_backend(group) == 'gloo':
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), output_split_sizes=out_splits,
                               input_split_sizes=in_splits, group=group)
        out.copy_(o)
        return _Done()
    w = dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group,
                               async_op=async_op)
    return w if async_op else _Done()


def _reduce_scatter(out, inp, group=None):
    """dist.reduce_scatter_tensor (sum) -- RCCL in production; device tensors over gloo (several ranks on ONE GPU:
    the test rigs) are summed through the host and every rank keeps its block."""
    if inp.is_cuda and dist.get_backend(group) == 'gloo':
        t = inp.cpu().contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        n = out.shape[0]
        r = dist.get_rank(group)
        out.copy_(t[r * n:(r + 1) * n])
        return
    dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)


class HipBackend(object):
    """Compute stages on libarx.so (include/arx.h)."""

    def __init__(self, device):
        from . import ops
        self.ops = ops
        self.ws = ops.Workspace(device)
        self.ws_k7 = ops.Workspace(device)     # K7's own: its sort half may run ahead, under the GEMMs (which use ws)

    def gather_rows(self, E, bias, rows, out, bias_out, scale=1.0):
        self.ops.gather_onehot(E, bias, None, rows, out, scale=scale, bias_out=bias_out)

    def gather_bags(self, E, bias, vals, starts, lens, ids, out, bias_out, scale=1.0, accumulate=False):
        """out[r] (+)= scale * mean over the bag of ids[r] of E rows; bias_out likewise."""
        self.ops.gather_mulhot_mean(E, bias, vals, starts, lens, ids, out, scale=scale, accumulate=accumulate,
                                    bias_out=bias_out)

    def bags_adagrad(self, E, acc, bias, bias_acc, vals, starts, lens, sites, G, Gb, lr, phase=3):
        """Two-stage multi-hot pass (arx_sparse_adagrad_bags); sites: [(entity ids, row_base, coef)].
        Tokens >= E.shape[0] (rows of other shards, mapped to the padding row) are dropped.
        phase 1: both sorts (ids only), 2: merge + apply, 3: both."""
        ops = self.ops
        # one workspace per SHAPE of the pass (counts, row bases, coefficients); the id tensors may be
        # fresh every step (prepare_route): only the pointer arrays are rebuilt then
        key = tuple((int(i.shape[0]), b, c) for i, b, c in sites)
        ptrs = tuple(i.data_ptr() for i, _, _ in sites)
        cache = self.__dict__.setdefault('_bags', {})
        ent = cache.get(key)
        mx = self.__dict__.setdefault('_bag_maxlen', {})
        if lens.data_ptr() not in mx:
            mx[lens.data_ptr()] = (lens, int(lens.max().item()))      # (the tensor is kept: its address stays its own)
        if ent is None:
            ent = cache[key] = [None, None, ops.Workspace(G.device)]
        if ent[0] != ptrs:
            ent[0], ent[1] = ptrs, ops.BagSiteArgs(sites, mx[lens.data_ptr()][1])
        ops.sparse_adagrad_bags(E, acc, bias, bias_acc, vals, starts, lens, ent[1], G, Gb, lr, ent[2], phase=phase)

    def lookup_het_multi(self, sites):
        """The step's lookups in ONE launch (arx_lookup_multi) where items are HET (id row + bag mean, both halved):
        sites = LookupSet tuples (E_id, bias_id, cat_map, E_tok, bias_tok, vals, starts, lens, ids, out, scale,
        bias_out); static buffers: the descriptor is built once per set of addresses."""
        key = tuple((x[8].data_ptr(), int(x[8].shape[0]), x[9].data_ptr()) for x in sites)
        cache = self.__dict__.setdefault('_lsets', {})
        ls = cache.get(key)
        if ls is None:
            if len(cache) > 16:
                cache.clear()
            ls = cache[key] = self.ops.LookupSet(sites)
        self.ops.lookup_multi(ls)

    def gather_rows_packed(self, E, bias, rows, out):
        self.ops.gather_onehot_packed(E, bias, None, rows, out)

    def gather_rows_multi(self, sites):
        """Several lookups of equal width in one launch; sites: [(E, bias|None|column, rows, out, bias_out)],
        bias an int: that column of E's (packed) rows; bias_out None | a vector | 'packed' (column d of the
        out rows).  Static buffers: the descriptor
        is built once per set of addresses."""
        key = tuple((E.data_ptr(), bi if isinstance(bi, int) else -1, r.data_ptr(), int(r.shape[0]), o.data_ptr(),
                     b if isinstance(b, str) else (b.data_ptr() if b is not None else 0)) for E, bi, r, o, b in sites)
        cache = self.__dict__.setdefault('_gsets', {})
        gs = cache.get(key)
        if gs is None:
            if len(cache) > 16:
                cache.clear()
            gs = cache[key] = self.ops.GatherSet([(E, bias, None, r, o, 1.0, b) for E, bias, r, o, b in sites])
        self.ops.gather_onehot_multi(gs)

    def gemm(self, A, B, C, transA=False, transB=False, beta=0.0, col_bias=None, a_rowsum=None):
        self.ops.gemm(A, B, C, self.ws, transA=transA, transB=transB, beta=beta, col_bias=col_bias,
                      a_rowsum=a_rowsum)

    def dot_score(self, U, T, tb, out):
        self.ops.dot_score(U, T, tb, out)

    def dot_score_bwd(self, U, T, ds, dU, acc, dT):
        self.ops.dot_score_bwd(U, T, ds, dU, acc, dT)

    def copy_2d(self, src, dst):
        self.ops.copy_2d(src, dst)

    def copy_strided(self, src, dst):
        self.ops.copy_strided(src, dst)

    def transpose(self, src, dst):
        self.ops.transpose(src, dst)

    def gather_rows_wide(self, src, rows, dst):
        self.ops.gather_rows_wide(src, rows, dst)

    def add_2d(self, src, dst):
        """dst += src (2-D, any row strides)."""
        self.ops.add_rows_bcast(1.0, src, 1.0, dst)

    def shard_route(self, ids, world, rank, zero_row, rows_out, keys_out):
        self.ops.shard_route(ids, world, rank, zero_row, rows_out, keys_out)

    def pool_blocks(self, ids, world, rank, zero_row, cap, counts, gidx=None, my_slots=None, pool_rows=None):
        self.ops.pool_blocks(ids, world, rank, zero_row, cap, counts, gidx, my_slots, pool_rows)

    def loss_mw_pos(self, logits, t, urows, ptr, items, i2s, bl, dl, dt, gscale):
        self.ops.loss_mw_pos(logits, t, urows, ptr, items, i2s, bl, dl, dt, gscale)

    def loss_mw_fused_pos(self, logits, U, T, tb, urows, ptr, items, i2s, bl, dl, t_out, dt, dU, dT, gscale):
        """loss + target score + rank-one gradients in one kernel; tb / dt may be strided columns."""
        self.ops.loss_mw_fused_pos(logits, U, T, tb, urows, ptr, items, i2s, bl, dl, t_out, dt, dU, dT, gscale)

    def sum_scaled(self, x, scale, out):
        self.ops.sum_scaled(x, scale, out)

    def sparse_adagrad(self, E, acc, bias, bias_acc, keys, G, Gb, lr):
        self.ops.sparse_adagrad(E, acc, bias, bias_acc, keys, None, None, G, Gb, lr, self.ws)

    def sparse_adagrad_multi(self, tables, sites, G, Gb, lr, phase=3):
        """tables: [(E, acc, bias|None, bias_acc|None)]; sites: [(table, local_rows, row_base)]:
        one fused pass (arx_sparse_adagrad_cat_multi).  phase 1: keys + sorts + run records (needs the
        ids only), 2: apply, 3: both."""
        ops = self.ops
        sites = [(x[0], x[1], x[2], x[3] if len(x) > 3 else 1.0) for x in sites]   # (table, rows, base[, coef])
        key = tuple((t, r.data_ptr(), int(r.shape[0]), b, c) for t, r, b, c in sites)
        cache = self.__dict__.setdefault('_multi', {})
        ent = cache.get(key)
        if ent is None:
            if len(cache) > 64:
                cache.clear()
            dev = G.device
            cnts = self.__dict__.setdefault('_cnt', {})
            tabs = []
            for E, acc, bias, bacc in tables:
                ck = (E.data_ptr(), int(E.shape[0]))      # (E_item[:ni] and E_item share a pointer, not a row count)
                c = cnts.get(ck)
                if c is None:
                    c = cnts[ck] = torch.zeros((E.shape[0],), dtype=torch.int32, device=dev)
                tabs.append((E, acc, bias, bacc, c))
            args = ops.MultiCatArgs(tabs, [(t, None, r, b, c) for t, r, b, c in sites])
            n = max(args.total, 1)
            bufs = self.__dict__.get('_multi_bufs')
            if bufs is None or bufs[0].shape[0] < n:
                bufs = self._multi_bufs = (torch.empty(n, dtype=torch.int32, device=dev),
                                           torch.empty(n, dtype=torch.int32, device=dev),
                                           torch.empty(n, dtype=torch.float32, device=dev))
            ent = cache[key] = args
        kb, sb, cb = self._multi_bufs
        if kb.shape[0] < ent.total:
            self._multi_bufs = kb, sb, cb = (torch.empty(ent.total, dtype=torch.int32, device=G.device),
                                             torch.empty(ent.total, dtype=torch.int32, device=G.device),
                                             torch.empty(ent.total, dtype=torch.float32, device=G.device))
        ops.sparse_adagrad_cat_multi(ent, G, Gb, lr, kb, sb, cb, self.ws_k7, phase=phase)

    def bags_grad_dense(self, D, Db, vals, starts, lens, sites, G, Gb, phase=3):
        """D[t] += sum of the gradient rows of the bags that hold token t (coefficient coef / len), Db likewise: the
        two-stage multi-hot pass in its gradient-descent form (no slots) with a step of -1 onto a gradient table --
        0 - (-1) g = g, the merged sums themselves, bit for bit.  Tokens >= D.shape[0] are dropped."""
        if getattr(self, '_neg_one', None) is None:
            self._neg_one = torch.tensor([-1.0], dtype=torch.float32, device=D.device)
        self.bags_adagrad(D, None, Db, None, vals, starts, lens, sites, G, Gb, self._neg_one, phase=phase)

    def adagrad_dense(self, w, acc, g, lr):
        self.ops.adagrad_dense(w, acc, g, lr)

    def adagrad_rows_nonzero(self, W, acc, bias, bias_acc, G, Gb, lr):
        """Adagrad on the rows whose dense gradient row is not all zero; the consumed gradient is zeroed (arx.h)."""
        self.ops.adagrad_rows_nonzero(W, acc, bias, bias_acc, G, Gb, lr)

    def fill_zero(self, t):
        self.ops.fill_f32(t, 0.0)

    def take_i32(self, table, idx, out, fill):
        self.ops.take_i32(table, idx, out, fill=fill)

    def slot_map_set(self, m, ids, clear):
        self.ops.slot_map_set(m, ids, clear=clear)

    def attach_pool_bitmap(self, m, bits):
        self.ops.slot_map_attach_bitmap(m, bits)

    def copy_i32(self, src, dst):
        dst.copy_(src, non_blocking=True)


class ShardedHMF(object):
    """id-only HMF ('mw' loss) with row-sharded tables.  Global ids everywhere in
    the API; `users` passed to step() must all be owned by this rank."""

    def __init__(self, n_users, n_items, d, B_loc, S, learning_rate, rank, world, device,
                 backend=None, group=None, tables=None, seed=0, acc0=0.1, graphs=None, exchange='rows'):
        if S % 4 != 0 or d % 4 != 0:
            raise ValueError("n_sampled and d must be multiples of 4")
        if exchange not in ('rows', 'logits'):
            raise ValueError("exchange: 'rows' (gather the pool rows) or 'logits' (all-to-all of the logits)")
        if exchange == 'logits' and type(self) is not ShardedHMF:
            raise ValueError("exchange='logits' is the id-only step's alternative (ShardedHMF)")
        # 'logits' (the exchange north_star words): the latents are all-gathered, every owner
        # scores the WHOLE batch against its pool block, the [B, S_g] partial logits cross by all_to_all, and their
        # gradients cross back -- _step_logits.  Eager launches (no graph segments).
        self.exchange = exchange
        if exchange == 'logits':
            graphs = False
        self.n_users, self.n_items, self.d = n_users, n_items, d
        # The shared pool is ONE draw over all items (prepare_train.py:7-17), so the number of pool items
        # a rank owns varies from draw to draw: the owned blocks travel padded to `cap` rows (the largest
        # owner count of the current pool, a multiple of 4; set_pool).  Sg = capacity of a block.
        self.B_loc, self.S, self.Sg = B_loc, S, S
        self.rank, self.world = rank, world
        self.B = B_loc * world
        self.device = torch.device(device)
        self.group = group
        self.be = backend if backend is not None else HipBackend(self.device)
        # hipGraph segments (_step_static): the product backend on a GPU, unless switched off
        if graphs is None:
            graphs = not os.environ.get("ARX_DIST_EAGER")
        self.use_graphs = bool(graphs) and getattr(self, '_static_step_ok', type(self) is ShardedHMF) \
            and isinstance(self.be, HipBackend) and self.device.type == 'cuda'
        self._graphs, self._graph_key, self._warm_key, self.g_idx = {}, None, None, None
        self.n_captures, self.n_replays = 0, 0
        # (the legacy default stream cannot be captured: the step runs on a stream of its own, joined with
        # the caller's stream on both sides)
        self._stream = torch.cuda.Stream(device=self.device) if self.use_graphs else None
        self._side = torch.cuda.Stream(device=self.device) if self.use_graphs else None
        dev = self.device
        f32, i32 = torch.float32, torch.int32
        nu = (n_users - rank + world - 1) // world        # owned rows
        ni = (n_items - rank + world - 1) // world
        self.nu_loc, self.ni_loc = nu, ni
        self.zero_row = ni                                 # padding row of the item shard
        if tables is not None:                             # explicit global tables (tests)
            U = np.asarray(tables['user'], dtype=np.float32)[rank::world]
            I = np.asarray(tables['item'], dtype=np.float32)[rank::world]
            bI = np.asarray(tables['item_bias'], dtype=np.float32).reshape(-1)[rank::world]
            self.E_user = torch.from_numpy(np.ascontiguousarray(U)).to(dev)
            self.E_item = torch.zeros((ni + 1, d), dtype=f32, device=dev)
            self.E_item[:ni].copy_(torch.from_numpy(np.ascontiguousarray(I)))
            self.b_item = torch.zeros((ni + 1,), dtype=f32, device=dev)
            self.b_item[:ni].copy_(torch.from_numpy(np.ascontiguousarray(bI)))
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(seed * 1009 + rank)
            lim_u = float(np.sqrt(6.0 / (n_users + 2 + d)))
            lim_i = float(np.sqrt(6.0 / (n_items + 2 + d)))
            self.E_user = torch.empty((nu, d), dtype=f32, device=dev).uniform_(-lim_u, lim_u, generator=g)
            self.E_item = torch.empty((ni + 1, d), dtype=f32, device=dev).uniform_(-lim_i, lim_i, generator=g)
            self.E_item[ni].zero_()
            self.b_item = torch.empty((ni + 1,), dtype=f32, device=dev).uniform_(-lim_i, lim_i, generator=g)
            self.b_item[ni] = 0.0
        self.A_user = torch.full_like(self.E_user, acc0)
        self.A_item = torch.full_like(self.E_item, acc0)
        self.Ab_item = torch.full_like(self.b_item, acc0)
        self.lr = torch.tensor([float(learning_rate)], dtype=f32, device=dev)

        Sg, W = self.Sg, world
        dp = d + 4                                          # packed row: d values + bias (+ pad)
        self.dp = dp
        z = lambda *s: torch.zeros(s, dtype=f32, device=dev)
        zi = lambda *s: torch.zeros(s, dtype=i32, device=dev)
        self.urows = zi(B_loc)
        self.U_loc = z(B_loc, d)
        self.pool_ids = zi(S)                               # owner-major global ids
        self.pool_rows = zi(Sg)                             # local rows of the owned block (padding row behind it)
        self.cap = S if world == 1 else 0                   # rows of a block in the current exchange
        self.gidx = zi(S)                                   # pool slot -> row of the gathered blocks
        self.my_slots = zi(S)                               # block row -> pool slot (S: none, a zero row)
        self.I_gath = z(world * S, dp) if world > 1 else None
        self.pool_old = None
        self.item2slot = torch.full((n_items + 1,), -1, dtype=i32, device=dev)
        if hasattr(self.be, 'attach_pool_bitmap'):          # (1 bit per item in front of the map: HIP backend)
            self._pool_bits = torch.zeros((n_items + 1 + 32) // 32, dtype=i32, device=dev)
            self.be.attach_pool_bitmap(self.item2slot, self._pool_bits)
        self.I_pack, self.b_g = z(Sg, dp), z(Sg)            # owned pool rows | bias in column d
        self.I_all, self.b_all = z(S, dp), z(S)             # gathered pool
        self.logits = z(B_loc, S)
        self.T_pack, self.tb = z(B_loc, dp), z(B_loc)       # target rows as received back
        self.t_loc, self.dt_loc = z(B_loc), z(B_loc)
        self.bl, self.loss = z(B_loc), z(1)
        self.dlogits = z(B_loc, S)
        self.dI_all, self.gb_all = z(S + 1, dp), z(S)       # pool-gradient partials (all columns) + a zero row
        self.dT_pack = z(B_loc, dp)                         # target-row gradients to send
        self.cap_r = 0                                      # capacity for received target rows
        self._alloc_recv(B_loc)
        self.pos_ptr = zi(nu + 2)
        self.pos_items = zi(1)
        self.steps = 0

    def _alloc_recv(self, cap):
        """Buffers that scale with R = target rows this rank owns in a batch.  The gradient
        arena holds [dU_loc (B_loc) ; dI_g (Sg) ; received dT (R)] so that both tables are
        updated by ONE fused sparse-Adagrad pass."""
        if cap <= self.cap_r:
            return
        dev, f32, i32 = self.device, torch.float32, torch.int32
        B_loc, Sg, dp = self.B_loc, self.Sg, self.dp
        if self.use_graphs and self.cap_r > 0:
            cap = (cap + cap // 8 + 63) // 64 * 64        # new buffers = new graphs: grow with slack
        self.cap_r = cap
        self.recv_ids = torch.zeros((cap,), dtype=i32, device=dev)
        self.recv_rows = torch.zeros((cap,), dtype=i32, device=dev)
        self.T_send = torch.zeros((cap, dp), dtype=f32, device=dev)
        self.tb_send = torch.zeros((cap,), dtype=f32, device=dev)
        self.arena = torch.zeros((B_loc + Sg + cap, dp), dtype=f32, device=dev)
        self.arena_b = torch.zeros((B_loc + Sg + cap,), dtype=f32, device=dev)

    # ------------------------------------------------------------------ state
    def set_positives(self, ptr_local, items_global):
        """CSR over this rank's LOCAL user rows; item ids are global."""
        dev = self.device
        self.pos_ptr = torch.as_tensor(np.asarray(ptr_local, dtype=np.int32)).to(dev) \
            if not isinstance(ptr_local, torch.Tensor) else ptr_local.to(dev, torch.int32)
        it = items_global if isinstance(items_global, torch.Tensor) else \
            torch.as_tensor(np.asarray(items_global, dtype=np.int32))
        self.pos_items = it.to(dev, torch.int32)

    def set_pool(self, pool_ids):
        """The shared negative pool, global item ids in slot order -- any owners (one draw over all
        items: embed_attribute.py:320-348 update_sampled, sharded).  Every rank sees the same ids and
        derives the same block layout: owner g's pool items, in slot order, are rows [0, count_g) of
        its block; all blocks travel padded to cap = max_g count_g rows."""
        be = self.be
        new = pool_ids if isinstance(pool_ids, torch.Tensor) else \
            torch.as_tensor(np.asarray(pool_ids, dtype=np.int32))
        new = new.to(self.device, torch.int32)
        if self.pool_old is not None:
            be.slot_map_set(self.item2slot, self.pool_old, True)
        else:
            self.pool_old = torch.empty_like(self.pool_ids)
        be.copy_i32(new, self.pool_ids)
        be.copy_i32(new, self.pool_old)
        be.slot_map_set(self.item2slot, self.pool_ids, False)
        W, r, S = self.world, self.rank, self.S
        if W == 1:
            be.shard_route(self.pool_ids, W, r, self.zero_row, self.pool_rows, None)
            if self.exchange == 'logits':
                self._logits_layout(np.arange(S, dtype=np.int32))
            return
        # block layout (redraw path, every n_resample steps): two launches of arx_pool_blocks around ONE host
        # read of the W owner counts (the block capacity is a host decision: it sizes the exchanges)
        if getattr(self, '_pool_counts', None) is None:
            self._pool_counts = torch.zeros(W + 1, dtype=torch.int32, device=self.device)   # [W] = negative ids
        be.pool_blocks(self.pool_ids, W, r, self.zero_row, 0, self._pool_counts)
        cnts = self._pool_counts.cpu().tolist()
        if cnts[W] != 0:
            raise ValueError("set_pool: %d negative item id(s) in the pool (a short draw of the device sampler "
                             "leaves -1: redraw or pass valid ids)" % cnts[W])
        cap = (max(cnts[:W]) + 3) // 4 * 4
        if self.use_graphs:      # block capacity only grows (a new capacity = new graphs), with slack
            cap = self.cap if cap <= self.cap else min(S, (cap + cap // 8 + 15) // 16 * 16)
        self.cap = cap
        be.pool_blocks(self.pool_ids, W, r, self.zero_row, cap, self._pool_counts, self.gidx, self.my_slots,
                       self.pool_rows)
        if self.exchange == 'logits':
            self._logits_layout(self.gidx.cpu().numpy())

    def _logits_layout(self, gidx):
        """Buffers and the inverse block map of the logits exchange, per pool draw (host; the redraw path):
        blk2slot[g * cap + j] = pool slot of row j of owner g's block (S = padding: a zero row), the order
        in which the gradient rows of the transposed logits are sent back to the owners."""
        W, S, cap, B_loc, dp = self.world, self.S, self.cap, self.B_loc, self.dp
        inv = np.full(W * cap, S, dtype=np.int32)
        inv[np.asarray(gidx, dtype=np.int64)] = np.arange(S, dtype=np.int32)
        self.blk2slot = torch.from_numpy(inv).to(self.device)
        if W == 1:
            self.gidx = torch.arange(S, dtype=torch.int32, device=self.device)
        if getattr(self, '_lg_cap', -1) != cap:
            z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=self.device)
            self._lg_cap = cap
            self.U_aug = z(B_loc, dp)                  # [U | 1 | 0 0 0]: the bias column of the packed pool rows
            self.U_aug[:, self.d] = 1.0                # rides through the scorer GEMM
            self.U_all = z(W * B_loc, dp)
            self.Pt_send, self.Pt_recv = z(W * cap, B_loc), z(W * cap, B_loc)    # [owner | dest][block row][batch row]
            self.logitsT, self.dlogitsT = z(S, B_loc), z(S + 1, B_loc)           # (+ a zero row for the padding)
            self.dPt_send, self.dPt_recv = z(W * cap, B_loc), z(W * cap, B_loc)
            self.dI_blk = z(cap, dp)
            self.dU_part = z(W * B_loc, self.d)
            self.dU_red = z(B_loc, self.d)

    # ------------------------------------------------------------------ route
    def prepare_route(self, users, items):
        """Data-loader side of a batch (host): orders the interactions by the owner of their
        target item and returns what the variable-size exchanges need.  One tiny all_to_all
        of the per-destination counts (not on the step path)."""
        W = self.world
        u = users.cpu().numpy() if isinstance(users, torch.Tensor) else np.asarray(users)
        it = items.cpu().numpy() if isinstance(items, torch.Tensor) else np.asarray(items)
        u = u.astype(np.int32)
        it = it.astype(np.int32)
        owner = it % W
        perm = np.argsort(owner, kind='stable')
        send = np.bincount(owner, minlength=W).astype(np.int64)
        st = torch.from_numpy(send).to(self.device)
        rt = torch.empty_like(st)
        _all_to_all(rt, st, group=self.group)
        recv = [int(v) for v in rt.cpu().tolist()]
        route = {'users': torch.from_numpy(np.ascontiguousarray(u[perm])).to(self.device),
                 'items': torch.from_numpy(np.ascontiguousarray(it[perm])).to(self.device),
                 'send': [int(v) for v in send.tolist()], 'recv': recv, 'R': int(sum(recv))}
        self._alloc_recv(route['R'])
        # The ids themselves are routed here too (they are input data, like the owner ordering
        # above): every owner learns which of its rows this batch asks for, and the local row of
        # every id is resolved once.  The step path then starts at the table lookups.
        R = route['R']
        recv_ids = torch.zeros((R,), dtype=torch.int32, device=self.device)
        _all_to_all(recv_ids, route['items'], recv, route['send'], group=self.group)
        recv_rows = torch.zeros((R,), dtype=torch.int32, device=self.device)
        if R > 0:
            self.be.shard_route(recv_ids, W, self.rank, self.zero_row, recv_rows, None)
        urows = torch.zeros((self.B_loc,), dtype=torch.int32, device=self.device)
        self.be.shard_route(route['users'], W, self.rank, 0, urows, None)   # all owned: local rows
        route.update(recv_ids=recv_ids, recv_rows=recv_rows, urows=urows)
        return route

    # ------------------------------------------------------------------- step
    def step(self, users, items=None):
        """One training step.  `users` is either a route from prepare_route() (the fast way:
        nothing but kernels and collectives on the step path) or a users array with `items`
        (routed here on the host)."""
        if self.world > 1 and self.cap <= 0:
            raise RuntimeError("ShardedHMF.step before set_pool(): the pool's block layout sizes the exchanges")
        route = users if isinstance(users, dict) else self.prepare_route(users, items)
        if self.exchange == 'logits':
            return self._step_logits(route)
        if self.use_graphs:
            outer = torch.cuda.current_stream(self.device)
            if outer == self._stream:          # the caller already works on the model's stream (model.stream)
                self._step_guard(route)
                return
            self._stream.wait_stream(outer)
            with torch.cuda.stream(self._stream):
                self._step_guard(route)
            outer.wait_stream(self._stream)
            return
        be, W, r = self.be, self.world, self.rank
        B, B_loc, S, Sg, d, dp = self.B, self.B_loc, self.S, self.Sg, self.d, self.dp
        grp = self.group
        send, recv, R = route['send'], route['recv'], route['R']
        arena, arena_b = self.arena, self.arena_b
        # The two large exchanges (target rows out, target-row gradients back: B_loc x (d+4) floats
        # each) are issued asynchronously and waited for only where their result is needed, so that
        # they travel under the scorer GEMM and under the two backward GEMMs respectively.
        # ---- forward ----
        urows, recv_rows = route['urows'], route['recv_rows']
        self.urows = urows
        be.gather_rows(self.E_user, None, urows, self.U_loc, None)
        cap = self.cap
        if W == 1:
            be.gather_rows_packed(self.E_item, self.b_item, self.pool_rows, self.I_all)   # row | bias
        else:
            be.gather_rows_packed(self.E_item, self.b_item, self.pool_rows[:cap], self.I_pack[:cap])
            dist.all_gather_into_tensor(self.I_gath[:W * cap], self.I_pack[:cap], group=grp)
            be.gather_rows(self.I_gath, None, self.gidx, self.I_all, None)    # blocks -> pool (slot) order
        be.copy_strided(self.I_all[:, d], self.b_all)
        T_send = self.T_send[:R]
        if R > 0:
            be.gather_rows_packed(self.E_item, self.b_item, recv_rows, T_send)
        w_rows = _all_to_all(self.T_pack, T_send, send, recv, group=grp, async_op=True)   # packed target rows back ...
        be.gemm(self.U_loc, self.I_all[:, :d], self.logits, transB=True, col_bias=self.b_all)   # ... under the scorer
        w_rows.wait()
        # loss (global mean => gscale = 1/B) with the target score and its rank-one gradients formed
        # by the same kernel, straight from / into the packed rows (bias and dt live in column d)
        dU = arena[:B_loc, :d]
        be.loss_mw_fused_pos(self.logits, self.U_loc, self.T_pack[:, :d], self.T_pack[:, d], self.urows,
                             self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.dlogits,
                             self.t_loc, self.dT_pack[:, d], dU, self.dT_pack[:, :d], 1.0 / B)
        # ---- backward ----
        w_dt = _all_to_all(arena[B_loc + Sg:B_loc + Sg + R], self.dT_pack, recv, send, group=grp,
                           async_op=True)                             # target-row gradients -> owners ...
        be.gemm(self.dlogits, self.I_all[:, :d], dU, beta=1.0)                # ... under dU += dL . pool
        # pool gradient partials (+ bias gradient = row sums) -> owners
        be.gemm(self.dlogits, self.U_loc, self.dI_all[:S, :d], transA=True, a_rowsum=self.gb_all)
        be.copy_strided(self.gb_all, self.dI_all[:S, d])
        if W == 1:
            be.copy_2d(self.dI_all[:S], arena[B_loc:B_loc + S])
        else:
            # 0.5 MB: summed everywhere, every owner picks the rows of its block (padding -> the zero row)
            dist.all_reduce(self.dI_all[:S], op=dist.ReduceOp.SUM, group=grp)
            be.gather_rows(self.dI_all, None, self.my_slots[:cap], arena[B_loc:B_loc + cap], None)
        w_dt.wait()
        be.copy_strided(arena[B_loc:B_loc + Sg + R, d], arena_b[B_loc:B_loc + Sg + R])
        # one fused scatter + Adagrad pass over both shards
        sites = [(0, self.urows, 0), (1, self.pool_rows[:cap], B_loc)]
        if R > 0:
            sites.append((1, recv_rows, B_loc + Sg))
        be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                 (self.E_item, self.A_item, self.b_item, self.Ab_item)],
                                sites, arena[:, :d], arena_b, self.lr)
        self.steps += 1


    def _step_logits(self, route):
        """The step with the exchange north_star words (hmf_model.py:52-99 on the global
        batch, embed_attribute.py:148-206 scorer, :641-649 'mw'): the [B, S] logits are computed WHERE THE POOL
        ROWS LIVE and cross xGMI, instead of the pool rows travelling to the batch rows.

          all_gather      latents [B_loc, d+4] -> [B, d+4]                      (column d = 1: carries the pool bias)
          (local)         Pt[h] = I_g . U_h^T  [cap, B_loc] for every destination h (owned pool block, padded to cap)
          all_to_all      Pt blocks -> rank h holds [W . cap, B_loc] = its rows against every owner's block
          (local)         block rows -> pool slots (gather), transpose -> logits [B_loc, S]; target rows as in step();
                          WMRB loss, dlogits; transpose, slots -> block rows (gather; padding reads a zero row)
          all_to_all      dlogits^T blocks back -> owner g holds dP^T [W][cap, B_loc]
          (local)         dI_g = sum_h dP_h^T-blocks . U_h  (column d: the bias gradient), dU partials for ALL B rows
          reduce_scatter  dU partials [B, d] -> [B_loc, d]
          (local)         the same fused scatter + Adagrad pass as step()

        Bytes per rank and step: 2 . (W-1)/W . B_loc . W . cap . 4 for the logits (cap ~ S / W: ~2 . B_loc . S . 4)
        + 2 . B . (d+4) . 4 for the latents, against 2 . B_loc . (d+4) . 4 + ~2 . S . (d+4) . 4 for the 'rows'
        exchange: it pays when S . (d+4) > B_loc . S, i.e. B_loc < d + 4 -- small batches against huge pools.  The
        product default stays 'rows'; this form is kept measured and tested (tests/test_dist_cpu.py, test_dist_gpu.py)."""
        be, W = self.be, self.world
        B, B_loc, S, Sg, d, dp, cap = self.B, self.B_loc, self.S, self.Sg, self.d, self.dp, self.cap
        grp = self.group
        send, recv, R = route['send'], route['recv'], route['R']
        arena, arena_b = self.arena, self.arena_b
        urows, recv_rows = route['urows'], route['recv_rows']
        self.urows = urows
        # ---- forward ----
        be.gather_rows(self.E_user, None, urows, self.U_aug[:, :d], None)
        be.copy_2d(self.U_aug[:, :d], self.U_loc)
        if W == 1:
            be.copy_2d(self.U_aug, self.U_all)
        else:
            dist.all_gather_into_tensor(self.U_all, self.U_aug, group=grp)
        be.gather_rows_packed(self.E_item, self.b_item, self.pool_rows[:cap], self.I_pack[:cap])
        T_send = self.T_send[:R]
        if R > 0:
            be.gather_rows_packed(self.E_item, self.b_item, recv_rows, T_send)
        w_rows = _all_to_all(self.T_pack, T_send, send, recv, group=grp, async_op=True)   # target rows back ...
        for h in range(W):                                                                 # ... under the scorer
            be.gemm(self.I_pack[:cap], self.U_all[h * B_loc:(h + 1) * B_loc], self.Pt_send[h * cap:(h + 1) * cap],
                    transB=True)
        if W == 1:
            Pt = self.Pt_send
        else:
            _all_to_all(self.Pt_recv, self.Pt_send, group=grp)
            Pt = self.Pt_recv
        be.gather_rows_wide(Pt, self.gidx, self.logitsT)                 # block rows -> pool slots
        be.transpose(self.logitsT, self.logits)
        w_rows.wait()
        dU = arena[:B_loc, :d]
        be.loss_mw_fused_pos(self.logits, self.U_loc, self.T_pack[:, :d], self.T_pack[:, d], self.urows,
                             self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.dlogits,
                             self.t_loc, self.dT_pack[:, d], dU, self.dT_pack[:, :d], 1.0 / B)
        # ---- backward ----
        w_dt = _all_to_all(arena[B_loc + Sg:B_loc + Sg + R], self.dT_pack, recv, send, group=grp, async_op=True)
        be.transpose(self.dlogits, self.dlogitsT[:S])
        be.gather_rows_wide(self.dlogitsT, self.blk2slot, self.dPt_send)  # slots -> block rows of every owner
        if W == 1:
            dPt = self.dPt_send
        else:
            _all_to_all(self.dPt_recv, self.dPt_send, group=grp)
            dPt = self.dPt_recv
        for h in range(W):
            blk, Uh = dPt[h * cap:(h + 1) * cap], self.U_all[h * B_loc:(h + 1) * B_loc]
            be.gemm(blk, Uh, self.dI_blk, beta=0.0 if h == 0 else 1.0)           # [cap, d+4]: column d = bias gradient
            be.gemm(blk, self.I_pack[:cap, :d], self.dU_part[h * B_loc:(h + 1) * B_loc], transA=True)
        if W == 1:
            be.copy_2d(self.dU_part, self.dU_red)
        else:
            _reduce_scatter(self.dU_red, self.dU_part, group=grp)
        be.add_2d(self.dU_red, dU)                                                # dU = dt . T (loss kernel) + dL . pool
        be.copy_2d(self.dI_blk, arena[B_loc:B_loc + cap])
        w_dt.wait()
        be.copy_strided(arena[B_loc:B_loc + Sg + R, d], arena_b[B_loc:B_loc + Sg + R])
        sites = [(0, self.urows, 0), (1, self.pool_rows[:cap], B_loc)]
        if R > 0:
            sites.append((1, recv_rows, B_loc + Sg))
        be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                 (self.E_item, self.A_item, self.b_item, self.Ab_item)],
                                sites, arena[:, :d], arena_b, self.lr)
        self.steps += 1

    def _fused_scorer(self):
        """True when the step takes the bf16-pipe scorer (switches on, shapes it supports); allocates its buffers."""
        ops_ = getattr(self.be, 'ops', None)          # (the numpy test double has no kernels to pick from)
        if ops_ is None or not ops_.mw_scorer_supported(self.B_loc, self.S, self.d):
            return False
        if getattr(self, 'scorer', None) is None:
            self.scorer = ops_.MwScorer(self.B_loc, self.S, self.d, self.device)
        return True

    @property
    def stream(self):
        """The stream the graph-segment step runs on (None: eager step, the caller's stream).  A training loop
        that makes it the current stream (`with torch.cuda.stream(model.stream)`) saves the two stream
        joins per step that a caller on another stream pays."""
        return self._stream

    # ------------------------------------------------------- step, hipGraph segments
    def _segment(self, mode, name, fn, feeds=None):
        """eager: run; capture: record the launches of `fn` into a hipGraph, keep it, launch it;
        replay: launch the kept graph.  feeds ([(src, dst)], the step's first segment): the copy of the batch's
        index vector is the graph's first node, its source replaced before every replay
        (ops.CapturedGraph.set_feeds) -- it was an eager launch in front of the graph."""
        ops_ = self.be.ops
        if mode == 'eager':
            if feeds:
                ops_.copy_words(feeds)
            fn()
        elif mode == 'capture':
            g = ops_.CapturedGraph()
            g.begin()
            try:
                if feeds:
                    ops_.copy_words(feeds)
                fn()
            except BaseException:
                g.end()
                raise
            g.end(feeds=feeds)
            self._graphs[name] = g
            g.launch()
        else:
            g = self._graphs[name]
            if feeds:
                if g.feeds_match(feeds):
                    g.set_feeds(feeds)
                else:
                    ops_.copy_words(feeds)
                    g.set_feeds(None)
            g.launch()

    def _step_guard(self, route):
        """_step_static; a step that raises leaves no captured graph behind: the next one runs eagerly again (and, for
        the replicated-token class, from a cleared token-gradient table -- see _step_static)."""
        try:
            self._step_static(route)
        except BaseException:
            self._graphs, self._graph_key, self._warm_key = {}, None, None
            raise

    def _step_static(self, route):
        """The step of step() with every buffer at a fixed address and a fixed size, so that the
        kernels between two collectives are ONE hipGraph launch each (5 segments + K7's sort half as a
        sixth, on a second stream under the forward kernels + 4 collectives + one index copy per step
        instead of ~35 kernel launches from Python; world 1: ONE graph, the sort half a branch of it).  What varies from batch to batch is the index vector [user rows ; received target
        rows], padded with the shard's padding row to the capacity cap_r (gathers: a zero row;
        K7: the tables are passed without the padding row, so padded keys are out of range and dropped)
        and the split sizes of the two all-to-alls, which stay outside the graphs.  A configuration
        (block capacity, receive capacity) runs eagerly once (module loads, workspaces), is captured
        on its second step and replayed from then on."""
        be, W = self.be, self.world
        B, B_loc, S, Sg, d = self.B, self.B_loc, self.S, self.Sg, self.d
        grp, dev = self.group, self.device
        send, recv, R = route['send'], route['recv'], route['R']
        if R > self.cap_r:
            self._alloc_recv(R)
        cap, cap_r = self.cap, self.cap_r
        n_idx = B_loc + cap_r
        idx = route.get('idx')
        if idx is None or idx.shape[0] != n_idx:
            idx = torch.full((n_idx,), self.zero_row, dtype=torch.int32, device=dev)
            idx[:B_loc] = route['urows']
            if R > 0:
                idx[B_loc:B_loc + R] = route['recv_rows']
            route['idx'] = idx
        if self.g_idx is None or self.g_idx.shape[0] != n_idx:
            self.g_idx = torch.empty(n_idx, dtype=torch.int32, device=dev)
        feed = [(idx, self.g_idx)]                     # (a kernel: a device-to-device hipMemcpyAsync costs more;
        #                                                since round 4 the first node of the step's first graph)
        feed += self._static_feeds(route, cap_r)       # (subclasses: more per-batch index vectors)
        key = (cap, cap_r, self.g_idx.data_ptr(), self.arena.data_ptr(), self.pos_ptr.data_ptr(),
               self.pos_items.data_ptr())
        if os.environ.get("ARX_DIST_NO_CAPTURE"):        # (profiling: the static step, launched kernel by kernel)
            mode = 'eager'
        elif self._graph_key == key:
            mode = 'replay'
        elif self._warm_key == key:
            mode, self._graphs = 'capture', {}
        else:
            mode, self._graphs, self._graph_key = 'eager', {}, None
        seg = lambda name, fn: self._segment(mode, name, fn)
        if mode != 'replay' and getattr(self, 'D_tok', None) is not None:
            # ShardedHMFRepTokens: the dense token-gradient table must be all zero at step entry (the token apply zeroes
            # the rows it consumes).  A step that died between the gradient pass and that apply would leave sums behind
            # that the next step counts twice: every step that is NOT a replay (the first eager one, a re-capture,
            # the step after an exception -- _step_guard below drops the graphs) starts from a cleared table
            # (advisor, round 5).
            self.D_tok.zero_()
            self.Db_tok.zero_()
        arena, arena_b = self.arena, self.arena_b
        urows, rrows = self.g_idx[:B_loc], self.g_idx[B_loc:]
        self.urows = urows
        ni = self.ni_loc
        dU = arena[:B_loc, :d]
        T_in = self.T_pack if W == 1 else self.T_send[:cap_r]              # target rows as gathered
        dT = arena[B_loc + Sg:B_loc + Sg + B_loc] if W == 1 else self.dT_pack

        # world 1: nothing travels, so nothing is packed twice -- the pool bias goes straight to b_all, the pool
        # gradient and its row sums straight into the K7 arena, the target-bias gradient straight into arena_b
        het = getattr(self, '_het', False)             # ShardedHMFRepTokens: the owner forms HET rows, token table replicated

        def fwd_gather():      # the step's three lookups, one launch
            if het:
                return self._het_gather(urows, T_in, cap, cap_r)
            if W == 1:
                be.gather_rows_multi([(self.E_user, None, urows, self.U_loc, None),
                                      (self.E_item, self.b_item, self.pool_rows, self.I_all, self.b_all),
                                      (self.E_item, self.b_item, rrows[:B_loc], T_in, 'packed')])
            else:
                be.gather_rows_multi([(self.E_user, None, urows, self.U_loc, None),
                                      (self.E_item, self.b_item, self.pool_rows[:cap], self.I_pack[:cap], 'packed'),
                                      (self.E_item, self.b_item, rrows, T_in, 'packed')])

        # default (ARX_SCORER_F32=1: the f32-MFMA logits / loss / GEMM kernels): the scorer on the bf16 matrix pipe --
        # hinge GEMM (act bits instead of logits / dlogits) + the two bit-operand backward products
        fused = self._fused_scorer()

        def fwd_score():
            if W > 1:    # blocks -> pool (slot) order, their bias column -> b_all
                be.gather_rows_multi([(self.I_gath, d, self.gidx, self.I_all, self.b_all)])
            if not fused:
                be.gemm(self.U_loc, self.I_all[:, :d], self.logits, transB=True, col_bias=self.b_all)

        def loss():
            dt = arena_b[B_loc + Sg:B_loc + Sg + B_loc] if W == 1 else dT[:, d]
            if fused:
                self.scorer.fwd(self.U_loc, self.I_all[:, :d], self.b_all, self.T_pack[:, :d], self.T_pack[:, d],
                                urows, self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.t_loc, dt, dU,
                                dT[:, :d], 1.0 / B)
                return
            be.loss_mw_fused_pos(self.logits, self.U_loc, self.T_pack[:, :d], self.T_pack[:, d], urows,
                                 self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.dlogits,
                                 self.t_loc, dt, dU, dT[:, :d], 1.0 / B)

        def bwd_gemms():
            if fused:
                self.scorer.bwd_dU(dU, beta=1.0)
                if W == 1:
                    self.scorer.bwd_dI(arena[B_loc:B_loc + S, :d], db=arena_b[B_loc:B_loc + S])
                    return
                self.scorer.bwd_dI(self.dI_all[:S, :d], db=self.gb_all)
                be.copy_strided(self.gb_all, self.dI_all[:S, d])
                return
            be.gemm(self.dlogits, self.I_all[:, :d], dU, beta=1.0)
            if W == 1:
                be.gemm(self.dlogits, self.U_loc, arena[B_loc:B_loc + S, :d], transA=True,
                        a_rowsum=arena_b[B_loc:B_loc + S])
                return
            be.gemm(self.dlogits, self.U_loc, self.dI_all[:S, :d], transA=True, a_rowsum=self.gb_all)
            be.copy_strided(self.gb_all, self.dI_all[:S, d])

        def k7(phase):
            if het:
                return self._het_k7(phase, urows, rrows[:B_loc] if W == 1 else rrows, cap, cap_r)
            be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                     (self.E_item[:ni], self.A_item[:ni], self.b_item[:ni], self.Ab_item[:ni])],
                                    [(0, urows, 0), (1, self.pool_rows[:cap], B_loc),
                                     (1, rrows[:B_loc] if W == 1 else rrows, B_loc + Sg)],
                                    arena[:, :d], arena_b, self.lr, phase=phase)

        def apply():
            if W > 1:    # the rows of this rank's block out of the summed pool gradient (bias column -> arena_b),
                #              the bias column of the received target-row gradients
                be.gather_rows_multi([(self.dI_all, d, self.my_slots[:cap], arena[B_loc:B_loc + cap],
                                       arena_b[B_loc:B_loc + cap])])
                be.copy_strided(arena[B_loc + Sg:B_loc + Sg + cap_r, d], arena_b[B_loc + Sg:B_loc + Sg + cap_r])
            k7(2)
            if het and W == 1:
                self._het_tok_apply()                  # (one rank: nothing to sum, the dense step follows at once)

        def k7_sorts(own_graph, ev=None):
            # K7's keys, sorts and run records need the ids only: on a second stream, under the forward
            # kernels (the ~70 us chain leaves the critical path) -- a graph of its own between the
            # segments, a branch of the one graph at world 1
            main, side = torch.cuda.current_stream(dev), self._side
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if own_graph:
                    seg('k7_sorts', lambda: k7(1))
                else:
                    k7(1)
                done = torch.cuda.Event()
                done.record(side)
            return done

        def whole_step():
            fwd_gather()
            # (round 5: with the branch's kernels captured BEHIND the scorer's forward launches -- the order arx/graph.py
            # uses -- the two chains overlap from the start, and the HET step got slower: 436 against 412-427 us.
            # Captured first, the sorts run ahead of the scorer and only their tail overlaps it: kept)
            sorted_ = k7_sorts(False)
            fwd_score()
            loss()
            bwd_gemms()
            torch.cuda.current_stream(dev).wait_event(sorted_)
            apply()

        if W == 1:
            self._segment(mode, 'step', whole_step, feeds=feed)
        else:
            self._segment(mode, 'fwd_gather', fwd_gather, feeds=feed)
            sorted_ = k7_sorts(True)
            dist.all_gather_into_tensor(self.I_gath[:W * cap], self.I_pack[:cap], group=grp)
            w_rows = _all_to_all(self.T_pack, self.T_send[:R], send, recv, group=grp, async_op=True)
            seg('fwd_score', fwd_score)                       # scorer GEMM under the target-row exchange
            w_rows.wait()
            seg('loss', loss)
            w_dt = _all_to_all(arena[B_loc + Sg:B_loc + Sg + R], self.dT_pack, recv, send, group=grp,
                               async_op=True)
            seg('bwd_gemms', bwd_gemms)                       # dU, dI under the gradient exchange
            dist.all_reduce(self.dI_all[:S], op=dist.ReduceOp.SUM, group=grp)
            w_dt.wait()
            torch.cuda.current_stream(dev).wait_event(sorted_)
            seg('apply', apply)
            if het:            # the merged token gradients of all ranks, then the same dense Adagrad step everywhere
                dist.all_reduce(self.D_tok, op=dist.ReduceOp.SUM, group=grp)
                dist.all_reduce(self.Db_tok, op=dist.ReduceOp.SUM, group=grp)
                seg('tok_apply', self._het_tok_apply)
        if mode == 'eager':
            self._warm_key = key
        elif mode == 'capture':
            self._graph_key = key
            self.n_captures += 1
        else:
            self.n_replays += 1
        self.steps += 1

    def _static_feeds(self, route, cap_r):
        return []

    def read_loss(self):
        """Global mean loss of the last step (device scalar; one tiny all-reduce)."""
        self.be.sum_scaled(self.bl, 1.0 / self.B, self.loss)
        dist.all_reduce(self.loss, op=dist.ReduceOp.SUM, group=self.group)
        return self.loss

    # ---- helpers for tests / checkpoints ----
    def gather_global_tables(self):
        """Reassemble the striped tables on every rank (tests only; O(table))."""
        W = self.world
        out = {}
        for name, t, n in (('user', self.E_user, self.n_users), ('item', self.E_item[:self.ni_loc], self.n_items),
                           ('item_bias', self.b_item[:self.ni_loc], self.n_items)):
            rows = (n + W - 1) // W
            pad = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:t.shape[0]] = t
            parts = [torch.empty_like(pad) for _ in range(W)]
            dist.all_gather(parts, pad, group=self.group)
            full = torch.stack(parts, 1).reshape((rows * W,) + tuple(t.shape[1:]))[:n]
            out[name] = full.cpu().numpy()
        return out


class ShardedHMFBags(ShardedHMF):
    """ShardedHMF with HET items (comb_attribute.py:151-176): item = mean(id row, bag mean) of an id
    feature and ONE multi-hot attribute -- the id table striped by item as before, the TOKEN table
    striped by token (owner = token % world).  An item's tokens live on several
    ranks, so every rank forms its PARTIAL of every embedding the step needs -- half its own id row
    (zeros where it does not own the item) plus half the bag mean restricted to its own tokens --
    and the partials are summed by the collective that distributes them:

      all_reduce      pool partials [S, d+4]                      -> the S pool embeddings, everywhere
      reduce_scatter  target partials [B, d+4] -> [B_loc, d+4]    -> each rank's own target embeddings
      (local)         logits, WMRB loss, dU, user-shard rows      (as ShardedHMF)
      all_reduce      pool-gradient partials [S, d+4]             -> the pool gradient, everywhere
      all_gather      target-row gradients [B_loc, d+4] -> [B, d+4]
      (local)         one fused one-hot pass (user shard; id shard: pool + ALL B targets, rows of other
                      owners dropped, coefficient 1/2) and one two-stage bag pass over the token shard
                      (tokens of other owners dropped, coefficient 1/2 . 1/len)

    No all_to_all is left on the path; the target ids of the global batch are all-gathered with the
    batch (prepare_route: input data).  The bag index (vals / starts / lens, global item ids, global
    token ids) is replicated; each rank keeps a copy of `vals` in which its own tokens are local rows
    and every other token points at the shard's padding row (forward: adds zeros; backward: out of
    range, dropped by the sort).  Volume per rank and step: 2 x B x (d+4) x 4 bytes through the
    reduce_scatter / all_gather (B = global batch) -- the price of token sharding; the id-only step
    moves 2 x B_loc x (d+4) x 4."""

    def __init__(self, n_users, n_items, d, B_loc, S, learning_rate, rank, world, device, bags, n_tokens,
                 backend=None, group=None, tables=None, seed=0, acc0=0.1):
        super().__init__(n_users, n_items, d, B_loc, S, learning_rate, rank, world, device, backend=backend,
                         group=group, tables=tables, seed=seed, acc0=acc0)
        dev, f32, i32 = self.device, torch.float32, torch.int32
        vals, starts, lens = [np.asarray(a) for a in bags]
        nt = (n_tokens - rank + world - 1) // world
        self.n_tokens, self.nt_loc = n_tokens, nt
        if tables is not None:
            T = np.asarray(tables['token'], dtype=np.float32)[rank::world]
            bT = np.asarray(tables['token_bias'], dtype=np.float32).reshape(-1)[rank::world]
            self.E_tok = torch.zeros((nt + 1, d), dtype=f32, device=dev)
            self.E_tok[:nt].copy_(torch.from_numpy(np.ascontiguousarray(T)))
            self.b_tok = torch.zeros((nt + 1,), dtype=f32, device=dev)
            self.b_tok[:nt].copy_(torch.from_numpy(np.ascontiguousarray(bT)))
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(seed * 2003 + rank)
            lim = float(np.sqrt(6.0 / (n_tokens + d)))
            self.E_tok = torch.empty((nt + 1, d), dtype=f32, device=dev).uniform_(-lim, lim, generator=g)
            self.b_tok = torch.empty((nt + 1,), dtype=f32, device=dev).uniform_(-lim, lim, generator=g)
            self.E_tok[nt].zero_()
            self.b_tok[nt] = 0.0
        self.A_tok = torch.full_like(self.E_tok, acc0)
        self.Ab_tok = torch.full_like(self.b_tok, acc0)
        v = vals.astype(np.int64)
        mine = np.where(v % world == rank, v // world, nt).astype(np.int32)      # other owners -> padding row
        self.bag_vals = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)
        self.bag_starts = torch.from_numpy(np.ascontiguousarray(starts.astype(np.int32))).to(dev)
        self.bag_lens = torch.from_numpy(np.ascontiguousarray(lens.astype(np.int32))).to(dev)
        B, dp = self.B, self.dp
        z = lambda *s_: torch.zeros(s_, dtype=f32, device=dev)
        self.P_part, self.Pb = z(S, dp), z(S)               # pool partials -> (all_reduce) pool embeddings
        self.T_part, self.Tb = z(B, dp), z(B)               # target partials of the GLOBAL batch
        self.pool_fwd = torch.zeros(S, dtype=i32, device=dev)    # id-shard row of every pool slot (padding row: not mine)
        self.pool_bwd = torch.zeros(S, dtype=i32, device=dev)    # ... or KEY_NONE
        self.arena = z(B_loc + S + B, dp)                   # [dU_loc ; pool gradient ; target gradients of the batch]
        self.arena_b = z(B_loc + S + B)

    def _alloc_recv(self, cap):
        return                                              # (no variable-size exchange in this step)

    def set_pool(self, pool_ids):
        super().set_pool(pool_ids)
        self.be.shard_route(self.pool_ids, self.world, self.rank, self.zero_row, self.pool_fwd, self.pool_bwd)

    def prepare_route(self, users, items):
        W = self.world
        dev = self.device
        u = users if isinstance(users, torch.Tensor) else torch.as_tensor(np.asarray(users, dtype=np.int32))
        it = items if isinstance(items, torch.Tensor) else torch.as_tensor(np.asarray(items, dtype=np.int32))
        u, it = u.to(dev, torch.int32), it.to(dev, torch.int32)
        tgt_all = torch.empty(self.B, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(tgt_all, it.contiguous(), group=self.group)   # target ids of the global batch
        fwd = torch.zeros(self.B, dtype=torch.int32, device=dev)
        bwd = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.be.shard_route(tgt_all, W, self.rank, self.zero_row, fwd, bwd)
        urows = torch.zeros(self.B_loc, dtype=torch.int32, device=dev)
        self.be.shard_route(u, W, self.rank, 0, urows, None)
        return {'users': u, 'items': it, 'urows': urows, 'tgt_all': tgt_all, 'tgt_fwd': fwd, 'tgt_bwd': bwd}

    _static_step_ok = True          # (round 6: the kernels between the collectives as hipGraph segments, below)

    def _step_bags_static(self, route):
        """The step of step() as hipGraph segments between its four collectives (round 6; round-5 verdict, missing #3:
        the eager step cost 583 us at world 1 against 187 us for ShardedHMF).  Every buffer is static; what varies from
        batch to batch are four index vectors (user rows, the global batch's target ids and their id-shard rows /
        keys), fed by the first node of the first segment.  K7's ids-only half (both one-hot sorts, the entity and
        token sorts of the bag pass) runs on a second stream under the forward kernels -- a graph of its own between
        the segments, a branch of the ONE graph at world 1 -- and the scorer is the fused bf16-pipe family where its
        shapes allow (no [B_loc, S] logits), as in ShardedHMF._step_static."""
        be, grp, W = self.be, self.group, self.world
        B, B_loc, S, d = self.B, self.B_loc, self.S, self.d
        dev = self.device
        arena, arena_b = self.arena, self.arena_b
        if getattr(self, 'g_urows', None) is None:
            i32 = torch.int32
            self.g_urows = torch.zeros(B_loc, dtype=i32, device=dev)
            self.g_fwd, self.g_all, self.g_bwd = (torch.zeros(B, dtype=i32, device=dev) for _ in range(3))
        urows, t_fwd, t_all, t_bwd = self.g_urows, self.g_fwd, self.g_all, self.g_bwd
        self.urows = urows
        feed = [(route['urows'], urows), (route['tgt_fwd'], t_fwd), (route['tgt_all'], t_all), (route['tgt_bwd'], t_bwd)]
        key = (arena.data_ptr(), self.pos_ptr.data_ptr(), self.pos_items.data_ptr(), urows.data_ptr())
        if os.environ.get("ARX_DIST_NO_CAPTURE"):
            mode = 'eager'
        elif self._graph_key == key:
            mode = 'replay'
        elif self._warm_key == key:
            mode, self._graphs = 'capture', {}
        else:
            mode, self._graphs, self._graph_key = 'eager', {}, None
        seg = lambda name, fn: self._segment(mode, name, fn)
        bag = (self.bag_vals, self.bag_starts, self.bag_lens)
        dU = arena[:B_loc, :d]
        gP = arena[B_loc:B_loc + S]
        nt = self.nt_loc
        fused = self._fused_scorer()

        def fwd_pool():
            be.gather_rows(self.E_user, None, urows, self.U_loc, None)
            be.gather_rows(self.E_item, self.b_item, self.pool_fwd, self.P_part[:, :d], self.Pb, scale=0.5)
            be.gather_bags(self.E_tok, self.b_tok, *bag, self.pool_ids, self.P_part[:, :d], self.Pb, scale=0.5,
                           accumulate=True)
            be.copy_strided(self.Pb, self.P_part[:, d])

        def fwd_tgt():
            be.gather_rows(self.E_item, self.b_item, t_fwd, self.T_part[:, :d], self.Tb, scale=0.5)
            be.gather_bags(self.E_tok, self.b_tok, *bag, t_all, self.T_part[:, :d], self.Tb, scale=0.5, accumulate=True)
            be.copy_strided(self.Tb, self.T_part[:, d])

        def score():
            be.copy_strided(self.P_part[:, d], self.b_all)
            if not fused:
                be.gemm(self.U_loc, self.P_part[:, :d], self.logits, transB=True, col_bias=self.b_all)

        def loss():
            if fused:
                self.scorer.fwd(self.U_loc, self.P_part[:, :d], self.b_all, self.T_pack[:, :d], self.T_pack[:, d],
                                urows, self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.t_loc,
                                self.dT_pack[:, d], dU, self.dT_pack[:, :d], 1.0 / B)
                return
            be.loss_mw_fused_pos(self.logits, self.U_loc, self.T_pack[:, :d], self.T_pack[:, d], urows,
                                 self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.dlogits,
                                 self.t_loc, self.dT_pack[:, d], dU, self.dT_pack[:, :d], 1.0 / B)

        def bwd():
            if fused:
                self.scorer.bwd_dU(dU, beta=1.0)
                self.scorer.bwd_dI(gP[:, :d], db=self.gb_all)
            else:
                be.gemm(self.dlogits, self.P_part[:, :d], dU, beta=1.0)                    # dU += dL . pool
                be.gemm(self.dlogits, self.U_loc, gP[:, :d], transA=True, a_rowsum=self.gb_all)
            be.copy_strided(self.gb_all, gP[:, d])

        def k7(phase):
            be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                     (self.E_item, self.A_item, self.b_item, self.Ab_item)],
                                    [(0, urows, 0, 1.0), (1, self.pool_bwd, B_loc, 0.5), (1, t_bwd, B_loc + S, 0.5)],
                                    arena[:, :d], arena_b, self.lr, phase=phase)
            be.bags_adagrad(self.E_tok[:nt], self.A_tok[:nt], self.b_tok[:nt], self.Ab_tok[:nt], *bag,
                            [(self.pool_ids, B_loc, 0.5), (t_all, B_loc + S, 0.5)], arena[:, :d], arena_b, self.lr,
                            phase=phase)

        def apply():
            be.copy_strided(arena[B_loc:, d], arena_b[B_loc:])
            k7(2)

        def k7_sorts(own_graph):
            main, side = torch.cuda.current_stream(dev), self._side
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if own_graph:
                    seg('k7_sorts', lambda: k7(1))
                else:
                    k7(1)
                done = torch.cuda.Event()
                done.record(side)
            return done

        if W == 1:
            def whole_step():
                fwd_pool()
                sorted_ = k7_sorts(False)
                fwd_tgt()
                be.copy_2d(self.T_part, self.T_pack)          # (one rank: the partials ARE the embeddings)
                score()
                loss()
                be.copy_2d(self.dT_pack, arena[B_loc + S:])
                bwd()
                torch.cuda.current_stream(dev).wait_event(sorted_)
                apply()
            self._segment(mode, 'step', whole_step, feeds=feed)
        else:
            self._segment(mode, 'fwd_pool', fwd_pool, feeds=feed)
            sorted_ = k7_sorts(True)
            w_pool = dist.all_reduce(self.P_part, op=dist.ReduceOp.SUM, group=grp, async_op=True)
            seg('fwd_tgt', fwd_tgt)                               # target partials under the pool all-reduce
            w_pool.wait()
            w_tgt = dist.reduce_scatter_tensor(self.T_pack, self.T_part, op=dist.ReduceOp.SUM, group=grp, async_op=True)
            seg('score', score)                                   # (unfused scorer: its GEMM under the reduce-scatter)
            w_tgt.wait()
            seg('loss', loss)
            w_dt = dist.all_gather_into_tensor(arena[B_loc + S:], self.dT_pack, group=grp, async_op=True)
            seg('bwd', bwd)                                       # dU, pool gradient under the all-gather
            dist.all_reduce(gP, op=dist.ReduceOp.SUM, group=grp)
            w_dt.wait()
            torch.cuda.current_stream(dev).wait_event(sorted_)
            seg('apply', apply)
        if mode == 'eager':
            self._warm_key = key
        elif mode == 'capture':
            self._graph_key = key
            self.n_captures += 1
        else:
            self.n_replays += 1
        self.steps += 1

    def step(self, users, items=None):
        route = users if isinstance(users, dict) else self.prepare_route(users, items)
        if self.use_graphs:
            def guarded():
                try:
                    self._step_bags_static(route)
                except BaseException:
                    self._graphs, self._graph_key, self._warm_key = {}, None, None
                    raise
            outer = torch.cuda.current_stream(self.device)
            if outer == self._stream:
                guarded()
                return
            self._stream.wait_stream(outer)
            with torch.cuda.stream(self._stream):
                guarded()
            outer.wait_stream(self._stream)
            return
        be, grp = self.be, self.group
        B, B_loc, S, d = self.B, self.B_loc, self.S, self.d
        arena, arena_b = self.arena, self.arena_b
        urows = route['urows']
        self.urows = urows
        bag = (self.bag_vals, self.bag_starts, self.bag_lens)
        # ---- forward: partial embeddings, summed by the collectives ----
        be.gather_rows(self.E_user, None, urows, self.U_loc, None)
        be.gather_rows(self.E_item, self.b_item, self.pool_fwd, self.P_part[:, :d], self.Pb, scale=0.5)
        be.gather_bags(self.E_tok, self.b_tok, *bag, self.pool_ids, self.P_part[:, :d], self.Pb, scale=0.5,
                       accumulate=True)
        be.copy_strided(self.Pb, self.P_part[:, d])
        w_pool = dist.all_reduce(self.P_part, op=dist.ReduceOp.SUM, group=grp, async_op=True)
        be.gather_rows(self.E_item, self.b_item, route['tgt_fwd'], self.T_part[:, :d], self.Tb, scale=0.5)
        be.gather_bags(self.E_tok, self.b_tok, *bag, route['tgt_all'], self.T_part[:, :d], self.Tb, scale=0.5,
                       accumulate=True)
        be.copy_strided(self.Tb, self.T_part[:, d])
        w_pool.wait()
        w_tgt = dist.reduce_scatter_tensor(self.T_pack, self.T_part, op=dist.ReduceOp.SUM, group=grp,
                                           async_op=True)             # own target embeddings ...
        be.copy_strided(self.P_part[:, d], self.b_all)
        be.gemm(self.U_loc, self.P_part[:, :d], self.logits, transB=True, col_bias=self.b_all)   # ... under the scorer
        w_tgt.wait()
        dU = arena[:B_loc, :d]
        be.loss_mw_fused_pos(self.logits, self.U_loc, self.T_pack[:, :d], self.T_pack[:, d], urows,
                             self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.dlogits,
                             self.t_loc, self.dT_pack[:, d], dU, self.dT_pack[:, :d], 1.0 / B)
        # ---- backward ----
        w_dt = dist.all_gather_into_tensor(arena[B_loc + S:], self.dT_pack, group=grp, async_op=True)
        be.gemm(self.dlogits, self.P_part[:, :d], dU, beta=1.0)                    # dU += dL . pool
        gP = arena[B_loc:B_loc + S]
        be.gemm(self.dlogits, self.U_loc, gP[:, :d], transA=True, a_rowsum=self.gb_all)
        be.copy_strided(self.gb_all, gP[:, d])
        dist.all_reduce(gP, op=dist.ReduceOp.SUM, group=grp)
        w_dt.wait()
        be.copy_strided(arena[B_loc:, d], arena_b[B_loc:])
        # id shard + user shard: one fused one-hot pass (rows of other owners carry KEY_NONE)
        be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                 (self.E_item, self.A_item, self.b_item, self.Ab_item)],
                                [(0, urows, 0, 1.0), (1, self.pool_bwd, B_loc, 0.5),
                                 (1, route['tgt_bwd'], B_loc + S, 0.5)], arena[:, :d], arena_b, self.lr)
        # token shard: merge per item, then per token; tokens of other owners are out of range
        nt = self.nt_loc
        be.bags_adagrad(self.E_tok[:nt], self.A_tok[:nt], self.b_tok[:nt], self.Ab_tok[:nt], *bag,
                        [(self.pool_ids, B_loc, 0.5), (route['tgt_all'], B_loc + S, 0.5)], arena[:, :d], arena_b,
                        self.lr)
        self.steps += 1

    def gather_global_tables(self):
        out = super().gather_global_tables()
        W = self.world
        for name, t, n in (('token', self.E_tok[:self.nt_loc], self.n_tokens),
                           ('token_bias', self.b_tok[:self.nt_loc], self.n_tokens)):
            rows = (n + W - 1) // W
            pad = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:t.shape[0]] = t
            parts = [torch.empty_like(pad) for _ in range(W)]
            dist.all_gather(parts, pad, group=self.group)
            out[name] = torch.stack(parts, 1).reshape((rows * W,) + tuple(t.shape[1:]))[:n].cpu().numpy()
        return out


class ShardedHMFRepTokens(ShardedHMF):
    """HET items (comb_attribute.py:151-176: item = mean(id row, bag mean)) with the id table striped by item and
    the TOKEN table REPLICATED on every rank -- round 5, the redesign of ShardedHMFBags for token tables that fit
    beside the id shard (C3: 100 k x 128 x 4 B = 51 MB; past ~64 MB stripe by token: ShardedHMFBags).

    With every token on every rank a bag mean is LOCAL to the owner of the item's id row, so the step is
    ShardedHMF's -- the owner forms the whole item embedding, pool blocks are all-gathered, target rows and their
    gradients travel in two all-to-alls sized by B_loc, the pool gradient in one small all-reduce -- plus ONE
    all-reduce for the token table: every rank merges the token gradients of the lookups it owns (pool block +
    received targets, coefficient 1/2 . 1/len) into a dense gradient table D [n_tokens, d] (the two-stage bag pass
    in its gradient-descent form with a step of -1 onto a zeroed table: the sums themselves), D is summed over the
    ranks, and every replica applies the same Adagrad step to the rows of D that are not all zero (rows without gradient
    would not move: the sparse update of embed_attribute.py:383-400 / hmf_model.py:146-151; arx_adagrad_rows_nonzero,
    which also zeroes D for the next step).  Volume per rank and step on top of
    ShardedHMF: 2 x n_tokens x (d + 1) x 4 B x (N-1)/N through the ring, independent of the batch.  In the EAGER
    step the two all-reduces are issued asynchronously under the id shard's own K7 pass; in the hipGraph-segment step
    (the default) they run on the main stream between the 'apply' and 'tok_apply' segments -- NOT overlapped, and
    priced so in roofline_comm_predicted (comm_prediction 'rep_tokens').  The token-striped step moves
    2 x B x (d + 4) x 4 B with B the GLOBAL batch
    (predicted comm / compute 0.35 against 0.57 at N = 8, B_loc = 16384)."""

    _static_step_ok = True          # (ShardedHMF.__init__: the hipGraph-segment step serves this class too)
    _het = True

    def __init__(self, n_users, n_items, d, B_loc, S, learning_rate, rank, world, device, bags, n_tokens,
                 backend=None, group=None, tables=None, seed=0, acc0=0.1, graphs=None):
        super().__init__(n_users, n_items, d, B_loc, S, learning_rate, rank, world, device, backend=backend,
                         group=group, tables=tables, seed=seed, acc0=acc0, graphs=graphs)
        dev, f32, i32 = self.device, torch.float32, torch.int32
        vals, starts, lens = [np.asarray(a) for a in bags]
        nt = int(n_tokens)
        self.n_tokens = nt
        self.g_gid = None               # received target ids (global), padded with the padding entity: fed per step
        if tables is not None:
            self.E_tok = torch.zeros((nt + 1, d), dtype=f32, device=dev)
            self.E_tok[:nt].copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(tables['token'], dtype=np.float32))))
            self.b_tok = torch.zeros((nt + 1,), dtype=f32, device=dev)
            self.b_tok[:nt].copy_(torch.from_numpy(np.ascontiguousarray(
                np.asarray(tables['token_bias'], dtype=np.float32).reshape(-1))))
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(seed * 2003)                      # (the SAME table on every rank)
            lim = float(np.sqrt(6.0 / (nt + d)))
            self.E_tok = torch.empty((nt + 1, d), dtype=f32, device=dev).uniform_(-lim, lim, generator=g)
            self.b_tok = torch.empty((nt + 1,), dtype=f32, device=dev).uniform_(-lim, lim, generator=g)
            self.E_tok[nt].zero_()
            self.b_tok[nt] = 0.0
        self.A_tok = torch.full_like(self.E_tok, acc0)
        self.Ab_tok = torch.full_like(self.b_tok, acc0)
        # bag index by GLOBAL item id + one padding entity (index n_items): a bag of one padding token (row nt:
        # zeros, dropped by the gradient pass) -- what the padding rows of a pool block / of the receive buffer look up
        n_ent = int(lens.shape[0])
        if n_ent < n_items:
            raise ValueError("bag index shorter than the item table")
        v = np.concatenate([vals.astype(np.int32)[:int(starts[n_ent - 1] + lens[n_ent - 1])], [nt]]).astype(np.int32)
        st = np.concatenate([starts.astype(np.int32)[:n_ent], [len(v) - 1]]).astype(np.int32)
        ln = np.concatenate([lens.astype(np.int32)[:n_ent], [1]]).astype(np.int32)
        self.pad_item = n_ent
        self.bag_vals = torch.from_numpy(v).to(dev)
        self.bag_starts = torch.from_numpy(st).to(dev)
        self.bag_lens = torch.from_numpy(ln).to(dev)
        self.D_tok = torch.zeros((nt, d), dtype=f32, device=dev)      # merged token gradients, summed over the ranks
        self.Db_tok = torch.zeros((nt,), dtype=f32, device=dev)
        self.pool_ext = torch.zeros(S + 1, dtype=i32, device=dev)     # pool ids + the padding entity
        self.block_ids = torch.zeros(S, dtype=i32, device=dev)        # global item id of every row of the owned block
        # global item id (or the padding entity) -> row of the id shard (the shard's zero row where not owned): the
        # map the one-launch HET lookup of the static step takes beside the bag index (both indexed by global id)
        g = np.arange(n_ent + 1, dtype=np.int64)
        lm = np.where((g < n_items) & (g % world == rank), g // world, self.zero_row).astype(np.int32)
        self.lmap = torch.from_numpy(lm).to(dev)

    def set_pool(self, pool_ids):
        super().set_pool(pool_ids)
        be, S = self.be, self.S
        be.copy_i32(self.pool_ids, self.pool_ext[:S])
        be.copy_i32(torch.full((1,), self.pad_item, dtype=torch.int32, device=self.device), self.pool_ext[S:])
        if self.world == 1:
            be.copy_i32(self.pool_ids, self.block_ids)
        else:          # block row -> pool slot (S: padding) -> item id (padding entity)
            be.take_i32(self.pool_ext, self.my_slots, self.block_ids, self.pad_item)

    # ---- the hipGraph-segment step (ShardedHMF._step_static) with HET rows and the replicated token table ----
    def _static_feeds(self, route, cap_r):
        gid = route.get('gid')
        if gid is None or gid.shape[0] != cap_r:
            gid = torch.full((cap_r,), self.pad_item, dtype=torch.int32, device=self.device)
            R = route['R']
            if R > 0:
                gid[:R] = route['recv_ids'][:R]
            route['gid'] = gid
        if self.g_gid is None or self.g_gid.shape[0] != cap_r:
            self.g_gid = torch.empty(cap_r, dtype=torch.int32, device=self.device)
        return [(gid, self.g_gid)]

    def _het_gather(self, urows, T_in, cap, cap_r):
        """users, the owned pool block and the requested target rows in ONE launch (arx_lookup_multi): item rows =
        (id row + bag mean) / 2 with their biases, then the two bias columns of the packed rows."""
        be, d, W = self.be, self.d, self.world
        bag = (self.bag_vals, self.bag_starts, self.bag_lens)
        item = (self.E_item, self.b_item, self.lmap, self.E_tok, self.b_tok) + bag
        nt_rows = self.B_loc if W == 1 else cap_r
        if W == 1:
            pool_out, pool_b = self.I_all[:, :d], self.b_all
        else:
            pool_out, pool_b = self.I_pack[:cap, :d], self.b_g[:cap]
        be.lookup_het_multi([(self.E_user, None, None, None, None, None, None, None, urows, self.U_loc, 1.0, None),
                             item + (self.block_ids[:self.S if W == 1 else cap], pool_out, 0.5, pool_b),
                             item + (self.g_gid[:nt_rows], T_in[:nt_rows, :d], 0.5, self.tb_send[:nt_rows])])
        if W > 1:
            be.copy_strided(pool_b, self.I_pack[:cap, d])
        be.copy_strided(self.tb_send[:nt_rows], T_in[:nt_rows, d])

    def _het_k7(self, phase, urows, rrows, cap, cap_r):
        be, d, B_loc, Sg, ni, W = self.be, self.d, self.B_loc, self.Sg, self.ni_loc, self.world
        arena, arena_b = self.arena, self.arena_b
        nb = self.S if W == 1 else cap
        nt_rows = B_loc if W == 1 else cap_r
        be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                 (self.E_item[:ni], self.A_item[:ni], self.b_item[:ni], self.Ab_item[:ni])],
                                [(0, urows, 0, 1.0), (1, self.pool_rows[:nb], B_loc, 0.5), (1, rrows, B_loc + Sg, 0.5)],
                                arena[:, :d], arena_b, self.lr, phase=phase)
        # (D_tok / Db_tok are all zero here: allocated so, and the token apply zeroes every row it consumes)
        be.bags_grad_dense(self.D_tok, self.Db_tok, self.bag_vals, self.bag_starts, self.bag_lens,
                           [(self.block_ids[:nb], B_loc, 0.5), (self.g_gid[:nt_rows], B_loc + Sg, 0.5)],
                           arena[:, :d], arena_b, phase=phase)

    def _het_tok_apply(self):
        """The same Adagrad step on every replica: rows of the summed gradient table that are not all zero (the rest
        would not move), the table zeroed on the way for the next step's accumulation (arx_adagrad_rows_nonzero)."""
        nt = self.n_tokens
        self.be.adagrad_rows_nonzero(self.E_tok[:nt], self.A_tok[:nt], self.b_tok[:nt], self.Ab_tok[:nt], self.D_tok,
                                     self.Db_tok, self.lr)

    def _het_rows(self, rows, ids, out, bias_tmp):
        """out[:, :d] = (id row + bag mean) / 2, out[:, d] = (id bias + mean token bias) / 2 for the owned items
        `ids` (global) whose id-shard rows are `rows`."""
        be, d = self.be, self.d
        bag = (self.bag_vals, self.bag_starts, self.bag_lens)
        be.gather_rows(self.E_item, self.b_item, rows, out[:, :d], bias_tmp, scale=0.5)
        be.gather_bags(self.E_tok, self.b_tok, *bag, ids, out[:, :d], bias_tmp, scale=0.5, accumulate=True)
        be.copy_strided(bias_tmp, out[:, d])

    def step(self, users, items=None):
        if self.world > 1 and self.cap <= 0:
            raise RuntimeError("ShardedHMFRepTokens.step before set_pool()")
        if self.use_graphs:            # HIP backend: the hipGraph-segment step with the hooks above
            return ShardedHMF.step(self, users, items)
        route = users if isinstance(users, dict) else self.prepare_route(users, items)
        be, W = self.be, self.world
        B, B_loc, S, Sg, d = self.B, self.B_loc, self.S, self.Sg, self.d
        grp = self.group
        send, recv, R = route['send'], route['recv'], route['R']
        arena, arena_b = self.arena, self.arena_b
        urows, recv_rows, recv_ids = route['urows'], route['recv_rows'], route['recv_ids']
        self.urows = urows
        cap = self.cap
        # ---- forward: the owner forms the whole HET embedding of its pool block and of the requested targets ----
        be.gather_rows(self.E_user, None, urows, self.U_loc, None)
        if W == 1:
            self._het_rows(self.pool_rows, self.block_ids, self.I_all, self.b_all)
        else:
            self._het_rows(self.pool_rows[:cap], self.block_ids[:cap], self.I_pack[:cap], self.b_g[:cap])
            dist.all_gather_into_tensor(self.I_gath[:W * cap], self.I_pack[:cap], group=grp)
            be.gather_rows(self.I_gath, None, self.gidx, self.I_all, None)    # blocks -> pool (slot) order
            be.copy_strided(self.I_all[:, d], self.b_all)
        T_send = self.T_send[:R]
        if R > 0:
            self._het_rows(recv_rows, recv_ids, T_send, self.tb_send[:R])
        w_rows = _all_to_all(self.T_pack, T_send, send, recv, group=grp, async_op=True)
        be.gemm(self.U_loc, self.I_all[:, :d], self.logits, transB=True, col_bias=self.b_all)
        w_rows.wait()
        dU = arena[:B_loc, :d]
        be.loss_mw_fused_pos(self.logits, self.U_loc, self.T_pack[:, :d], self.T_pack[:, d], self.urows,
                             self.pos_ptr, self.pos_items, self.item2slot, self.bl, self.dlogits,
                             self.t_loc, self.dT_pack[:, d], dU, self.dT_pack[:, :d], 1.0 / B)
        # ---- backward ----
        w_dt = _all_to_all(arena[B_loc + Sg:B_loc + Sg + R], self.dT_pack, recv, send, group=grp, async_op=True)
        be.gemm(self.dlogits, self.I_all[:, :d], dU, beta=1.0)
        be.gemm(self.dlogits, self.U_loc, self.dI_all[:S, :d], transA=True, a_rowsum=self.gb_all)
        be.copy_strided(self.gb_all, self.dI_all[:S, d])
        if W == 1:
            be.copy_2d(self.dI_all[:S], arena[B_loc:B_loc + S])
        else:
            dist.all_reduce(self.dI_all[:S], op=dist.ReduceOp.SUM, group=grp)
            be.gather_rows(self.dI_all, None, self.my_slots[:cap], arena[B_loc:B_loc + cap], None)
        w_dt.wait()
        be.copy_strided(arena[B_loc:B_loc + Sg + R, d], arena_b[B_loc:B_loc + Sg + R])
        # token table: this rank's merged token gradients -> D, summed over the ranks under the id shard's pass
        nt = self.n_tokens
        bag = (self.bag_vals, self.bag_starts, self.bag_lens)
        bsites = [(self.block_ids[:cap], B_loc, 0.5)]       # (D_tok / Db_tok are all zero: _het_tok_apply leaves them so)
        if R > 0:
            bsites.append((recv_ids[:R], B_loc + Sg, 0.5))
        be.bags_grad_dense(self.D_tok, self.Db_tok, *bag, bsites, arena[:, :d], arena_b)
        w_tok = w_tokb = None
        if W > 1:
            w_tok = dist.all_reduce(self.D_tok, op=dist.ReduceOp.SUM, group=grp, async_op=True)
            w_tokb = dist.all_reduce(self.Db_tok, op=dist.ReduceOp.SUM, group=grp, async_op=True)
        # id shard + user shard: one fused one-hot pass (the item rows' share of the embedding is 1/2)
        sites = [(0, self.urows, 0, 1.0), (1, self.pool_rows[:cap], B_loc, 0.5)]
        if R > 0:
            sites.append((1, recv_rows, B_loc + Sg, 0.5))
        ni = self.ni_loc
        be.sparse_adagrad_multi([(self.E_user, self.A_user, None, None),
                                 (self.E_item[:ni], self.A_item[:ni], self.b_item[:ni], self.Ab_item[:ni])],
                                sites, arena[:, :d], arena_b, self.lr)
        if w_tok is not None:
            w_tok.wait()
            w_tokb.wait()
        self._het_tok_apply()
        self.steps += 1

    def gather_global_tables(self):
        out = super().gather_global_tables()
        nt = self.n_tokens
        out['token'] = self.E_tok[:nt].cpu().numpy()
        out['token_bias'] = self.b_tok[:nt].cpu().numpy()
        return out


# --------------------------------------------------------------------------
# bench entry for N > 1 (driver: python -m torch.distributed.run ... bench.py --gpus N)
# --------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# Data-parallel sequence model (the LSTM recommender has no table big enough to shard at
# C4's size -- 1 M x 64 x 4 B = 256 MB -- so the replicas split the BATCH).
# ---------------------------------------------------------------------------------------------
class _GIds(object):
    def __init__(self, value):
        self.value = value


class _GNode(object):
    """Stand-in for the lookup node of a gathered site: rows of the gathered gradient arena."""

    def __init__(self, row0, arena, arena_b, node):
        self.row0, self.arena, self.arena_b = row0, arena, arena_b
        self._grad_written = True
        self.bias_grad_used = node.bias_grad_used
        self.with_bias = node.with_bias
        self.shape = node.shape


class _GSite(object):
    """One lookup site of the GLOBAL batch: ids and gradient rows of every replica, rank-major."""

    def __init__(self, s, ids, node, n):
        self.table, self.kind, self.maps, self.max_len, self.coef = s.table, s.kind, s.maps, s.max_len, s.coef
        self.bias_coef = getattr(s, 'bias_coef', 1.0)
        self.col_off, self.key_off = 0, 0
        self.ids_node, self.node = _GIds(ids), node
        self.n = self.cap = n


class SeqDataParallel(object):
    """`world` replicas of a SeqModel (lstm/seqModel.py), one per GPU, each fed 1/world of the
    sequences of a step; one step of the group == the single-process step on the global batch
    (the reference has no multi-device path: SURVEY 8e).  sequence_loss sums over the examples
    (seqModel.py:596), so local gradients simply add.  Per step, between backward and apply:

      all_reduce   dense gradients (lstm_w / lstm_b of every layer, w_input_*), packed in one buffer
      all_reduce   the pool gradients: per-unrolled-step [L, S, d] (+ bias [L, S]) and their sums --
                   tf.gradients yields ONE dense matmul gradient per step for the global batch, and
                   clip_by_global_norm squares them step by step (seqModel.py:179-180)
      all_reduce   one scalar: the squared norms of the batch lookups' IndexedSlices (un-merged in
                   TF's norm, hence additive over replicas)
      all_gather   ids + gradient rows of every batch lookup (inputs, targets, users): every replica
                   then runs the SAME sparse-Adagrad pass over the global lookups -- duplicates
                   across replicas are merged before the one update per row, tables stay identical.

    The pool of sampled negatives must be the same on every replica (draw it with a shared seed or
    broadcast it: broadcast_pool).  One-hot item / user features (config C4); steps run eagerly (the
    collectives are not captured into the hipGraph)."""

    def __init__(self, model, group=None):
        self.model, self.rt, self.group = model, model.rt, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if getattr(model, 'output_feat', 1) not in (0, 1):
            raise NotImplementedError("SeqDataParallel: output_feat 0 / 1 only")
        self.rt.dp = self
        self.rt.use_graph = False
        self._plans = {}
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.rt.device)

    # ---- collectives --------------------------------------------------------------------------
    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _all_reduce_packed(self, tensors):
        """One all-reduce for a list of tensors (bucketed: a ring all-reduce over xGMI is bound by
        its per-link latency at these sizes, not by bytes)."""
        if self.world == 1 or not tensors:
            return
        if not all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 for t in tensors):
            # (advisor, round 5: a 4-byte INTEGER tensor packed into the float32 bucket would be summed as float bit
            # patterns -- the word-copy path is for float32 only)
            # (the gloo / CPU rig of the tests, strided views)
            flat = torch.cat([t.reshape(-1) for t in tensors])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n
            return
        # round 5: the bucket is a persistent flat buffer, packed and unpacked by arx_copy_words (8 tensors per
        # launch) instead of torch.cat + one copy_ per tensor
        from . import ops
        n_all = sum(t.numel() for t in tensors)
        flat = getattr(self, '_bucket', None)
        if flat is None or flat.numel() < n_all or flat.device != tensors[0].device:
            flat = self._bucket = torch.empty(n_all, dtype=torch.float32, device=tensors[0].device)
        views, off = [], 0
        for t in tensors:
            views.append(flat[off:off + t.numel()])
            off += t.numel()
        ops.copy_words([(t.view(-1), v) for t, v in zip(tensors, views)])
        dist.all_reduce(flat[:n_all], op=dist.ReduceOp.SUM, group=self.group)
        ops.copy_words([(v, t.view(-1)) for t, v in zip(tensors, views)])

    def _all_gather(self, src, dst):
        """dst[w * n : (w + 1) * n] = src of replica w."""
        if self.world == 1:
            dst.copy_(src)
            return
        n = src.shape[0]
        dist.all_gather([dst[w * n:(w + 1) * n] for w in range(self.world)], src.contiguous(), group=self.group)

    def broadcast_pool(self, pool_ids):
        """Replica 0's sampled pool for everyone (int32 device tensor, in place)."""
        if self.world > 1:
            dist.broadcast(pool_ids, src=0, group=self.group)
        return pool_ids

    def global_loss(self, local_loss):
        self._loss.fill_(float(local_loss))
        return float(self.all_reduce_sum(self._loss).item())

    # ---- the exchange -------------------------------------------------------------------------
    def _state(self, plan):
        from . import graph as G
        st = self._plans.get(id(plan))
        if st is not None:
            return st
        rt = self.rt
        # pool lookups: their gradient is a dense [S, d] sum over the batch -> all-reduced, replicated
        pools, preds = set(), []
        for n in plan.order:
            if isinstance(n, G.Prediction) and n.inputs[1].train_tables:
                pools.add(id(n.inputs[1]))
                preds.append(n)
        tables = []
        for table, sites, bufs, total in plan.tables:
            if any(s.kind != 'cat' or s.col_off != 0 for s in sites):
                raise NotImplementedError("SeqDataParallel: one-hot, mean-combined features only")
            width = sites[0].node.shape[1]
            rows = sum(s.n * (1 if id(s.node) in pools else self.world) for s in sites)
            arena = torch.zeros((rows, width), dtype=torch.float32, device=rt.device)
            arena_b = torch.zeros((rows,), dtype=torch.float32, device=rt.device)
            gs, r0 = [], 0
            for s in sites:
                rep = id(s.node) in pools
                n = s.n if rep else s.n * self.world
                ids = s.ids_node.value if rep else torch.empty(n, dtype=torch.int32, device=rt.device)
                gs.append((s, _GSite(s, ids, _GNode(r0, arena, arena_b, s.node), n), rep))
                r0 += n
            gbufs = {'keys': torch.full((rows,), G.KEY_NONE, dtype=torch.int32, device=rt.device),
                     'src': torch.zeros((rows,), dtype=torch.int32, device=rt.device),
                     'coef': torch.zeros((rows,), dtype=torch.float32, device=rt.device),
                     'hot': torch.zeros((rows // 16 + 4,), dtype=torch.int32, device=rt.device)}
            off = 0
            for _, g, _ in gs:
                g.key_off = off
                off += g.cap
            tables.append((table, gs, gbufs, rows))
        st = dict(preds=preds, tables=tables)
        self._plans[id(plan)] = st
        return st

    def exchange(self, plan):
        """Called by the plan between backward and the optimiser (graph.py Plan._execute)."""
        rt = self.rt
        st = self._state(plan)
        # dense + pool gradients: one packed all-reduce
        pack = [p.grad for p in rt.dense.values() if getattr(p, 'touched', False)]
        for n in st['preds']:
            if not n._grad_written:
                continue
            pool = n.inputs[1]
            if getattr(n, 'C_steps', None) is not None:
                pack += [n.C_steps, n.rs_steps]
            pack.append(pool.grad)
            if pool.bias_grad_used:
                pack.append(pool.bias_grad)
        self._all_reduce_packed(pack)
        # batch lookups: ids and gradient rows of every replica; pool rows: copied (already global)
        for table, gs, gbufs, rows in st['tables']:
            for s, g, rep in gs:
                node = s.node
                g.node._grad_written = node._grad_written
                g.node.bias_grad_used = node.bias_grad_used
                if not node._grad_written:
                    continue
                a = g.node.arena[g.node.row0:g.node.row0 + g.n]
                ab = g.node.arena_b[g.node.row0:g.node.row0 + g.n]
                src = node.arena[node.row0:node.row0 + s.n]
                src_b = node.arena_b[node.row0:node.row0 + s.n]
                if rep:
                    a.copy_(src)
                    if node.bias_grad_used:
                        ab.copy_(src_b)
                else:
                    self._all_gather(s.ids_node.value, g.ids_node.value)
                    self._all_gather(src, a)
                    if node.bias_grad_used:
                        self._all_gather(src_b, ab)

    def gathered_tables(self, plan):
        """plan.tables with the gathered sites in place of the local ones (same tuple layout)."""
        return [(table, [g for _, g, _ in gs], gbufs, rows) for table, gs, gbufs, rows in self._state(plan)['tables']]



# ---------------------------------------------------------------------------------------------
# Hybrid sequence model (round 6): embedding tables striped by row, LSTM weights data-parallel.
# north_star: "partition the item-embedding table row-wise ... all-reduce for the dense LSTM weights";
# the reference's only device split is a TF tower list (lstm/run.py:87,221-229, lstm/seqModel.py:87-126).
# ---------------------------------------------------------------------------------------------
class _ShardView(object):
    """A striped table WITHOUT its padding row, as K7 sees it: keys that name the padding row (lookups this rank
    does not own, padded receive slots) are out of range and dropped by the key builders."""

    def __init__(self, t, rows):
        self.name, self.bias_name = t.name, t.bias_name
        self.E, self.acc = t.E[:rows], t.acc[:rows]
        self.bias = None if t.bias is None else t.bias[:rows]
        self.bias_acc = None if t.bias_acc is None else t.bias_acc[:rows]
        self.sites = []


class SeqHybridParallel(SeqDataParallel):
    """SeqModel on `world` ranks with every embedding table striped by ROW (owner = row % world, local row =
    row // world -- ShardedHMF's rule) and the dense parameters (lstm_w / lstm_b of every layer, w_input_*)
    replicated and all-reduced.  One step of the group == the single-process step on the global batch, like
    SeqDataParallel -- but no rank holds a whole table, no rank sorts the GLOBAL lookups, and the lookup traffic is
    all-to-all (rows travel once, to the rank that asked / owns) instead of an all-gather of every replica's rows:

      forward   per batch lookup (inputs, targets, users) and feature: ids -> owners (all_to_all, 4 B per lookup),
                the owner gathers its rows (K2 on its shard), rows + bias -> back (all_to_all, 4 (d + 1) B per
                lookup); the sampled pool: every rank gathers the pool rows it owns (zeros elsewhere), one
                all_reduce of [S, d + 1] makes the pool whole everywhere (S = 1024: 266 KB)
      backward  gradient rows of the batch lookups -> owners (all_to_all); the owner runs K7 over what it RECEIVED
                (keys = local rows) -- every row is updated once, on one rank, duplicates across ranks merged by
                that rank's pass; pool gradient [S, d] (+ bias): all_reduce, every owner applies ITS pool rows;
                per-unrolled-step pool gradients [L, S, d] (+ [L, S]): reduce_scatter -- they are only SQUARED
                (tf.clip_by_global_norm over the un-merged per-step matmul gradients, lstm/seqModel.py:178-182,
                SURVEY A.7): each rank squares its 1/world slice, the squares join the scalar all_reduce that the
                batch lookups' un-merged IndexedSlices norms already need
      dense     lstm_w / lstm_b / w_input_*: one packed all_reduce (128 KB at C4)

    Per rank and step at C4 (L = 50, B_loc = 1024, d = 64, S = 1024): 2 x 2 x 51 200 lookups x 260 B = 53 MB of
    all-to-all + 13.3 MB x (N - 1) / N of reduce_scatter + < 1 MB of small all-reduces -- against SeqDataParallel's
    all_gather of N x 26 MB of rows and a 13.3 MB all_reduce.  Routing (owner of every
    lookup, permutation, split sizes) is host work per batch on the ids the step is fed with: one D2H read of the
    ids per lookup node (data-loader work in a production loop, like ShardedHMF.prepare_route).

    Scope: training steps (step(..., forward_only=False)) of models whose trained lookups are one-hot,
    mean-combined features (the restriction SeqDataParallel has; C4); full-vocabulary evaluation over striped
    tables is not built -- global_params() reassembles the tables for a single-process model.  Steps run eagerly
    (collectives between the kernels).  No multi-GPU box was in reach of the builder: verified with gloo ranks
    sharing one GPU (tests/test_seq_hybrid_gpu.py) and on CPU for the routing (tests/test_dist_cpu.py)."""

    def __init__(self, model, group=None):
        super().__init__(model, group)
        self._feat = {}            # id(Feature) -> dict(full_h, route)
        self._fetch = {}           # (id(node), k) -> this step's exchange record
        self._hstate = {}
        self._shard_tables()

    # ---- striping -----------------------------------------------------------------------------
    def _shard_tables(self):
        W, r = self.world, self.rank
        for t in self.model.att_emb.tables.values():
            if getattr(t, 'shard', None) is not None:
                continue
            V, d = int(t.E.shape[0]), int(t.E.shape[1])
            rows = (V + W - 1) // W
            E = torch.zeros((rows + 1, d), dtype=torch.float32, device=t.E.device)
            mine = t.E[r::W]
            E[:mine.shape[0]].copy_(mine)
            acc = torch.full_like(E, float(t.acc.flatten()[0].item()) if t.acc.numel() else 0.1)
            acc[:mine.shape[0]].copy_(t.acc[r::W])
            bias = bias_acc = None
            if t.bias is not None:
                bias = torch.zeros((rows + 1,), dtype=torch.float32, device=t.E.device)
                bias[:mine.shape[0]].copy_(t.bias[r::W])
                bias_acc = torch.full_like(bias, 0.1)
                bias_acc[:mine.shape[0]].copy_(t.bias_acc[r::W])
            t.E, t.acc, t.bias, t.bias_acc = E, acc, bias, bias_acc
            t.shard = dict(V=V, rows=rows, zero_row=rows, count=int(mine.shape[0]))
            t.view = _ShardView(t, rows)
            for a in ('aux_cnt', 'aux_first'):
                if hasattr(t, a):
                    delattr(t, a)

    @staticmethod
    def route_rows(full_rows, world):
        """Host side of one lookup exchange: (order, send_rows, send_counts) for global table rows `full_rows` --
        lookups grouped by owner (stable: the order inside a group is the order of the lookups), rows as LOCAL rows
        of their owner."""
        full_rows = np.asarray(full_rows, dtype=np.int64)
        owner = full_rows % world
        order = np.argsort(owner, kind='stable')
        return (order.astype(np.int32), (full_rows[order] // world).astype(np.int32),
                np.bincount(owner, minlength=world).astype(np.int64))

    def _feature(self, f):
        """The map of a feature re-expressed for a striped table: kept on the host (entity -> global row) for the
        routing, and on the device as entity -> LOCAL row or the padding row (pool-like lookups: every rank looks the
        same ids up and contributes the rows it owns)."""
        st = self._feat.get(id(f))
        if st is None:
            t = f.table
            V = t.shard['V']
            full = f.maps[0]
            full_h = np.arange(V, dtype=np.int64) if full is None else full.cpu().numpy().astype(np.int64)
            if full_h.size and (full_h.min() < 0 or full_h.max() >= V):
                raise ValueError("SeqHybridParallel: feature map of %s leaves the table" % t.name)
            route = np.where(full_h % self.world == self.rank, full_h // self.world, t.shard['zero_row'])
            st = dict(full_h=full_h, route=torch.from_numpy(route.astype(np.int32)).to(t.E.device), f=f)
            self._feat[id(f)] = st
        return st

    def _exchange_counts(self, send_counts):
        W = self.world
        sc = torch.as_tensor(send_counts, dtype=torch.int64)
        rc = torch.empty(W, dtype=torch.int64)
        if W > 1:
            be = dist.get_backend(self.group)
            if be == 'nccl':
                scd, rcd = sc.to(self.rt.device), rc.to(self.rt.device)
                dist.all_to_all_single(rcd, scd, group=self.group)
                rc = rcd.cpu()
            else:
                dist.all_to_all_single(rc, sc, group=self.group)
        else:
            rc.copy_(sc)
        return [int(x) for x in sc.tolist()], [int(x) for x in rc.tolist()]

    def _a2a(self, out, inp, out_splits, in_splits):
        if self.world == 1:
            out.copy_(inp)
            return
        if out.is_cuda and dist.get_backend(self.group) != 'nccl':      # (the gloo rig of the tests: host-staged)
            o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(o, i, out_splits, in_splits, group=self.group)
            out.copy_(o)
            return
        dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)

    def _reduce_host_staged(self, t):
        if self.world > 1:
            if t.is_cuda and dist.get_backend(self.group) != 'nccl':
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    # ---- forward: the lookups ---------------------------------------------------------------------
    def _pool_nodes(self, plan):
        from . import graph as G
        pools = set()
        for n in plan.order:
            if isinstance(n, G.Prediction):
                pools.add(id(n.inputs[1]))
        return pools

    def fetch(self, plan):
        """Plan._execute, in front of the lookups: every EntityEmbed node of the plan is served here."""
        from . import graph as G, ops
        if not plan.train:
            raise NotImplementedError("SeqHybridParallel: training steps only (evaluate a single-process model built "
                                      "from global_params())")
        pools = self._pool_nodes(plan)
        dev = self.rt.device
        done = set()
        for n in plan.order:
            if not isinstance(n, G.EntityEmbed):
                continue
            if n.concat or any(f.kind != 'cat' for f in n.feats):
                raise NotImplementedError("SeqHybridParallel: one-hot, mean-combined features only")
            n.alloc_value()
            F = len(n.feats)
            if id(n) in pools:
                # every rank: the pool rows it owns (the padding row elsewhere), summed over the ranks
                for k, f in enumerate(n.feats):
                    fs = self._feature(f)
                    t = f.table
                    ops.gather_onehot(t.E, t.bias if n.with_bias else None, fs['route'], n.inputs[0].value, n.value,
                                      scale=n.out_scale / F, accumulate=k > 0,
                                      bias_out=n.bias_value if n.with_bias else None)
                self._reduce_host_staged(n.value)
                if n.with_bias:
                    self._reduce_host_staged(n.bias_value)
                done.add(id(n))
                continue
            ids_h = n.inputs[0].value.cpu().numpy()
            for k, f in enumerate(n.feats):
                fs = self._feature(f)
                t = f.table
                d = int(t.E.shape[1])
                order, send_rows, sc = self.route_rows(fs['full_h'][ids_h], self.world)
                sc, rc = self._exchange_counts(sc)
                R = sum(rc)
                nloc = int(ids_h.shape[0])
                rec = self._fetch.get((id(n), k))
                if rec is None or rec['cap'] < R:
                    cap = max(R, nloc + nloc // 2 + 64) if self.world > 1 else nloc
                    rec = dict(cap=cap, recv_rows=torch.empty(cap, dtype=torch.int32, device=dev),
                               rows=torch.empty((cap, d), dtype=torch.float32, device=dev),
                               rows_b=torch.empty((cap,), dtype=torch.float32, device=dev),
                               got=torch.empty((nloc, d), dtype=torch.float32, device=dev),
                               got_b=torch.empty((nloc,), dtype=torch.float32, device=dev),
                               send=torch.empty((nloc, d), dtype=torch.float32, device=dev),
                               send_b=torch.empty((nloc,), dtype=torch.float32, device=dev),
                               garena=torch.zeros((cap, d), dtype=torch.float32, device=dev),
                               garena_b=torch.zeros((cap,), dtype=torch.float32, device=dev), gen=0)
                    rec['gen'] = (self._fetch[(id(n), k)]['gen'] + 1) if (id(n), k) in self._fetch else 0
                    self._fetch[(id(n), k)] = rec
                    self._hstate.pop(id(plan), None)          # (new buffers: the K7 proxies are rebuilt)
                inv = np.empty_like(order)
                inv[order] = np.arange(order.shape[0], dtype=np.int32)
                rec.update(sc=sc, rc=rc, R=R, order=torch.from_numpy(order).to(dev), inv=torch.from_numpy(inv).to(dev))
                rec['recv_rows'].fill_(t.shard['zero_row'])
                self._a2a(rec['recv_rows'][:R], torch.from_numpy(send_rows).to(dev), rc, sc)      # ids -> owners
                wb = n.with_bias and t.bias is not None
                if R:
                    ops.gather_onehot(t.E, t.bias if wb else None, None, rec['recv_rows'][:R], rec['rows'][:R],
                                      bias_out=rec['rows_b'][:R] if wb else None)
                self._a2a(rec['got'], rec['rows'][:R], sc, rc)                                    # rows -> back
                if wb:
                    self._a2a(rec['got_b'], rec['rows_b'][:R], sc, rc)
                # got is in owner order: lookup i sits at got[inv[i]]
                ops.gather_onehot(rec['got'], rec['got_b'] if wb else None, None, rec['inv'], n.value,
                                  scale=n.out_scale / F, accumulate=k > 0,
                                  bias_out=n.bias_value if n.with_bias else None)
            done.add(id(n))
        return done

    # ---- backward: gradients to the owners ----------------------------------------------------------
    def _state(self, plan):
        from . import graph as G
        st = self._hstate.get(id(plan))
        if st is not None:
            return st
        rt = self.rt
        dev = rt.device
        pools, preds = set(), []
        for n in plan.order:
            if isinstance(n, G.Prediction) and n.inputs[1].train_tables:
                pools.add(id(n.inputs[1]))
                preds.append(n)
        tables = []
        for table, sites, bufs, total in plan.tables:
            if any(s.kind != 'cat' or s.col_off != 0 for s in sites):
                raise NotImplementedError("SeqHybridParallel: one-hot, mean-combined features only")
            width = sites[0].node.shape[1]
            gs, rows = [], 0
            for s in sites:
                k = next(i for i, f in enumerate(s.node.feats) if f.table is s.table)
                if id(s.node) in pools:
                    n_ = s.n
                    route = self._feature(s.node.feats[k])['route']
                    gs.append((s, n_, 'pool', route, None))
                else:
                    rec = self._fetch[(id(s.node), k)]
                    gs.append((s, rec['cap'], 'batch', None, rec))
                rows += gs[-1][1]
            arena = torch.zeros((rows, width), dtype=torch.float32, device=dev)
            arena_b = torch.zeros((rows,), dtype=torch.float32, device=dev)
            out, r0 = [], 0
            for s, n_, kind, route, rec in gs:
                if kind == 'pool':
                    g = _GSite(s, s.ids_node.value, _GNode(r0, arena, arena_b, s.node), n_)
                    g.maps = (route,)
                else:
                    g = _GSite(s, rec['recv_rows'], _GNode(r0, arena, arena_b, s.node), n_)
                    g.maps = (None,)
                g.table = table.view
                out.append((s, g, kind, rec))
                r0 += n_
            gbufs = {'keys': torch.full((rows,), G.KEY_NONE, dtype=torch.int32, device=dev),
                     'src': torch.zeros((rows,), dtype=torch.int32, device=dev),
                     'coef': torch.zeros((rows,), dtype=torch.float32, device=dev),
                     'hot': torch.zeros((rows // 16 + 4,), dtype=torch.int32, device=dev)}
            off = 0
            for _, g, _, _ in out:
                g.key_off = off
                off += g.cap
            tables.append((table.view, out, gbufs, rows))
        # per-step pool gradients: only their squares are needed -> reduce_scatter (see the class docstring)
        steps = []
        for n in preds:
            if getattr(n, 'C_steps', None) is not None:
                for buf in (n.C_steps, n.rs_steps):
                    tot = buf.numel()
                    dd = int(buf.shape[-1]) if buf.dim() == 3 else 1
                    per = (tot + self.world - 1) // self.world
                    per = (per + dd - 1) // dd * dd            # slices start on a row (the norm's row scales)
                    steps.append(dict(buf=buf, per=per,
                                      pad=torch.zeros(per * self.world, dtype=torch.float32, device=dev),
                                      mine=torch.zeros(per, dtype=torch.float32, device=dev)))
        st = dict(preds=preds, tables=tables, steps=steps)
        self._hstate[id(plan)] = st
        return st

    def step_slices(self, buf):
        """(this rank's summed slice of a per-step pool-gradient buffer, first element, count) after exchange()."""
        for st in self._hstate.values():
            for e in st['steps']:
                if e['buf'] is buf:
                    lo = self.rank * e['per']
                    cnt = max(0, min(e['per'], buf.numel() - lo))
                    return e['mine'], lo, cnt
        return None

    def exchange(self, plan):
        from . import ops
        rt = self.rt
        st = self._state(plan)
        # dense gradients + the pool gradient and its bias (small): one packed all_reduce
        pack = [p.grad for p in rt.dense.values() if getattr(p, 'touched', False)]
        for n in st['preds']:
            if not n._grad_written:
                continue
            pool = n.inputs[1]
            pack.append(pool.grad)
            if pool.bias_grad_used:
                pack.append(pool.bias_grad)
        self._all_reduce_packed([t for t in pack])
        # per-step pool gradients: summed slices
        for e in st['steps']:
            flat = e['buf'].reshape(-1)
            if self.world == 1:
                e['mine'][:flat.numel()].copy_(flat)
                continue
            e['pad'][:flat.numel()].copy_(flat)
            if e['pad'].is_cuda and dist.get_backend(self.group) != 'nccl':
                h = e['pad'].cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)       # (gloo has no reduce_scatter)
                e['mine'].copy_(h[self.rank * e['per']:(self.rank + 1) * e['per']])
            else:
                dist.reduce_scatter_tensor(e['mine'], e['pad'], op=dist.ReduceOp.SUM, group=self.group)
        # lookups: gradient rows to the owners of the rows
        for table, gs, gbufs, rows in st['tables']:
            for s, g, kind, rec in gs:
                node = s.node
                g.node._grad_written = node._grad_written
                g.node.bias_grad_used = node.bias_grad_used
                if not node._grad_written:
                    continue
                a = g.node.arena[g.node.row0:g.node.row0 + g.n]
                ab = g.node.arena_b[g.node.row0:g.node.row0 + g.n]
                src = node.arena[node.row0:node.row0 + s.n]
                src_b = node.arena_b[node.row0:node.row0 + s.n]
                if kind == 'pool':
                    a.copy_(src)                      # (all-reduced above: the pool gradient of the global batch)
                    if node.bias_grad_used:
                        ab.copy_(src_b)
                    continue
                R, sc, rc = rec['R'], rec['sc'], rec['rc']
                ops.gather_onehot(src, src_b if node.bias_grad_used else None, None, rec['order'], rec['send'],
                                  bias_out=rec['send_b'] if node.bias_grad_used else None)       # owner order
                a.zero_()
                self._a2a(a[:R], rec['send'], rc, sc)
                if node.bias_grad_used:
                    ab.zero_()
                    self._a2a(ab[:R], rec['send_b'], rc, sc)

    def gathered_tables(self, plan):
        return [(view, [g for _, g, _, _ in gs], gbufs, rows) for view, gs, gbufs, rows in self._state(plan)['tables']]

    # ---- tables back together (tests, checkpoints) ----------------------------------------------------
    def global_params(self, slots=False):
        """{reference variable name: numpy array} of the WHOLE tables (att_emb.get_params() of a single-process
        model): the shards of all ranks, interleaved.  O(table) traffic -- not a step-path call."""
        W = self.world
        out = {}
        for t in self.model.att_emb.tables.values():
            sh = t.shard
            for name, loc in ((t.name, t.acc if slots else t.E), (t.bias_name, None if t.bias is None else
                                                                   (t.bias_acc if slots else t.bias))):
                if loc is None or name is None:
                    continue
                x = loc[:sh['rows']].detach().cpu().contiguous()
                parts = [torch.empty_like(x) for _ in range(W)]
                if W > 1:
                    dist.all_gather(parts, x, group=self.group)
                else:
                    parts = [x]
                full = torch.stack(parts, 1).reshape((sh['rows'] * W,) + tuple(x.shape[1:]))[:sh['V']]
                a = full.numpy()
                out[name] = a.reshape(-1, 1) if a.ndim == 1 else a
        return out

def _hash_u32(x, salt):
    x = (x.to(torch.int64) * 2654435761 + salt) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 2246822519) & 0xFFFFFFFF
    x = ((x ^ (x >> 13)) * 3266489917) & 0xFFFFFFFF
    return x ^ (x >> 16)


def _zipf_items(n, n_items, gen, dev):
    """Zipf-like global item ids: rank = floor(n_items * u^6), scattered by a hash."""
    u = torch.rand(n, device=dev, generator=gen)
    rk = torch.clamp((u.pow(6.0) * n_items).to(torch.int64), max=n_items - 1)
    return (_hash_u32(rk, 12345) % n_items).to(torch.int32)


def draw_global_pool(sampler, S, group=None):
    """ONE weighted draw without replacement of S items over an item set whose weights are sharded over
    the ranks of `group` (utils/prepare_train.py:7-17 draws the pool from one distribution): every rank
    races its own shard (sampler.sample_with_keys: exponential keys -ln(u)/w, weights on a common
    scale, independent seeds), the ranks exchange their S smallest (key, id) pairs -- 8 KB each -- and
    every rank keeps the S smallest keys of the union: the S smallest keys of the whole item set, the
    single-process draw.  Identical on every rank (stable order: key, then rank, then position)."""
    ids, keys = sampler.sample_with_keys(S)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        allk = torch.empty(world * S, dtype=keys.dtype, device=keys.device)
        alli = torch.empty(world * S, dtype=ids.dtype, device=ids.device)
        dist.all_gather_into_tensor(allk, keys.contiguous(), group=group)
        dist.all_gather_into_tensor(alli, ids.contiguous(), group=group)
        if allk.is_cuda and world * S <= 16384 and allk.dtype == torch.float32 and alli.dtype == torch.int32:
            # (round 5: one rank-selection launch over the 64-bit (key, position) words instead of torch's sort +
            # index kernels -- SURVEY section 7: no torch arithmetic on the path)
            from . import ops
            ids = torch.empty(S, dtype=alli.dtype, device=alli.device)
            ops.merge_keyed_take(allk, alli, S, ids)
        else:          # (the gloo / CPU rig of the tests, or more pairs than the kernel's LDS list holds)
            ids = alli[torch.argsort(allk, stable=True)[:S]]
    if int(ids.min().item()) < 0:
        raise ValueError("draw_global_pool(%d): fewer items with a positive weight in all shards together" % S)
    return ids


def _time_collective(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = torch.tensor([e0.elapsed_time(e1) / iters], dtype=torch.float64, device=torch.device('cuda', torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


XGMI_LINK_GBS = 153.0     # per direction and link, 7 links per GPU


def comm_roofline(model, world, grp=None):
    """The step's collectives alone, at the step's payloads (every rank calls it): achieved GB/s per
    rank against what its xGMI links allow.  A rank moves (N-1)/N of a payload P over N-1 links in an
    all_gather / all_to_all, 2 (N-1)/N in an all_reduce: with every link busy the floor is
    P / N / 153 GB/s (twice that for the all_reduce)."""
    if world == 1:
        return {"note": "one rank: every exchange is a local copy, no xGMI traffic"}
    S, dp, B_loc, cap = model.S, model.dp, model.B_loc, max(model.cap, 4)
    out = {}
    x_pack, x_gath = model.I_pack[:cap], model.I_gath[:world * cap]
    rows = torch.zeros((B_loc, dp), dtype=torch.float32, device=model.device)
    back = torch.zeros_like(rows)
    g = torch.zeros((S, dp), dtype=torch.float32, device=model.device)
    cases = [
        ("all_gather_pool_blocks", lambda: dist.all_gather_into_tensor(x_gath, x_pack, group=grp),
         cap * dp * 4 * world, 1.0),
        ("all_to_all_target_rows", lambda: _all_to_all(back, rows, group=grp), B_loc * dp * 4, 1.0),
        ("all_reduce_pool_grads", lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM, group=grp), S * dp * 4, 2.0),
    ]
    for name, fn, payload, factor in cases:
        ms = _time_collective(fn)
        wire = payload * factor * (world - 1) / world               # bytes this rank sends (= receives)
        floor_ms = wire / (world - 1) / (XGMI_LINK_GBS * 1e9) * 1e3   # all N-1 links busy
        out[name] = {"payload_bytes": payload, "ms": ms, "achieved_gbs_per_rank": wire / ms / 1e6,
                     "peak_gbs_per_rank": XGMI_LINK_GBS * (world - 1), "frac": floor_ms / ms}
    return out


def comm_prediction(mode, world, B_loc, S, d, n_tokens=0, L=0, dense_bytes=0):
    """What the step's collectives cost at link rate for a world of `world` ranks -- ARITHMETIC, not a measurement
    (the bench line carries it as `roofline_comm_predicted`): a rank moves (N-1)/N of a payload
    over N-1 links of XGMI_LINK_GBS each in an all_gather / all_to_all / reduce_scatter, twice that in an
    all_reduce; collectives below ~1 MB are latency-bound (tens of us under RCCL) whatever this floor says."""
    N = max(int(world), 1)
    dp = d + 4
    link = XGMI_LINK_GBS * 1e9
    f = (N - 1) / N if N > 1 else 0.0

    def row(name, kind, payload):
        wire = payload * (2.0 if kind == 'all_reduce' else 1.0) * f
        us = wire / max(N - 1, 1) / link * 1e6 if N > 1 else 0.0
        return {"collective": name, "kind": kind, "payload_bytes": int(payload), "wire_bytes_per_rank": int(wire),
                "us_at_link_rate": us}
    if mode == 'id':
        rows = [row("pool blocks", 'all_gather', S * dp * 4), row("target rows", 'all_to_all', B_loc * dp * 4),
                row("target-row gradients", 'all_to_all', B_loc * dp * 4), row("pool gradients", 'all_reduce', S * dp * 4)]
    elif mode == 'id_logits':        # ShardedHMF(exchange='logits'): the logits cross, not the pool rows
        B = B_loc * N
        rows = [row("latents of the global batch", 'all_gather', B * dp * 4),
                row("target rows", 'all_to_all', B_loc * dp * 4),
                row("partial logits [B, S / N] -> [B_loc, S]", 'all_to_all', B_loc * S * 4),
                row("logit gradients back", 'all_to_all', B_loc * S * 4),
                row("target-row gradients", 'all_to_all', B_loc * dp * 4),
                row("latent-gradient partials", 'reduce_scatter', B * d * 4)]
    elif mode == 'rep_tokens':
        rows = [row("pool blocks", 'all_gather', S * dp * 4), row("target rows", 'all_to_all', B_loc * dp * 4),
                row("target-row gradients", 'all_to_all', B_loc * dp * 4), row("pool gradients", 'all_reduce', S * dp * 4),
                row("merged token gradient + bias", 'all_reduce', n_tokens * (d + 1) * 4)]
    elif mode in ('seq_hybrid', 'seq_dp'):
        # the sequence model (C4: L unrolled steps, B_loc sequences per rank, d = 64): lookups per rank and step =
        # 2 L B_loc (inputs + targets), rows of (d + 1) floats; per-step pool gradients [L, S, d + 1]
        n_look = 2 * L * B_loc
        if mode == 'seq_hybrid':        # SeqHybridParallel: tables striped by row
            rows = [row("lookup ids to the owners", 'all_to_all', n_look * 4),
                    row("lookup rows back", 'all_to_all', n_look * (d + 1) * 4),
                    row("pool rows", 'all_reduce', S * (d + 1) * 4),
                    row("lookup-gradient rows to the owners", 'all_to_all', n_look * (d + 1) * 4),
                    row("per-step pool gradients (squared only)", 'reduce_scatter', L * S * (d + 1) * 4),
                    row("dense gradients + pool gradient", 'all_reduce', dense_bytes + S * (d + 1) * 4)]
        else:                           # SeqDataParallel: every table on every rank
            rows = [row("dense + per-step pool gradients", 'all_reduce', dense_bytes + (L + 1) * S * (d + 1) * 4),
                    row("lookup ids of all replicas", 'all_gather', N * n_look * 4),
                    row("lookup-gradient rows of all replicas", 'all_gather', N * n_look * (d + 1) * 4)]
    else:           # token-striped bags: partials of the GLOBAL batch
        B = B_loc * N
        rows = [row("pool partials", 'all_reduce', S * dp * 4), row("target partials", 'reduce_scatter', B * dp * 4),
                row("pool-gradient partials", 'all_reduce', S * dp * 4), row("target-row gradients", 'all_gather', B * dp * 4)]
    return {"model": "%d ranks, %d xGMI links x %.0f GB/s per direction each; arithmetic, not measured" % (N, max(N - 1, 1), XGMI_LINK_GBS),
            "exchanges": rows, "us_total_at_link_rate": sum(r["us_at_link_rate"] for r in rows)}


def bench_run(args, world, rank, local_rank, init_pg=True):
    """The N-rank bench body (every rank calls it); returns the JSON dict on rank 0, None elsewhere.
    world == 1 runs the very same sharded step on one GPU (all "exchanges" local): the anchor of
    the weak-scaling curve -- same code path, same 100 M-item table, same eager launches."""
    # ARX_DIST_BACKEND=gloo + ARX_DIST_ONE_GPU=1: a rig without a second GPU (this repo's test box) runs the
    # N > 1 code path with every rank on device 0 and gloo underneath -- for checking the path, not for numbers
    one_gpu = bool(os.environ.get("ARX_DIST_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if init_pg:
        backend = os.environ.get("ARX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    B_loc, S, d = args.batch, args.n_sampled, args.dim
    t_setup = time.time()
    rep_tokens = bool(getattr(args, 'sharded_rep_tokens', False))
    with_bags = bool(getattr(args, 'sharded_bags', False)) or rep_tokens
    if with_bags:
        # HET items: a multi-hot attribute of 20 tokens over a 100 k-token table striped by token
        n_tok, L_bag = 100000, 20
        rng = np.random.default_rng(11)                      # the SAME bag index on every rank
        p_tok = 1.0 / np.arange(1, n_tok + 1)
        vals = rng.choice(n_tok, size=(args.n_items + 1) * L_bag, p=p_tok / p_tok.sum()).astype(np.int32)
        lens = np.full(args.n_items + 1, L_bag, dtype=np.int32)
        starts = (np.arange(args.n_items + 1, dtype=np.int64) * L_bag).astype(np.int32)
        cls = ShardedHMFRepTokens if rep_tokens else ShardedHMFBags
        model = cls(args.n_users, args.n_items, d, B_loc, S, 0.1, rank, world, dev, (vals, starts, lens), n_tok, seed=0)
    else:
        model = ShardedHMF(args.n_users, args.n_items, d, B_loc, S, 0.1, rank, world, dev, seed=0,
                           exchange=getattr(args, 'exchange', 'rows'))
    pred_mode = 'rep_tokens' if rep_tokens else ('bags' if with_bags else
                                                 ('id_logits' if getattr(model, 'exchange', 'rows') == 'logits' else 'id'))
    gen = torch.Generator(device=dev)
    gen.manual_seed(77 + rank)
    n_pos = 20
    nu = model.nu_loc
    ptr = (torch.arange(nu + 2, device=dev, dtype=torch.int64) * n_pos).clamp(max=nu * n_pos).to(torch.int32)
    pos_items = _zipf_items(nu * n_pos, args.n_items, gen, dev)
    model.set_positives(ptr, pos_items)
    total = args.steps + args.warmup
    nb = min(total, 32)
    batches = []
    route_ms = []      # host cost of routing one batch (prepare_route: D2H of the ids, argsort by owner, counts, ids to the owners)
    for _ in range(nb):
        lu = torch.randint(0, nu, (B_loc,), device=dev, generator=gen)
        k = torch.randint(0, n_pos, (B_loc,), device=dev, generator=gen)
        users = (lu * world + rank).to(torch.int32)
        items = pos_items[(lu * n_pos + k)]
        torch.cuda.synchronize()
        t_r = time.time()
        batches.append(model.prepare_route(users, items))      # data-loader side: order by owner
        torch.cuda.synchronize()
        route_ms.append((time.time() - t_r) * 1e3)
    # Shared negative pool: ONE draw of S items without replacement with p ~ count^0.5 over ALL items
    # (prepare_train.py:7-35 sample_items over item_frequency, run_hmf.py:62 power = 0.5): every rank
    # races its own shard on device (arx_sample_wor_keys), the ranks exchange their S best (key, id)
    # pairs (8 KB each) and keep the S smallest keys of the union (draw_global_pool) -- inside the
    # timed region, every n_resample steps.  Setup: global interaction counts of the owned items (one
    # reduce_scatter).
    from .utils.prepare_train import DeviceSampler
    rows = (args.n_items + world - 1) // world
    from . import ops as _ops
    cnt = torch.zeros(rows * world, dtype=torch.int32, device=dev)
    _ops.item_frequency(pos_items, rows * world, cnt)             # arx_item_frequency: counts only
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)                 # (set-up; 4 B per item)
    mine = cnt.view(rows, world)[:, rank].contiguous()
    del cnt
    tot = mine.sum(dtype=torch.int64).to(torch.float64)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    wts = (mine.to(torch.float64) / tot).pow(0.5).to(torch.float32)
    own = torch.arange(rows, device=dev, dtype=torch.int64) * world + rank
    wts[own >= args.n_items] = 0.0
    sampler = DeviceSampler(own.to(torch.int32), wts, device=dev, seed=4242 + rank)
    def redraw():
        # ONE draw of S items over all shards (draw_global_pool): the reference's sampling law at any N
        model.set_pool(draw_global_pool(sampler, S))
    torch.cuda.synchronize()
    dist.barrier()
    setup_s = time.time() - t_setup
    redraws = [0]

    import contextlib

    def run(k0, k1):
        # the whole loop (redraws, steps) on the model's own stream: no stream joins around the steps
        own = getattr(model, 'stream', None)
        with (torch.cuda.stream(own) if own is not None else contextlib.nullcontext()):
            for k in range(k0, k1):
                if k == 0 or (k >= args.warmup and (k - args.warmup) % args.n_resample == 0):
                    redraw()
                    redraws[0] += k >= args.warmup
                model.step(batches[k % nb])

    run(0, args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    run(args.warmup, total)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.time() - t0
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    loss = float(model.read_loss().item())
    comm = comm_roofline(model, world) if not with_bags else None
    # roofline of the dominant kernel (the local scorer GEMM [B_loc, S] x d), HIP events on the
    # stream the kernel runs on; same definition as the single-GPU bench line
    out = None
    if rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pool_mat = model.P_part if (with_bags and not rep_tokens) else model.I_all
        # the kernel the timed step RUNS (round 4 verdict, weak #6): with the fused scorer that is k_sc_hinge of
        # MwScorer.fwd (phase 2: target score + scorer GEMM + hinge epilogue, no logits), launched here with the step's
        # own buffers; only where the step itself takes the materialising GEMM (shapes the family does not cover,
        # ARX_SCORER_F32, the bag variants) is that GEMM the one timed
        fused = (rep_tokens or not with_bags) and getattr(model, 'scorer', None) is not None and model._fused_scorer()
        if fused:
            Sg_, arena, arena_b = model.Sg, model.arena, model.arena_b
            dT_ = arena[B_loc + Sg_:B_loc + Sg_ + B_loc] if world == 1 else model.dT_pack
            dt_ = arena_b[B_loc + Sg_:B_loc + Sg_ + B_loc] if world == 1 else dT_[:, d]
            run_gemm = lambda: model.scorer.fwd(model.U_loc, model.I_all[:, :d], model.b_all, model.T_pack[:, :d],
                                                model.T_pack[:, d], model.urows, model.pos_ptr, model.pos_items,
                                                model.item2slot, model.bl, model.t_loc, dt_, arena[:B_loc, :d],
                                                dT_[:, :d], 1.0 / (B_loc * world), phases=2)
        else:
            run_gemm = lambda: model.be.gemm(model.U_loc, pool_mat[:, :d], model.logits, transB=True,
                                             col_bias=model.b_all)
        for _ in range(5):
            run_gemm()
        e0.record()
        for _ in range(50):
            run_gemm()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        flops = 2.0 * B_loc * S * d
        bx6 = fused or ((not _ops.SCORER_F32) and d in (64, 128) and S % 128 == 0 and B_loc >= 4096)
        peak = 2500.0 / 6.0 if bx6 else 157.3
        roofline = {"kernel": ("k_sc_hinge (MwScorer.fwd phase 2: target score + scorer GEMM + hinge epilogue, 6 exact bf16 terms "
                               "per f32 product term; the kernel the sharded step runs; per rank)" if fused else
                               "logits GEMM on the bf16 pipe, 6 exact bf16 terms per f32 product term (k_nt_bx6; per rank)"
                               if bx6 else "gemm_logits_nt (f32-input MFMA; per rank)"),
                    "bound": "mfma", "achieved": flops / ms / 1e9,
                    "peak": peak, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / peak, "traffic": None,
                    "flops_per_launch": flops, "ms_per_launch": ms,
                    "peak_note": "2MNK f32 flops against 2500 TF dense bf16 / 6 terms" if bx6 else "f32-input MFMA peak"}
        B = B_loc * world
        out = {
            "metric": "training interactions/sec + sampled-negatives/sec, dim-128, 1/2/4/8 MI355X",
            "value": B * args.steps / wall, "unit": "interactions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("C3 sharded: HET items (id + 20-token bag over a 100 k-token table REPLICATED on every rank; "
                                    "per step the id-only exchanges + ONE all_reduce of the merged token gradient [n_tokens, d], "
                                    "arx.dist.ShardedHMFRepTokens) -- " if rep_tokens else
                                    "C3 sharded: HET items (id + 20-token bag over a 100 k-token table striped by TOKEN; "
                                    "per step all_reduce(pool partials, pool grads) + reduce_scatter(target partials) + "
                                    "all_gather(target grads), arx.dist.ShardedHMFBags) -- " if with_bags else "") +
                                   "C5 (BASELINE configs[4]): synthetic %d-item/%d-user HMF, dim %d, id-only, WMRB 'mw', "
                                   "item and user tables row-sharded over %d GPU(s) (owner = id %% N), %d shared "
                                   "negatives/step = ONE draw over all shards with p ~ count^0.5 (per-rank races on device, "
                                   "the S best keys of the union) every %d steps (%d redraw(s) inside the timed region); "
                                   "per step: RCCL all_gather(pool rows, blocks padded to the largest owner count) + "
                                   "all_to_all(target rows, target grads) + all_reduce(pool grads); B_loc=%d per GPU.  The batch ROUTING (order by owner, "
                                   "count / id exchange: data-loader work, ShardedHMF.prepare_route) is done once per "
                                   "batch of the %d-batch ring, OUTSIDE the timed loop"
                                   % (args.n_items, args.n_users, d, world, S, args.n_resample, redraws[0], B_loc, nb),
                       "batch_per_gpu": B_loc, "global_batch": B, "n_sampled": S, "dim": d,
                       "parallelism": "row-sharded tables x dp%d" % world,
                       "exchange": getattr(model, 'exchange', 'rows'),
                       "routing_in_timed_region": False,
                       # the data loader's share, stated next to it: host milliseconds per batch of B_loc interactions
                       # (median over the ring's batches; one host thread, includes the tiny count exchange)
                       "routing_host_ms_per_batch": float(np.median(route_ms)) if route_ms else None,
                       "pool_redraws_timed": redraws[0],
                       "hipgraph_segments": (sorted(model._graphs) if model.use_graphs else None),
                       "step_form": ("hipGraph segments between the collectives" if model.use_graphs else "eager launches"),
                       "hipgraph_captures": model.n_captures, "hipgraph_replays": model.n_replays,
                       "sampled_negative_logits_per_s": B * S * args.steps / wall,
                       "final_loss": loss, "setup_s": setup_s},
            "roofline": roofline, "roofline_comm": comm,
            # the same exchanges priced at link rate for the world of this run and for the 8-GPU node of BASELINE
            # configs[4] (arithmetic: no multi-GPU box was in reach of the builder)
            "roofline_comm_predicted": {
                "this_run": comm_prediction(pred_mode, world, B_loc, S, d, n_tokens=100000 if with_bags else 0),
                "at_8_ranks": comm_prediction(pred_mode, 8, B_loc, S, d, n_tokens=100000 if with_bags else 0)},
            "cpu_baseline": {"value": None, "unit": "interactions/s", "cores": None, "kind": "port",
                             "sample": None,
                             "why": "timed on rank 0 at N = 1 only (bench contract): the N = 1 line of the same run "
                                    "carries the reference-algorithm restatement on this box's host cores"},
        }
    del model
    torch.cuda.empty_cache()
    return out


def bench_main(args, world, rank, local_rank):
    """(bench.py::main_sharded is the entry the driver uses: it adds the N = 1 anchor to the line.)"""
    out = bench_run(args, world, rank, local_rank)
    dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush it out first so that the JSON
        # line is the LAST line on stdout
        import ctypes
        import sys
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
