"""Row-sharded HMF step over RCCL / xGMI.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The
item embedding table (+ bias, + Adagrad slots) is striped row-wise over the
ranks (owner = item % world); the user table is striped the same way and every
rank trains on interactions of ITS users (pure data parallelism on that side).
One step with B = world * B_loc interactions and a pool of S shared negatives
(S/world owned by each rank) is algebraically the single-process step of
hmf_model.py on the global batch; the exchanges are

  all_gather   U_loc [B_loc,d]  -> U [B,d]            user latents
  (local)      P_g = U . I_g^T + b_g   [B, S/world]    partial negative logits
  all_to_all   P_g -> logits_loc [B_loc, S]            <- the negative-sample logits
  reduce_scatter target scores t_g [B] -> t_loc [B_loc] (owner computes u_r . I[target_r])
  (local)      WMRB loss fwd+bwd on [B_loc, S]
  all_to_all   dlogits back to the owners  -> dP_g [B, S/world]
  all_gather   dt_loc -> dt [B]
  (local)      dI_g = dP_g^T . U (+ targets) -> scatter + sparse Adagrad on the shard
  reduce_scatter dU partials [B,d] -> dU_loc [B_loc,d] -> Adagrad on the user shard

No table gradient ever crosses xGMI.  All exchanges are all-to-all shaped, so
every one of the 7 links of a GPU is used concurrently; there is no ring
all-reduce on the path (the dense LSTM weights of the sequence model would use
one, 128 KB, latency-bound).

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


class HipBackend(object):
    """Compute stages on libarx.so (include/arx.h)."""

    def __init__(self, device):
        from . import ops
        self.ops = ops
        self.ws = ops.Workspace(device)

    def gather_rows(self, E, bias, rows, out, bias_out):
        self.ops.gather_onehot(E, bias, None, rows, out, bias_out=bias_out)

    def gemm(self, A, B, C, transA=False, transB=False, beta=0.0, col_bias=None, a_rowsum=None):
        self.ops.gemm(A, B, C, self.ws, transA=transA, transB=transB, beta=beta, col_bias=col_bias,
                      a_rowsum=a_rowsum)

    def dot_score(self, U, T, tb, out):
        self.ops.dot_score(U, T, tb, out)

    def dot_score_bwd(self, U, T, ds, dU, acc, dT):
        self.ops.dot_score_bwd(U, T, ds, dU, acc, dT)

    def copy_2d(self, src, dst):
        self.ops.copy_2d(src, dst)

    def shard_route(self, ids, world, rank, zero_row, rows_out, keys_out):
        self.ops.shard_route(ids, world, rank, zero_row, rows_out, keys_out)

    def loss_mw_pos(self, logits, t, urows, ptr, items, i2s, bl, dl, dt, gscale):
        self.ops.loss_mw_pos(logits, t, urows, ptr, items, i2s, bl, dl, dt, gscale)

    def sum_scaled(self, x, scale, out):
        self.ops.sum_scaled(x, scale, out)

    def sparse_adagrad(self, E, acc, bias, bias_acc, keys, G, Gb, lr):
        self.ops.sparse_adagrad(E, acc, bias, bias_acc, keys, None, None, G, Gb, lr, self.ws)

    def slot_map_set(self, m, ids, clear):
        self.ops.slot_map_set(m, ids, clear=clear)

    def copy_i32(self, src, dst):
        dst.copy_(src, non_blocking=True)


class ShardedHMF(object):
    """id-only HMF ('mw' loss) with row-sharded tables.  Global ids everywhere in
    the API; `users` passed to step() must all be owned by this rank."""

    def __init__(self, n_users, n_items, d, B_loc, S, learning_rate, rank, world, device,
                 backend=None, group=None, tables=None, seed=0, acc0=0.1):
        if S % world != 0:
            raise ValueError("n_sampled must be divisible by the world size (stratified pool)")
        if (S // world) % 4 != 0 or d % 4 != 0:
            raise ValueError("S/world and d must be multiples of 4")
        self.n_users, self.n_items, self.d = n_users, n_items, d
        self.B_loc, self.S, self.Sg = B_loc, S, S // world
        self.rank, self.world = rank, world
        self.B = B_loc * world
        self.device = torch.device(device)
        self.group = group
        self.be = backend if backend is not None else HipBackend(self.device)
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

        B, Sg, W = self.B, self.Sg, world
        z = lambda *s: torch.zeros(s, dtype=f32, device=dev)
        zi = lambda *s: torch.zeros(s, dtype=i32, device=dev)
        self.users_in, self.items_in = zi(B_loc), zi(B_loc)
        self.urows = zi(B_loc)
        self.U_loc, self.U = z(B_loc, d), z(B, d)
        self.tg_all, self.tg_rows, self.tg_keys = zi(B), zi(B), zi(B)
        self.pool_ids = zi(S)                               # owner-major global ids
        self.pool_rows = zi(Sg)                             # local rows of the owned block
        self.pool_old = None
        self.item2slot = torch.full((n_items + 1,), -1, dtype=i32, device=dev)
        self.I_g, self.b_g = z(Sg, d), z(Sg)
        self.P_g = z(B, Sg)                                 # partial logits, rank-major rows
        self.blk = z(W, B_loc, Sg)                          # all-to-all landing / staging
        self.logits = z(B_loc, S)
        self.T_g, self.tb_g, self.t_g = z(B, d), z(B), z(B)
        self.t_loc, self.dt_loc, self.dt = z(B_loc), z(B_loc), z(B)
        self.bl, self.loss_part, self.loss = z(B_loc), z(1), z(1)
        self.dlogits = z(B_loc, S)
        self.dP_g = z(B, Sg)
        self.G_item = z(Sg + B, d)                          # [dI_g ; dT_g]
        self.Gb_item = z(Sg + B)
        self.keys_item = torch.full((Sg + B,), 0x7FFFFFFF, dtype=i32, device=dev)
        self.dU, self.dU_loc = z(B, d), z(B_loc, d)
        self.pos_ptr = zi(nu + 2)
        self.pos_items = zi(1)
        self.steps = 0

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
        """Stratified shared pool: pool_ids[g*Sg:(g+1)*Sg] must be owned by rank g
        (embed_attribute.py:320-348 update_sampled, sharded)."""
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
        mine = self.pool_ids[self.rank * self.Sg:(self.rank + 1) * self.Sg]
        be.shard_route(mine, self.world, self.rank, self.zero_row, self.pool_rows, None)

    # ------------------------------------------------------------------- step
    def step(self, users, items):
        be, W, r = self.be, self.world, self.rank
        B, B_loc, S, Sg, d = self.B, self.B_loc, self.S, self.Sg, self.d
        grp = self.group
        be.copy_i32(users.to(self.device, torch.int32) if isinstance(users, torch.Tensor) else
                    torch.as_tensor(np.asarray(users, dtype=np.int32)).to(self.device), self.users_in)
        be.copy_i32(items.to(self.device, torch.int32) if isinstance(items, torch.Tensor) else
                    torch.as_tensor(np.asarray(items, dtype=np.int32)).to(self.device), self.items_in)
        # user side: local rows (all owned), gather, all_gather
        be.shard_route(self.users_in, W, r, 0, self.urows, None)
        be.gather_rows(self.E_user, None, self.urows, self.U_loc, None)
        dist.all_gather_into_tensor(self.U, self.U_loc, group=grp)
        dist.all_gather_into_tensor(self.tg_all, self.items_in, group=grp)
        # owned pool block -> partial logits for ALL rows
        be.gather_rows(self.E_item, self.b_item, self.pool_rows, self.I_g, self.b_g)
        be.gemm(self.U, self.I_g, self.P_g, transB=True, col_bias=self.b_g)
        # all_to_all: rank-major row blocks of P_g -> owner-major column blocks
        dist.all_to_all_single(self.blk.view(-1), self.P_g.view(-1), group=grp)
        for g in range(W):
            be.copy_2d(self.blk[g], self.logits[:, g * Sg:(g + 1) * Sg])
        # targets: the owner scores u_r . I[target_r] + b, everybody else contributes 0
        be.shard_route(self.tg_all, W, r, self.zero_row, self.tg_rows, self.tg_keys)
        be.gather_rows(self.E_item, self.b_item, self.tg_rows, self.T_g, self.tb_g)
        be.dot_score(self.U, self.T_g, self.tb_g, self.t_g)
        dist.reduce_scatter_tensor(self.t_loc, self.t_g, op=dist.ReduceOp.SUM, group=grp)
        # loss (global mean => gscale = 1/B)
        be.loss_mw_pos(self.logits, self.t_loc, self.urows, self.pos_ptr, self.pos_items,
                       self.item2slot, self.bl, self.dlogits, self.dt_loc, 1.0 / B)
        # backward exchanges
        for g in range(W):
            be.copy_2d(self.dlogits[:, g * Sg:(g + 1) * Sg], self.blk[g])
        dist.all_to_all_single(self.dP_g.view(-1), self.blk.view(-1), group=grp)
        dist.all_gather_into_tensor(self.dt, self.dt_loc, group=grp)
        # dU partial = dP_g . I_g + dt * T_g ; dT_g = dt * U
        be.gemm(self.dP_g, self.I_g, self.dU)
        be.dot_score_bwd(self.U, self.T_g, self.dt, self.dU, True, self.G_item[Sg:])
        # dI_g = dP_g^T . U, bias gradient = row sums
        be.gemm(self.dP_g, self.U, self.G_item[:Sg], transA=True, a_rowsum=self.Gb_item[:Sg])
        be.copy_i32(self.dt, self.Gb_item[Sg:])
        be.copy_i32(self.pool_rows, self.keys_item[:Sg])
        be.copy_i32(self.tg_keys, self.keys_item[Sg:])
        be.sparse_adagrad(self.E_item, self.A_item, self.b_item, self.Ab_item, self.keys_item,
                          self.G_item, self.Gb_item, self.lr)
        dist.reduce_scatter_tensor(self.dU_loc, self.dU, op=dist.ReduceOp.SUM, group=grp)
        be.sparse_adagrad(self.E_user, self.A_user, None, None, self.urows, self.dU_loc, None, self.lr)
        self.steps += 1

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


# --------------------------------------------------------------------------
# bench entry for N > 1 (driver: python -m torch.distributed.run ... bench.py --gpus N)
# --------------------------------------------------------------------------
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


def bench_main(args, world, rank, local_rank):
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=dev)
    B_loc, S, d = args.batch, args.n_sampled, args.dim
    t_setup = time.time()
    model = ShardedHMF(args.n_users, args.n_items, d, B_loc, S, 0.1, rank, world, dev, seed=0)
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
    for _ in range(nb):
        lu = torch.randint(0, nu, (B_loc,), device=dev, generator=gen)
        k = torch.randint(0, n_pos, (B_loc,), device=dev, generator=gen)
        users = (lu * world + rank).to(torch.int32)
        items = pos_items[(lu * n_pos + k)]
        batches.append((users, items))
    # stratified shared pools, identical on every rank
    pg = torch.Generator(device=dev)
    pg.manual_seed(4242)
    n_pools = total // args.n_resample + 2
    Sg = S // world
    pools = []
    for _ in range(n_pools):
        blocks = []
        for g in range(world):
            ng = (args.n_items - g + world - 1) // world
            loc = torch.randperm(ng, device=dev, generator=pg)[:Sg]
            blocks.append((loc * world + g).to(torch.int32))
        pools.append(torch.cat(blocks))
    torch.cuda.synchronize()
    dist.barrier()
    setup_s = time.time() - t_setup

    def run(k0, k1):
        for k in range(k0, k1):
            if k % args.n_resample == 0:
                model.set_pool(pools[k // args.n_resample])
            u, i = batches[k % nb]
            model.step(u, i)

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
    if rank == 0:
        B = B_loc * world
        out = {
            "metric": "training interactions/sec + sampled-negatives/sec, dim-128, 1/2/4/8 MI355X",
            "value": B * args.steps / wall, "unit": "interactions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C5-style: synthetic %d-item/%d-user HMF, dim %d, id-only, WMRB 'mw', item "
                                   "and user tables row-sharded over %d GPUs (owner = id %% N), %d shared "
                                   "negatives/step (S/N per owner), RCCL all_gather(U) + all_to_all(logits, "
                                   "dlogits) + reduce_scatter(t, dU); B_loc=%d per GPU"
                                   % (args.n_items, args.n_users, d, world, S, B_loc),
                       "batch_per_gpu": B_loc, "global_batch": B, "n_sampled": S, "dim": d,
                       "parallelism": "row-sharded tables x dp%d" % world,
                       "sampled_negative_logits_per_s": B * S * args.steps / wall,
                       "final_loss": loss, "setup_s": setup_s},
            "roofline": None, "cpu_baseline": None,
        }
        print(json.dumps(out))
    dist.destroy_process_group()
