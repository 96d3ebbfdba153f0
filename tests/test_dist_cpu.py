"""world_size-2, -3 and -4 gloo tests of the row-sharded HMF step (arx.dist.ShardedHMF):
the exchange / routing logic with a numpy compute double must reproduce the
single-process oracle step on the global batch (loss and updated tables) -- including
steps whose targets are spread unevenly over the owners and a step in which some ranks
own NO target row at all (R = 0: empty all_to_all blocks, empty gather / scatter sites)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir, exchange='rows'):
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arx.dist import ShardedHMF
    from arx.utils.synthetic import SyntheticHMF
    from numpy_backend import NumpyBackend
    from oracle import ref_graph as rg

    n_users, n_items, d, B_loc, S = 60, 90, 16, 8, 16
    syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=6)
    params = syn.glorot_params(d, seed=2, scale=0.5)
    tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
              'item_bias': params['item_bias_cat_0'][2:]}
    model = ShardedHMF(n_users, n_items, d, B_loc, S, 0.5, rank, world, 'cpu',
                       backend=NumpyBackend(), tables=tables, exchange=exchange)
    # positives CSR over this rank's local user rows (global item ids)
    own_users = np.arange(rank, n_users, world)
    ptr = np.zeros(len(own_users) + 2, dtype=np.int32)
    items = []
    for k, u in enumerate(own_users):
        its = syn.pos_items[syn.pos_ptr[u]:syn.pos_ptr[u + 1]]
        items.extend(its.tolist())
        ptr[k + 1] = len(items)
    ptr[-1] = ptr[-2]
    model.set_positives(ptr, np.asarray(items, dtype=np.int32))

    B = B_loc * world
    i2l = syn.item_ind2logit_ind_dict()
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, i2l, syn.logit_ind2item_ind,
                                   loss_function='mw', n_sampled=S, params=params, dtype=np.float64)
    pos = syn.positives_dict()
    ref.prepare_warp(pos, pos)
    rng = np.random.default_rng(5)          # identical stream on both ranks
    for step in range(5):
        pool = None
        if step % 2 == 0:
            # the pool is ONE draw over all items (prepare_train.py:7-17): any owners, in any order --
            # step 0 the usual mixed case, step 2 every pool item on ONE owner (the other ranks' blocks
            # are all padding), step 4 a skewed mix
            if step == 0:
                pool = rng.choice(n_items, size=S, replace=False)
            elif step == 2:
                pool = rng.choice(np.arange(world - 1, n_items, world), size=S, replace=False)
            else:
                hot = rng.choice(np.arange(0, n_items, world), size=(3 * S) // 4, replace=False)
                rest = rng.choice(np.setdiff1d(np.arange(n_items), hot), size=S - len(hot), replace=False)
                pool = rng.permutation(np.concatenate([hot, rest]))
            pool = pool.astype(np.int32)
            id2idx = {int(v): i for i, v in enumerate(pool)}
            model.set_pool(pool)
        gu, gi = [], []
        for g in range(world):              # every rank draws users it owns
            lu = rng.integers(0, len(np.arange(g, n_users, world)), size=B_loc)
            users = lu * world + g
            k = rng.integers(0, syn.n_pos, size=B_loc)
            gu.append(users)
            gi.append(syn.pos_items[syn.pos_ptr[users] + k])
        gu[0][1] = gu[0][0]                 # duplicate user / target rows
        gi[1][2] = gi[1][3]
        if step == 0:
            gi[0][0] = pool[np.nonzero(pool % world == 1)[0][0]]     # a target that is also a pool slot (owned by rank 1)
        if step == 3:                       # every target owned by rank 0: R = 0 on all other ranks
            for g in range(world):
                gi[g] = (rng.integers(0, n_items // world, size=B_loc) * world).astype(gi[g].dtype)
        if step == 4:                       # uneven: three quarters of the targets on the last owner
            for g in range(world):
                k_ = (3 * B_loc) // 4
                gi[g][:k_] = rng.integers(0, (n_items - (world - 1) + world - 1) // world - 1, size=k_) * world + (world - 1)
        l_ref = ref.step(np.concatenate(gu).tolist(), np.concatenate(gi).tolist(), pool, id2idx,
                         loss='mw')
        model.step(gu[rank].astype(np.int32), gi[rank].astype(np.int32))
        l_got = float(model.read_loss().item())
        assert abs(l_got - l_ref) <= 1e-5 * abs(l_ref), (step, l_got, l_ref)
    got = model.gather_global_tables()
    np.testing.assert_allclose(got['user'], ref.att_emb.params['userembed_cat_0'][2:], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got['item'], ref.att_emb.params['itemembed_cat_0'][2:], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got['item_bias'], ref.att_emb.params['item_bias_cat_0'][2:, 0],
                               rtol=1e-4, atol=1e-6)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_step_matches_oracle_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 400) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_step_logits_exchange_matches_oracle_gloo(tmp_path, world):
    """exchange='logits' (north_star's all-to-all of the negative-sample logits): the same
    five steps -- mixed / one-owner / skewed pools, R = 0 ranks, duplicate rows -- against the same oracle."""
    import torch.multiprocessing as mp
    port = 29950 + (os.getpid() % 400) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), 'logits'), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def _dp_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arx.dist import SeqDataParallel
    dp = object.__new__(SeqDataParallel)          # the exchange helpers alone (no model / GPU runtime)
    dp.world, dp.rank, dp.group = world, rank, None
    # packed all-reduce: every tensor ends as the sum over replicas, shapes and views preserved
    big = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4) * (rank + 1)
    view = torch.zeros(10)
    part = view[2:7]
    part.copy_(torch.full((5,), float(rank + 1)))
    scalar = torch.tensor([0.5 * (rank + 1)])
    dp._all_reduce_packed([big, part, scalar])
    tot = sum(r + 1 for r in range(world))
    assert torch.equal(big, torch.arange(24, dtype=torch.float32).reshape(2, 3, 4) * tot)
    assert torch.equal(view, torch.tensor([0, 0] + [float(tot)] * 5 + [0, 0, 0]))
    assert abs(float(scalar) - 0.5 * tot) < 1e-6
    # rank-major gather of lookup ids and gradient rows
    ids = torch.arange(3, dtype=torch.int32) + 10 * rank
    rows = torch.full((3, 2), float(rank))
    g_ids = torch.empty(3 * world, dtype=torch.int32)
    g_rows = torch.empty(3 * world, 2)
    dp._all_gather(ids, g_ids)
    dp._all_gather(rows, g_rows)
    assert g_ids.tolist() == [k + 10 * r for r in range(world) for k in range(3)]
    assert torch.equal(g_rows, torch.repeat_interleave(torch.arange(world, dtype=torch.float32), 3)[:, None].expand(-1, 2))
    dp._loss = torch.zeros(1)
    assert abs(dp.global_loss(1.5 + rank) - sum(1.5 + r for r in range(world))) < 1e-6
    with open(os.path.join(out_dir, "dp%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_seq_data_parallel_collectives_gloo(tmp_path, world):
    """The exchange helpers of arx.dist.SeqDataParallel over gloo (the model-level test against the
    oracle needs the HIP runtime: tests/test_seq_dp_gpu.py)."""
    import torch.multiprocessing as mp
    port = 29100 + (os.getpid() % 400) + world
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("dp%d" % r)) for r in range(world))


def _bags_worker(rank, world, port, out_dir, replicated=False):
    """Multi-hot item table on the sharded step: HET items = mean(id row, bag mean), id table striped by item and
    the token table striped by token (arx.dist.ShardedHMFBags) or REPLICATED with one all-reduce of its merged
    gradient (arx.dist.ShardedHMFRepTokens, round 5); vs the single-process oracle."""
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arx.dist import ShardedHMFBags, ShardedHMFRepTokens
    from arx.utils.synthetic import SyntheticHMF
    from numpy_backend import NumpyBackend
    from oracle import ref_graph as rg

    n_users, n_items, d, B_loc, S, V = 60, 90, 16, 8, 16, 37
    syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=6,
                       item_mulhot=True, mulhot_vocab=V, avg_len=4, max_len=9)
    ia = syn.i_attr
    n_tok = ia._embedding_classes_list_mulhot[0]
    params = syn.glorot_params(d, seed=2, scale=0.5)
    tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
              'item_bias': params['item_bias_cat_0'][2:], 'token': params['itemembed_mulhot_0'],
              'token_bias': params['item_bias_mulhot_0']}
    bags = (np.asarray(ia.features_mulhot[0]), np.asarray(ia.mulhot_starts[0]), np.asarray(ia.mulhot_lengths[0]))
    cls = ShardedHMFRepTokens if replicated else ShardedHMFBags
    model = cls(n_users, n_items, d, B_loc, S, 0.5, rank, world, 'cpu', bags, n_tok,
                backend=NumpyBackend(), tables=tables)
    own_users = np.arange(rank, n_users, world)
    ptr = np.zeros(len(own_users) + 2, dtype=np.int32)
    items = []
    for k, u in enumerate(own_users):
        items.extend(syn.pos_items[syn.pos_ptr[u]:syn.pos_ptr[u + 1]].tolist())
        ptr[k + 1] = len(items)
    ptr[-1] = ptr[-2]
    model.set_positives(ptr, np.asarray(items, dtype=np.int32))
    B = B_loc * world
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                   syn.logit_ind2item_ind, loss_function='mw', n_sampled=S, params=params,
                                   dtype=np.float64)
    pos = syn.positives_dict()
    ref.prepare_warp(pos, pos)
    rng = np.random.default_rng(5)
    for step in range(4):
        pool = None
        if step % 2 == 0:
            blocks = [rng.choice(np.arange(g, n_items, world), size=S // world + (1 if g < S % world else 0),
                                 replace=False) for g in range(world)]
            pool = np.concatenate(blocks).astype(np.int32)
            id2idx = {int(v): i for i, v in enumerate(pool)}
            model.set_pool(pool)
        gu, gi = [], []
        for g in range(world):
            users = rng.integers(0, len(np.arange(g, n_users, world)), size=B_loc) * world + g
            gu.append(users)
            gi.append(syn.pos_items[syn.pos_ptr[users] + rng.integers(0, syn.n_pos, size=B_loc)])
        gu[0][1] = gu[0][0]
        gi[1][2] = gi[1][3]                        # duplicate targets: merged before the token stage
        if step == 0:
            gi[0][0] = pool[S // world]            # a target that is also a pool slot
        if step == 3:                              # every target's id row owned by rank 0
            for g in range(world):
                gi[g] = (rng.integers(0, n_items // world, size=B_loc) * world).astype(gi[g].dtype)
        l_ref = ref.step(np.concatenate(gu).tolist(), np.concatenate(gi).tolist(), pool, id2idx, loss='mw')
        model.step(gu[rank].astype(np.int32), gi[rank].astype(np.int32))
        l_got = float(model.read_loss().item())
        assert abs(l_got - l_ref) <= 1e-5 * abs(l_ref), (step, l_got, l_ref)
    got = model.gather_global_tables()
    P = ref.att_emb.params
    for name, want in (('user', P['userembed_cat_0'][2:]), ('item', P['itemembed_cat_0'][2:]),
                       ('item_bias', P['item_bias_cat_0'][2:, 0]), ('token', P['itemembed_mulhot_0']),
                       ('token_bias', P['item_bias_mulhot_0'][:, 0])):
        np.testing.assert_allclose(got[name], want, rtol=1e-4, atol=1e-6, err_msg=name)
    with open(os.path.join(out_dir, "bags%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_token_sharded_bags_match_oracle_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 400) + world
    mp.spawn(_bags_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("bags%d" % r)) for r in range(world))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_replicated_token_table_matches_oracle_gloo(tmp_path, world):
    """ShardedHMFRepTokens: id table striped, token table replicated, its merged gradient all-reduced (round 5)."""
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 400) + world
    mp.spawn(_bags_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("bags%d" % r)) for r in range(world))


def _pool_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arx.dist import draw_global_pool
    n_items, S = 997, 64
    rng = np.random.default_rng(123)                 # the same weights and uniforms on every rank
    w = (1.0 / np.arange(1, n_items + 1) ** 1.2)
    w[rng.integers(0, n_items, 50)] = 0.0

    class Shard(object):
        """Stand-in for DeviceSampler.sample_with_keys on this rank's rows (id % world == rank): the
        exponential race arx_sample_wor_keys runs on device, here in numpy."""
        def __init__(self):
            self.draw = 0

        def sample_with_keys(self, n):
            u = np.random.default_rng(1000 + self.draw).random(n_items)       # one uniform per ITEM, any world
            self.draw += 1
            keys = np.where(w > 0, -np.log(u) / np.maximum(w, 1e-300), np.inf)
            mine = np.arange(rank, n_items, world)
            order = mine[np.argsort(keys[mine], kind='stable')][:n]
            ids = np.full(n, -1, dtype=np.int32)
            ks = np.full(n, np.inf, dtype=np.float32)
            live = np.isfinite(keys[order])
            ids[:live.sum()] = order[live]
            ks[:live.sum()] = keys[order][live]
            return torch.from_numpy(ids), torch.from_numpy(ks)
    sh = Shard()
    for draw in range(3):
        pool = draw_global_pool(sh, S).numpy()
        u = np.random.default_rng(1000 + draw).random(n_items)
        keys = np.where(w > 0, -np.log(u) / np.maximum(w, 1e-300), np.inf).astype(np.float32)
        want = np.argsort(keys, kind='stable')[:S]                              # the single-process draw
        np.testing.assert_array_equal(np.sort(pool), np.sort(want))
        np.testing.assert_array_equal(keys[pool], np.sort(keys[want]))          # ... in draw order
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_global_pool_draw_equals_single_process_draw_gloo(tmp_path, world):
    """arx.dist.draw_global_pool: per-rank races over the owned rows + the S best keys of the union ==
    the S smallest keys over ALL items (the reference's one draw, utils/prepare_train.py:7-17), the
    same pool on every rank -- not S/N items per owner."""
    import torch.multiprocessing as mp
    port = 29950 + (os.getpid() % 40) + world
    mp.spawn(_pool_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


# ---------------------------------------------------------------------------------------------------------------------
# SeqHybridParallel (round 6): the routing and the two lookup exchanges of the striped-table sequence model, on CPU
# tensors over gloo -- ids to the owners, rows back, gradient rows to the owners -- against plain indexing of the whole
# table (the kernels on either side of the collectives are the HIP gathers, covered by tests/test_seq_hybrid_gpu.py).
# ---------------------------------------------------------------------------------------------------------------------
def test_hybrid_route_rows_groups_by_owner_stably():
    from arx.dist import SeqHybridParallel
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        rows = rng.integers(0, 1000, size=257)
        order, send_rows, counts = SeqHybridParallel.route_rows(rows, world)
        assert counts.sum() == len(rows) and len(counts) == world
        owner = rows[order] % world
        assert np.all(np.diff(owner) >= 0)                             # grouped by owner, owners ascending
        for g in range(world):                                         # stable inside a group
            idx = order[owner == g]
            assert np.all(np.diff(idx) > 0)
        np.testing.assert_array_equal(send_rows.astype(np.int64) * world + owner, rows[order])
    order, send_rows, counts = SeqHybridParallel.route_rows(np.zeros(0, dtype=np.int64), 4)
    assert len(order) == 0 and counts.tolist() == [0, 0, 0, 0]


def _hybrid_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arx.dist import SeqHybridParallel

    class _Rt(object):
        device = torch.device('cpu')
    h = SeqHybridParallel.__new__(SeqHybridParallel)                    # the exchange helpers only: no model, no GPU
    h.world, h.rank, h.group, h.rt = world, rank, None, _Rt()
    V, d = 101, 6
    rng = np.random.default_rng(3)                                     # the same table on every rank
    table = rng.standard_normal((V, d)).astype(np.float32)
    rows_loc = (V + world - 1) // world
    shard = np.zeros((rows_loc + 1, d), dtype=np.float32)
    shard[:len(table[rank::world])] = table[rank::world]
    for step in range(4):
        r2 = np.random.default_rng(100 * step + rank)                  # every rank looks other rows up
        n = [0, 17, 64, 5][(step + rank) % 4]                          # ... ragged, incl. NO lookup at all on a rank
        full = r2.integers(0, V, size=n)
        if step == 2:
            full[:] = 7                                                # every lookup of every rank on ONE owner
        order, send_rows, sc = h.route_rows(full, world)
        sc, rc = h._exchange_counts(sc)
        R = sum(rc)
        recv_rows = torch.empty(R, dtype=torch.int32)
        h._a2a(recv_rows, torch.from_numpy(send_rows), rc, sc)         # ids -> owners
        assert R == 0 or int(recv_rows.max()) < rows_loc
        rows = torch.from_numpy(shard)[recv_rows.long()]               # the owner's gather
        got = torch.empty((n, d), dtype=torch.float32)
        h._a2a(got, rows, sc, rc)                                      # rows -> back
        inv = np.empty_like(order)
        inv[order] = np.arange(n, dtype=np.int32)
        np.testing.assert_array_equal(got.numpy()[inv], table[full])   # == a lookup in the whole table
        # backward: gradient rows to the owners; the owners' scatter-adds together == the global one
        G = r2.standard_normal((n, d)).astype(np.float32)
        recv_g = torch.zeros((R, d), dtype=torch.float32)
        h._a2a(recv_g, torch.from_numpy(G[order]), rc, sc)
        mine = np.zeros((rows_loc, d), dtype=np.float64)
        np.add.at(mine, recv_rows.numpy(), recv_g.numpy().astype(np.float64))
        objs = [None] * world
        dist.all_gather_object(objs, (full, G))
        ref = np.zeros((V, d), dtype=np.float64)
        for f_, g_ in objs:
            np.add.at(ref, f_, g_.astype(np.float64))
        np.testing.assert_allclose(mine[:len(ref[rank::world])], ref[rank::world], rtol=1e-6, atol=1e-6)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_hybrid_lookup_exchanges_equal_whole_table_lookup(tmp_path, world):
    import torch.multiprocessing as mp
    port = 29100 + (os.getpid() % 400) + 7 * world
    mp.spawn(_hybrid_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))
