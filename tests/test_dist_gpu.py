"""ShardedHMF on the real HIP backend (libarx.so) with a 1-rank RCCL group: the
same code path bench.py runs per rank at N>1, checked against the oracle."""
import os

import numpy as np
import pytest

from oracle import ref_graph as rg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,S", [(32, 64), (128, 128)])     # (the second shape is one the bf16-pipe scorer takes)
@pytest.mark.parametrize("graphs", [True, False, 'logits'])
def test_sharded_hip_backend_world1(dev, graphs, B, S):
    """graphs=True: the step is ONE hipGraph (ShardedHMF._step_static: eager on step 0, captured on step
    1, replayed from step 2 on -- through two pool redraws and fresh batches).  'logits': the step with the
    all-to-all-of-logits exchange (_step_logits; at world 1 its transposes, block gathers and per-owner GEMMs)."""
    exchange = 'logits' if graphs == 'logits' else 'rows'
    graphs = graphs is True
    import torch
    import torch.distributed as dist
    from arx.dist import ShardedHMF
    from arx.utils.synthetic import SyntheticHMF
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29733")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n_users, n_items, d = 300, 500, 64
        syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=8)
        params = syn.glorot_params(d, seed=2, scale=0.5)
        tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
                  'item_bias': params['item_bias_cat_0'][2:]}
        model = ShardedHMF(n_users, n_items, d, B, S, 0.5, 0, 1, dev, tables=tables, graphs=graphs, exchange=exchange)
        assert model.use_graphs == graphs
        ptr = np.concatenate([syn.pos_ptr[:n_users + 1], [syn.pos_ptr[n_users]]]).astype(np.int32)
        model.set_positives(ptr, syn.pos_items)
        ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                       syn.logit_ind2item_ind, loss_function='mw', n_sampled=S,
                                       params=params, dtype=np.float64)
        pos = syn.positives_dict()
        ref.prepare_warp(pos, pos)
        rng = np.random.default_rng(3)
        for step in range(6):
            pool = None
            if step in (0, 2, 4):
                pool = syn.sample_pool(S, rng)
                id2idx = {int(v): i for i, v in enumerate(pool)}
                model.set_pool(pool)
            users, items = syn.sample_batch(B, rng)
            l_ref = ref.step(list(users), list(items), pool, id2idx, loss='mw')
            model.step(users, items)
            l_got = float(model.read_loss().item())
            np.testing.assert_allclose(l_got, l_ref, rtol=1e-4)
        assert (model._graph_key is not None and set(model._graphs) == {'step'}) == graphs
        got = model.gather_global_tables()
        np.testing.assert_allclose(got['user'], ref.att_emb.params['userembed_cat_0'][2:], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(got['item'], ref.att_emb.params['itemembed_cat_0'][2:], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(got['item_bias'], ref.att_emb.params['item_bias_cat_0'][2:, 0], rtol=1e-4,
                                   atol=2e-6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("replicated", [False, True])
@pytest.mark.parametrize("n_users,n_items,V,d,B,S", [(300, 500, 120, 64, 32, 64), (3000, 4000, 900, 32, 2048, 256)])
def test_token_sharded_bags_hip_backend_world1(dev, n_users, n_items, V, d, B, S, replicated):
    """ShardedHMFBags (HET items: id table striped by item, token table striped by token) and ShardedHMFRepTokens
    (token table replicated, its merged gradient all-reduced: round 5) on the HIP backend with a 1-rank RCCL group
    vs the oracle; the second shape is past the rank-sort limits (radix sort + window apply in both K7 passes)."""
    import torch
    import torch.distributed as dist
    from arx.dist import ShardedHMFBags, ShardedHMFRepTokens
    from arx.utils.synthetic import SyntheticHMF
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29734")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=8,
                           item_mulhot=True, mulhot_vocab=V, avg_len=5, max_len=12)
        ia = syn.i_attr
        n_tok = ia._embedding_classes_list_mulhot[0]
        params = syn.glorot_params(d, seed=2, scale=0.5)
        tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
                  'item_bias': params['item_bias_cat_0'][2:], 'token': params['itemembed_mulhot_0'],
                  'token_bias': params['item_bias_mulhot_0']}
        bags = (np.asarray(ia.features_mulhot[0]), np.asarray(ia.mulhot_starts[0]), np.asarray(ia.mulhot_lengths[0]))
        cls = ShardedHMFRepTokens if replicated else ShardedHMFBags
        model = cls(n_users, n_items, d, B, S, 0.5, 0, 1, dev, bags, n_tok, tables=tables)
        ptr = np.concatenate([syn.pos_ptr[:n_users + 1], [syn.pos_ptr[n_users]]]).astype(np.int32)
        model.set_positives(ptr, syn.pos_items)
        ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                       syn.logit_ind2item_ind, loss_function='mw', n_sampled=S,
                                       params=params, dtype=np.float64)
        pos = syn.positives_dict()
        ref.prepare_warp(pos, pos)
        rng = np.random.default_rng(3)
        for step in range(5):              # eager, captured, replayed x 3 (round 6: both classes run hipGraph segments)
            pool = None
            if step in (0, 2):
                pool = syn.sample_pool(S, rng)
                id2idx = {int(v): i for i, v in enumerate(pool)}
                model.set_pool(pool)
            users, items = syn.sample_batch(B, rng)
            l_ref = ref.step(list(users), list(items), pool, id2idx, loss='mw')
            model.step(users, items)
            np.testing.assert_allclose(float(model.read_loss().item()), l_ref, rtol=1e-4)
        got = model.gather_global_tables()
        P = ref.att_emb.params
        for name, want in (('user', P['userembed_cat_0'][2:]), ('item', P['itemembed_cat_0'][2:]),
                           ('item_bias', P['item_bias_cat_0'][2:, 0]), ('token', P['itemembed_mulhot_0']),
                           ('token_bias', P['item_bias_mulhot_0'][:, 0])):
            np.testing.assert_allclose(got[name], want, rtol=1e-4, atol=2e-6, err_msg=name)
    finally:
        dist.destroy_process_group()


def test_sharded_c5_shape_world1(dev):
    """BASELINE configs[4] at its full shape on ONE rank (the driver's box has one GPU): 100 M items x
    dim 128 row-sharded table (51 GB + 51 GB of Adagrad slots), 1 M users, B = 16384, S = 1024, 'mw',
    through the sharded step (1-rank RCCL group: every exchange is a local copy; step 0 eager, step 1 captured
    into one hipGraph, step 2 replayed).  Three consecutive
    steps; before each, the rows the step will touch are read back from the device and the step is
    restated in fp64 numpy (oracle.ref_embed's 'mw' arithmetic: scorer, WMRB, rank-one gradients,
    duplicates merged, one Adagrad update per row) -- loss, every touched user / item row, item bias
    and their slots at rtol 1e-4; a sample of untouched rows stays bit-identical."""
    import torch
    import torch.distributed as dist
    from arx.dist import ShardedHMF
    from oracle import ref_embed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29735")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n_users, n_items, d, B, S, lr = 1000000, 100000000, 128, 16384, 1024, 0.1
        model = ShardedHMF(n_users, n_items, d, B, S, lr, 0, 1, dev, seed=5)
        rng = np.random.default_rng(11)
        hot = rng.integers(0, n_items, size=300)                        # popular items: duplicate targets
        watch = torch.from_numpy(rng.integers(0, n_items, size=4096)).to(dev)

        def rows(t, idx):
            return t[torch.from_numpy(idx).to(dev)].double().cpu().numpy()
        for step in range(3):
            users = rng.choice(n_users, size=B, replace=True).astype(np.int32)
            items = np.where(rng.random(B) < 0.4, hot[rng.integers(0, len(hot), size=B)],
                             rng.integers(0, n_items, size=B)).astype(np.int32)
            if step == 0:
                pool = np.unique(np.concatenate([items[:8], hot[:16], rng.integers(0, n_items, size=2 * S)]))
                pool = rng.permutation(pool)[:S].astype(np.int32)
                assert len(pool) == S
                model.set_pool(pool)
            ui, ii = np.unique(users), np.unique(np.concatenate([items, pool]))
            Eu, Au = rows(model.E_user, ui), rows(model.A_user, ui)
            Ei, Ai = rows(model.E_item, ii), rows(model.A_item, ii)
            bi, Abi = rows(model.b_item, ii), rows(model.Ab_item, ii)
            keep = watch[~torch.isin(watch, torch.from_numpy(ii).to(dev))]
            w_before = model.E_item[keep].clone()
            # ---- the step in fp64 on those rows (oracle/ref_embed.py EmbedSpaceHMF.step) ----
            pu, pp, pt = np.searchsorted(ui, users), np.searchsorted(ii, pool), np.searchsorted(ii, items)
            U, P, T = Eu[pu], Ei[pp], Ei[pt]
            logits = U @ P.T + bi[pp]
            t = (U * T).sum(1) + bi[pt]
            bl, dl, dt = ref_embed._Model.sampled_loss('mw', logits, t, np.ones((B, S), dtype=bool))
            dl, dt = dl / B, dt / B
            gU = np.zeros_like(Eu)
            np.add.at(gU, pu, dl @ P + dt[:, None] * T)
            gI, gb = np.zeros_like(Ei), np.zeros_like(bi)
            np.add.at(gI, pp, dl.T @ U)
            np.add.at(gb, pp, dl.sum(0))
            np.add.at(gI, pt, dt[:, None] * U)
            np.add.at(gb, pt, dt)
            model.step(users, items)
            np.testing.assert_allclose(float(model.read_loss().item()), float(bl.mean()), rtol=1e-4)
            for name, E, A, E0, A0, g, idx in (('user', model.E_user, model.A_user, Eu, Au, gU, ui),
                                               ('item', model.E_item, model.A_item, Ei, Ai, gI, ii),
                                               ('item_bias', model.b_item, model.Ab_item, bi, Abi, gb, ii)):
                a1 = A0 + g * g
                w1 = E0 - lr * g / np.sqrt(a1)
                np.testing.assert_allclose(rows(A, idx), a1, rtol=1e-4, atol=1e-9, err_msg='%s slots, step %d' % (name, step))
                np.testing.assert_allclose(rows(E, idx), w1, rtol=1e-4, atol=2e-7, err_msg='%s rows, step %d' % (name, step))
            assert torch.equal(model.E_item[keep], w_before)              # rows outside (batch u pool): untouched
    finally:
        dist.destroy_process_group()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_rank_worker(rank, world, port, out_dir, graphs=True, exchange='rows'):
    import sys
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from arx.dist import ShardedHMF, draw_global_pool
    from arx.utils.prepare_train import DeviceSampler
    from arx.utils.synthetic import SyntheticHMF

    n_users, n_items, d, B_loc, S = 301, 503, 64, 32, 60             # S not a multiple of the world size
    syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=8)
    params = syn.glorot_params(d, seed=2, scale=0.5)
    tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
              'item_bias': params['item_bias_cat_0'][2:]}
    model = ShardedHMF(n_users, n_items, d, B_loc, S, 0.5, rank, world, dev, tables=tables, graphs=graphs,
                       exchange=exchange)
    own_users = np.arange(rank, n_users, world)
    ptr = np.zeros(len(own_users) + 2, dtype=np.int32)
    its = []
    for k, u in enumerate(own_users):
        its.extend(syn.pos_items[syn.pos_ptr[u]:syn.pos_ptr[u + 1]].tolist())
        ptr[k + 1] = len(its)
    ptr[-1] = ptr[-2]
    model.set_positives(ptr, np.asarray(its, dtype=np.int32))
    B = B_loc * world
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                   syn.logit_ind2item_ind, loss_function='mw', n_sampled=S, params=params,
                                   dtype=np.float64)
    pos = syn.positives_dict()
    ref.prepare_warp(pos, pos)
    # the pool: ONE device draw over both shards (p ~ a Zipf weight per item), the same on every rank
    w_all = (1.0 / np.arange(1, n_items + 1) ** 0.8).astype(np.float32)
    mine = np.arange(rank, n_items, world)
    sampler = DeviceSampler(mine.astype(np.int32), w_all[mine], device=dev, seed=100 + rank)
    rng = np.random.default_rng(5)                                    # identical stream on both ranks
    # graphs: step 0 eager, 1 captured, 2 eager again (the one-owner pool needs larger blocks), 3 captured,
    # 4..6 replayed -- 4 with a fresh pool inside the kept capacity; target rows received vary per step
    for step in range(7):
        pool = None
        if step in (0, 2, 4):
            pool_t = draw_global_pool(sampler, S)
            both = [torch.empty_like(pool_t) for _ in range(world)]
            dist.all_gather(both, pool_t)
            assert all(torch.equal(both[0], b) for b in both)         # every rank holds the same pool
            pool = pool_t.cpu().numpy()
            assert len(np.unique(pool)) == S and pool.min() >= 0 and pool.max() < n_items
            if step == 2:                                             # ... and an all-on-one-owner pool
                pool = rng.choice(np.arange(1, n_items, world), size=S, replace=False).astype(np.int32)
            id2idx = {int(v): i for i, v in enumerate(pool)}
            model.set_pool(pool)
            cur = pool
        gu, gi = [], []
        for g in range(world):
            lu = rng.integers(0, len(np.arange(g, n_users, world)), size=B_loc)
            users = lu * world + g
            k = rng.integers(0, syn.n_pos, size=B_loc)
            gu.append(users)
            gi.append(syn.pos_items[syn.pos_ptr[users] + k])
        gi[0][0] = cur[0]                                             # a target that is also a pool slot
        l_ref = ref.step(np.concatenate(gu).tolist(), np.concatenate(gi).tolist(), pool, id2idx, loss='mw')
        model.step(gu[rank].astype(np.int32), gi[rank].astype(np.int32))
        np.testing.assert_allclose(float(model.read_loss().item()), l_ref, rtol=1e-4, err_msg='step %d' % step)
    if graphs:
        assert model._graph_key is not None and set(model._graphs) == {'fwd_gather', 'k7_sorts', 'fwd_score', 'loss', 'bwd_gemms',
                                                                        'apply'}
    got = model.gather_global_tables()
    np.testing.assert_allclose(got['user'], ref.att_emb.params['userembed_cat_0'][2:], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(got['item'], ref.att_emb.params['itemembed_cat_0'][2:], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(got['item_bias'], ref.att_emb.params['item_bias_cat_0'][2:, 0], rtol=1e-4, atol=2e-6)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def _two_rank_rep_worker(rank, world, port, out_dir):
    """ShardedHMFRepTokens on the HIP backend, two rank processes on the one GPU (gloo underneath): id table striped,
    token table replicated, the merged token gradient all-reduced; five steps vs the oracle on the global batch."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from arx.dist import ShardedHMFRepTokens
    from arx.utils.synthetic import SyntheticHMF

    n_users, n_items, d, B_loc, S, V = 601, 1003, 64, 1024, 256, 300   # past the rank-sort limits of both K7 passes
    syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=8,
                       item_mulhot=True, mulhot_vocab=V, avg_len=5, max_len=12)
    ia = syn.i_attr
    n_tok = ia._embedding_classes_list_mulhot[0]
    params = syn.glorot_params(d, seed=2, scale=0.5)
    tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
              'item_bias': params['item_bias_cat_0'][2:], 'token': params['itemembed_mulhot_0'],
              'token_bias': params['item_bias_mulhot_0']}
    bags = (np.asarray(ia.features_mulhot[0]), np.asarray(ia.mulhot_starts[0]), np.asarray(ia.mulhot_lengths[0]))
    model = ShardedHMFRepTokens(n_users, n_items, d, B_loc, S, 0.5, rank, world, dev, bags, n_tok, tables=tables)
    own_users = np.arange(rank, n_users, world)
    ptr = np.zeros(len(own_users) + 2, dtype=np.int32)
    its = []
    for k, u in enumerate(own_users):
        its.extend(syn.pos_items[syn.pos_ptr[u]:syn.pos_ptr[u + 1]].tolist())
        ptr[k + 1] = len(its)
    ptr[-1] = ptr[-2]
    model.set_positives(ptr, np.asarray(its, dtype=np.int32))
    B = B_loc * world
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                   syn.logit_ind2item_ind, loss_function='mw', n_sampled=S, params=params,
                                   dtype=np.float64)
    pos = syn.positives_dict()
    ref.prepare_warp(pos, pos)
    rng = np.random.default_rng(5)                                    # identical stream on both ranks
    for step in range(5):
        pool = None
        if step in (0, 2):
            pool = syn.sample_pool(S, rng)                            # any owners: blocks of unequal size
            if step == 2:                                             # ... and an all-on-one-owner pool
                pool = rng.choice(np.arange(1, n_items, world), size=S, replace=False).astype(np.int32)
            id2idx = {int(v): i for i, v in enumerate(pool)}
            model.set_pool(pool)
            cur = pool
        gu, gi = [], []
        for g in range(world):
            lu = rng.integers(0, len(np.arange(g, n_users, world)), size=B_loc)
            users = lu * world + g
            k = rng.integers(0, syn.n_pos, size=B_loc)
            gu.append(users)
            gi.append(syn.pos_items[syn.pos_ptr[users] + k])
        gi[0][0] = cur[0]                                             # a target that is also a pool slot
        l_ref = ref.step(np.concatenate(gu).tolist(), np.concatenate(gi).tolist(), pool, id2idx, loss='mw')
        model.step(gu[rank].astype(np.int32), gi[rank].astype(np.int32))
        np.testing.assert_allclose(float(model.read_loss().item()), l_ref, rtol=1e-4, err_msg='step %d' % step)
    got = model.gather_global_tables()
    P = ref.att_emb.params
    for name, want in (('user', P['userembed_cat_0'][2:]), ('item', P['itemembed_cat_0'][2:]),
                       ('item_bias', P['item_bias_cat_0'][2:, 0]), ('token', P['itemembed_mulhot_0']),
                       ('token_bias', P['item_bias_mulhot_0'][:, 0])):
        np.testing.assert_allclose(got[name], want, rtol=1e-4, atol=2e-6, err_msg=name)
    with open(os.path.join(out_dir, "rep%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_replicated_token_table_two_ranks_one_gpu(dev, tmp_path):
    """Round 5: the redesigned multi-hot step of the sharded model (ShardedHMFRepTokens) with its N > 1 branches on the
    HIP backend -- two rank processes on the test box's one GPU, gloo underneath (as the id-only two-rank test)."""
    import torch.multiprocessing as mp
    port = 29760 + (os.getpid() % 100)
    mp.spawn(_two_rank_rep_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / ("rep%d" % r)) for r in range(2))


def test_sharded_logits_exchange_two_ranks_one_gpu(dev, tmp_path):
    """exchange='logits' on the HIP backend, two rank processes on the one GPU (gloo underneath): latents
    all-gathered, per-owner partial logits crossing by all_to_all and their gradients back, dU reduce-scattered --
    the same seven steps (uneven blocks, an all-on-one-owner pool, S not divisible by the world size) vs the oracle."""
    import torch.multiprocessing as mp
    port = 29640 + (os.getpid() % 100)
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), False, 'logits'), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(2))


@pytest.mark.parametrize("graphs", [True, False])
def test_sharded_hip_backend_two_ranks_one_gpu(dev, tmp_path, graphs):
    """The N > 1 branches of the sharded step on the HIP backend: two rank PROCESSES share the test box's
    one GPU and exchange over gloo (RCCL refuses two ranks on one device; the all-to-all is staged
    through the host there -- arx.dist._all_to_all).  Pool = one device draw over both shards
    (draw_global_pool over DeviceSampler.sample_with_keys), blocks padded to the largest owner
    count, an all-on-one-owner pool, S not divisible by the world size; seven steps vs the oracle on
    the global batch.  graphs=True: the kernels between the collectives are five hipGraph segments."""
    import torch.multiprocessing as mp
    port = 29860 + (os.getpid() % 100)
    mp.spawn(_two_rank_worker, args=(2, port + (50 if graphs else 0), str(tmp_path), graphs), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(2))


@pytest.mark.parametrize("S,world,neg", [(1024, 8, 0), (1000, 3, 0), (128, 2, 0), (4096, 64, 0), (7, 4, 0),
                                         (8192, 8, 0), (20000, 200, 0), (3000, 8, 5)])
def test_pool_blocks_match_numpy_double(dev, S, world, neg):
    """arx_pool_blocks (the block layout of a pool striped over the owners; dist.ShardedHMF.set_pool) == the
    numpy restatement the gloo tests run on, for every rank: counts, slot -> gathered row, the rank's slots and
    local rows with their padding; incl. an owner without pool items, pools past round 4's 4096-slot / 64-owner
    limits (advisor, round 4) and negative ids (a short device draw leaves -1): counted in counts[world], no row."""
    import torch
    from arx import ops
    from numpy_backend import NumpyBackend
    rng = np.random.default_rng(S + world)
    ids = rng.choice(10 * S + 100, size=S, replace=False).astype(np.int32)
    if world > 2:
        ids = ids[ids % world != 1]                      # owner 1 owns nothing
        ids = np.concatenate([ids, (ids[:S - len(ids)] // world) * world * 7 + 0]).astype(np.int32)[:S]
    if neg:
        ids[rng.choice(len(ids), size=neg, replace=False)] = -1
    S = len(ids)
    nb = NumpyBackend()
    t_ids = torch.from_numpy(ids).to(dev)
    c_ids = torch.from_numpy(ids)
    for rank in range(min(world, 4)):
        cnt_ref = torch.zeros(world + 1, dtype=torch.int32)
        nb.pool_blocks(c_ids, world, rank, 777, 0, cnt_ref)
        assert int(cnt_ref[world]) == neg
        cap = (int(cnt_ref[:world].max()) + 3) // 4 * 4
        g_ref, ms_ref, pr_ref = (torch.zeros(S, dtype=torch.int32) for _ in range(3))
        nb.pool_blocks(c_ids, world, rank, 777, cap, cnt_ref, g_ref, ms_ref, pr_ref)
        cnt = torch.zeros(world + 1, dtype=torch.int32, device=dev)
        ops.pool_blocks(t_ids, world, rank, 777, 0, cnt)
        assert torch.equal(cnt.cpu(), cnt_ref)
        g, ms, pr = (torch.full((S,), -5, dtype=torch.int32, device=dev) for _ in range(3))
        ops.pool_blocks(t_ids, world, rank, 777, cap, cnt, g, ms, pr)
        assert torch.equal(cnt.cpu(), cnt_ref)
        assert torch.equal(g.cpu(), g_ref)
        assert torch.equal(ms.cpu(), ms_ref)
        assert torch.equal(pr.cpu(), pr_ref)
