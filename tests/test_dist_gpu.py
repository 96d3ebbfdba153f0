"""ShardedHMF on the real HIP backend (libarx.so) with a 1-rank RCCL group: the
same code path bench.py runs per rank at N>1, checked against the oracle."""
import os

import numpy as np
import pytest

from oracle import ref_graph as rg

pytestmark = pytest.mark.gpu


def test_sharded_hip_backend_world1(dev):
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
        n_users, n_items, d, B, S = 300, 500, 64, 32, 64
        syn = SyntheticHMF(n_users=n_users, n_items=n_items, seed=1, permute_logits=False, n_pos=8)
        params = syn.glorot_params(d, seed=2, scale=0.5)
        tables = {'user': params['userembed_cat_0'][2:], 'item': params['itemembed_cat_0'][2:],
                  'item_bias': params['item_bias_cat_0'][2:]}
        model = ShardedHMF(n_users, n_items, d, B, S, 0.5, 0, 1, dev, tables=tables)
        ptr = np.concatenate([syn.pos_ptr[:n_users + 1], [syn.pos_ptr[n_users]]]).astype(np.int32)
        model.set_positives(ptr, syn.pos_items)
        ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                       syn.logit_ind2item_ind, loss_function='mw', n_sampled=S,
                                       params=params, dtype=np.float64)
        pos = syn.positives_dict()
        ref.prepare_warp(pos, pos)
        rng = np.random.default_rng(3)
        for step in range(3):
            pool = None
            if step != 1:
                pool = syn.sample_pool(S, rng)
                id2idx = {int(v): i for i, v in enumerate(pool)}
                model.set_pool(pool)
            users, items = syn.sample_batch(B, rng)
            l_ref = ref.step(list(users), list(items), pool, id2idx, loss='mw')
            model.step(users, items)
            l_got = float(model.read_loss().item())
            np.testing.assert_allclose(l_got, l_ref, rtol=1e-4)
        got = model.gather_global_tables()
        np.testing.assert_allclose(got['user'], ref.att_emb.params['userembed_cat_0'][2:], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(got['item'], ref.att_emb.params['itemembed_cat_0'][2:], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(got['item_bias'], ref.att_emb.params['item_bias_cat_0'][2:, 0], rtol=1e-4,
                                   atol=2e-6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_users,n_items,V,d,B,S", [(300, 500, 120, 64, 32, 64), (3000, 4000, 900, 32, 2048, 256)])
def test_token_sharded_bags_hip_backend_world1(dev, n_users, n_items, V, d, B, S):
    """ShardedHMFBags (HET items: id table striped by item, token table striped by token) on the HIP
    backend with a 1-rank RCCL group vs the oracle; the second shape is past the rank-sort limits
    (radix sort + window apply in both K7 passes)."""
    import torch
    import torch.distributed as dist
    from arx.dist import ShardedHMFBags
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
        model = ShardedHMFBags(n_users, n_items, d, B, S, 0.5, 0, 1, dev, bags, n_tok, tables=tables)
        ptr = np.concatenate([syn.pos_ptr[:n_users + 1], [syn.pos_ptr[n_users]]]).astype(np.int32)
        model.set_positives(ptr, syn.pos_items)
        ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                       syn.logit_ind2item_ind, loss_function='mw', n_sampled=S,
                                       params=params, dtype=np.float64)
        pos = syn.positives_dict()
        ref.prepare_warp(pos, pos)
        rng = np.random.default_rng(3)
        for step in range(3):
            pool = None
            if step != 1:
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
