"""Real data end to end: a slice of the reference's ML-1m example dataset goes
through arx.attributes.input_attribute.read_data (HET: 4 categorical user attributes, item id +
genres / title bags; MIX: one bag per entity) into LatentProductModel on the HIP path, trained
the way hmf/run_hmf.py:219-300 does (item_frequency -> sample_items pool, (user, item) batches in
log order), against the numpy oracle on the same attribute maps."""
import os

import numpy as np
import pytest

from oracle import ref_graph as rg

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ml1m_small')
RTOL, ATOL = 1e-4, 2e-6


def _tables(u_attr, i_attr, d, seed):
    rng = np.random.default_rng(seed)
    out = {}
    for pf, a, bias in (('user', u_attr, False), ('item', i_attr, True)):
        for kind, sizes, n in (('cat', a._embedding_classes_list_cat, a.num_features_cat),
                               ('mulhot', a._embedding_classes_list_mulhot, a.num_features_mulhot)):
            for i in range(n):
                out['%sembed_%s_%d' % (pf, kind, i)] = rng.uniform(-0.5, 0.5, (sizes[i], d)).astype(np.float32)
                if bias:
                    out['%s_bias_%s_%d' % (pf, kind, i)] = rng.uniform(-0.5, 0.5, (sizes[i], 1)).astype(np.float32)
    return out


@pytest.mark.parametrize("comb,loss", [('het', 'mw'), ('mix', 'mw'), ('het', 'ce')])
def test_read_data_feeds_hmf_training(dev, tmp_path, comb, loss):
    from arx.attributes.input_attribute import read_data
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.prepare_train import item_frequency, positive_items, sample_items
    V, d, B, S = 300, 32, 32, 64
    (data_tr, data_va, u_attr, i_attr, i2l, l2i, user_index, item_index) = read_data(
        DATA, str(tmp_path / comb), comb, V, 1, mylog=lambda m: None)
    data_tr = [p for p in data_tr if p[1] in i2l]            # run_hmf.py:169-170: targets must own a logit
    data_va = [p for p in data_va if p[1] in i2l]
    assert len(data_tr) > 4 * B
    params = _tables(u_attr, i_attr, d, seed=5)
    n_s = S if loss == 'mw' else None
    model = LatentProductModel(len(user_index), len(item_index), d, 1, B, 0.5, 1.0, u_attr, i_attr,
                               i2l, l2i, loss_function=loss, n_sampled=n_s, params=params)
    ref = rg.RefLatentProductModel(d, B, 0.5, u_attr, i_attr, i2l, l2i, loss_function=loss,
                                   n_sampled=n_s, params=params, dtype=np.float64)
    pos, pos_va = positive_items(data_tr, data_va)
    if loss == 'mw':
        model.prepare_warp(pos, pos_va)
        ref.prepare_warp(pos, pos_va)
    item_population, p_item = item_frequency(data_tr, 0.5)
    np.random.seed(1)
    id2idx = None
    for step in range(4):
        batch = data_tr[step * B:(step + 1) * B]
        users, items = [p[0] for p in batch], [p[1] for p in batch]
        pool = None
        if loss == 'mw' and step % 2 == 0:
            pool, id2idx = sample_items(item_population, S, p_item)
        l_ref = ref.step(users, items, pool, id2idx, loss=loss)
        l_got = model.step(None, users, items, None, pool, id2idx if pool is not None else None, loss=loss)
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
    got, slots = model.att_emb.get_params(), model.att_emb.get_slots()
    assert len(got) == len(params)
    for name, val in got.items():
        np.testing.assert_allclose(val, ref.att_emb.params[name], rtol=RTOL, atol=ATOL, err_msg=name)
        np.testing.assert_allclose(slots[name], ref.att_emb.slots[name], rtol=RTOL, atol=ATOL, err_msg=name)
    # validation loss over the held-out split, full vocabulary
    vb = data_va[:B]
    users, items = [p[0] for p in vb], [p[1] for p in vb]
    e_ref = ref.step(users, items, forward_only=True, loss=loss)
    e_got = model.step(None, users, items, forward_only=True, loss=loss)
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)
