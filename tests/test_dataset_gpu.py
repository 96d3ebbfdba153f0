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


def test_train_recommend_evaluate_loop(dev, tmp_path):
    """hmf/run_hmf.py in miniature on the ML-1m slice: read_data -> train (ce, permuted batches)
    -> recommend for the evaluation users (run_hmf.py:340-409) -> utils/evaluate.py scores.
    Training on real interactions must lift the ranking metrics well above the untrained model."""
    import shutil
    from arx.attributes.input_attribute import read_data
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.evaluate import Evaluation
    raw = str(tmp_path / 'raw')
    shutil.copytree(DATA, raw)
    V, d, B, top_n = 400, 32, 32, 30
    (data_tr, data_va, u_attr, i_attr, i2l, l2i, user_index, item_index) = read_data(
        raw, str(tmp_path / 'cache'), 'het', V, 1, mylog=lambda m: None)
    data_tr = [p for p in data_tr if p[1] in i2l]
    model = LatentProductModel(len(user_index), len(item_index), d, 1, B, 1.0, 1.0, u_attr, i_attr, i2l, l2i,
                               loss_function='ce', top_N_items=top_n, seed=3)
    ev = Evaluation(raw, test=False)
    uids = ev.get_uids()
    ind2id = {v: k for k, v in item_index.items()}
    uids_of = {v: k for k, v in user_index.items()}

    def recommend():
        uinds = [user_index[u] for u in uids]
        R = {}
        for s in range(0, len(uinds), B):
            chunk = uinds[s:s + B]
            users = chunk + [0] * (B - len(chunk))
            recs = model.step(None, users, None, None, forward_only=True, recommend=True)
            for k in range(len(chunk)):
                R[uids[s + k]] = [ind2id[l2i[int(v)]] for v in recs[k]]
        return R

    from arx.utils.eval_metrics import metrics
    T_train = {}
    for u, i, _ in data_tr:
        T_train.setdefault(uids_of[u], []).append(str(ind2id[i]))

    def scores(R):
        ev.eval_on(R)                                      # validation truth, as the runner reports it
        s_self, s_ex = ev.get_scores()
        assert len(s_self) == len(s_ex) == 20 and np.all(np.isfinite(s_self + s_ex))
        fit = metrics({u: [str(v) for v in r] for u, r in R.items()},
                      {u: t for u, t in T_train.items() if u in R})
        return fit['prec'][1], s_self[1]                   # prec@5 on the training items / validation

    fit0, val0 = scores(recommend())
    np.random.seed(0)
    losses = []
    for step in range(400):
        users, items, _ = model.get_permuted_batch(data_tr)
        losses.append(model.step(None, users, items, loss='ce'))
    fit1, val1 = scores(recommend())
    assert np.mean(losses[-20:]) < 0.8 * np.mean(losses[:20])
    # 60 users x ~10 training interactions: the model must at least rank what it was trained on
    assert fit1 > 0.3 and fit1 > 10 * max(fit0, 0.005), (fit0, fit1, val0, val1)
