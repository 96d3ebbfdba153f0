"""word2vec-style recommenders: arx.word2vec.{skipgram,cbow}_model.Model on the HIP
path vs oracle.ref_w2v.RefW2VModel, same batches; separate input / output item tables."""
import numpy as np
import pytest

from oracle import ref_w2v

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 2e-6

CFG_ID = dict(n_users=300, n_items=500, logit_size=400)
CFG_HET = dict(n_users=300, n_items=500, logit_size=500, item_mulhot=True, user_mulhot=True,
               mulhot_vocab=150, avg_len=5, max_len=12)


def _build(kind, cfg, loss, d, B, n_in, seed, sep=True):
    from arx.utils.synthetic import SyntheticHMF
    from arx.word2vec import cbow_model, skipgram_model
    syn = SyntheticHMF(seed=seed, **cfg)
    syn.u_attr.set_model_size(d)
    syn.i_attr.set_model_size(d)
    params = syn.glorot_params(d, seed=seed + 1, item_output=sep, scale=0.5)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    mod = skipgram_model if kind == 'skipgram' else cbow_model
    model = mod.Model(syn.n_users, syn.n_items, d, B, 0.5, 1.0, syn.u_attr, syn.i_attr, i2l, l2i,
                      n_input_items=n_in, loss_function=loss, use_sep_item=sep, top_N_items=8,
                      params=params)
    ref = ref_w2v.RefW2VModel(kind, d, B, 0.5, syn.u_attr, syn.i_attr, i2l, l2i, n_input_items=n_in,
                              loss_function=loss, use_sep_item=sep, params=params, top_N_items=8)
    if loss in ('warp', 'bbpr'):
        pos = syn.positives_dict()
        model.prepare_warp(pos, pos)
        ref.prepare_warp(pos, pos)
    return syn, model, ref


@pytest.mark.parametrize("kind,cfg,loss,n_in,sep", [
    ('skipgram', CFG_ID, 'ce', 1, True),
    ('skipgram', CFG_HET, 'warp', 3, True),     # trains on the first context item, tests on all three
    ('cbow', CFG_HET, 'ce', 4, True),
    ('cbow', CFG_ID, 'bbpr', 2, False),          # shared input / output item tables
])
def test_w2v_steps_match_oracle(dev, kind, cfg, loss, n_in, sep):
    d, B = 32, 32
    syn, model, ref = _build(kind, cfg, loss, d, B, n_in, seed=21, sep=sep)
    rng = np.random.default_rng(9)
    for step in range(3):
        users, targets = syn.sample_batch(B, rng)
        ctx = np.stack([syn.sample_batch(B, rng)[1] for _ in range(n_in)], 0)      # [n_in, B]
        ctx[0, :3] = targets[:3]                   # input and output tables hit the same items
        l_ref = ref.step(list(users), ctx.tolist(), list(targets))
        l_got = model.step(None, list(users), ctx.tolist(), list(targets))
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
        got, slots = model.att_emb.get_params(), model.att_emb.get_slots()
        for name, val in got.items():
            np.testing.assert_allclose(val, ref.att_emb.params[name], rtol=RTOL, atol=ATOL, err_msg=name)
            np.testing.assert_allclose(slots[name], ref.att_emb.slots[name], rtol=RTOL, atol=ATOL,
                                       err_msg=name + '/Adagrad')
    users, targets = syn.sample_batch(B, rng)
    ctx = np.stack([syn.sample_batch(B, rng)[1] for _ in range(n_in)], 0)
    e_ref = ref.step(list(users), ctx.tolist(), list(targets), forward_only=True)
    e_got = model.step(None, list(users), ctx.tolist(), list(targets), forward_only=True)
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)
    r_ref = ref.step(list(users), ctx.tolist(), recommend=True)
    r_got = model.step(None, list(users), ctx.tolist(), recommend=True)
    np.testing.assert_array_equal(r_got, r_ref)
