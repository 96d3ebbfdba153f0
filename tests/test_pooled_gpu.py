"""EmbeddingAttribute.get_prediction with output_feat 2 / 3 (embed_attribute.py:194-200: segment_max /
score_max + log(1 + segment_sum(exp(score - score_max))) over the bag of a multi-hot output feature):
a small train plan (user lookup -> scorer -> cross-entropy -> Adagrad) against the oracle's
reference-form restatement, which back-propagates through the pooling and -- for output_feat 3 --
through tf.reduce_max of the whole table's score matrix."""
import numpy as np
import pytest

from oracle import ref_graph as rg

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-6


@pytest.mark.parametrize("pool,of", [('sampled', 2), ('sampled', 3), ('full', 2), ('full', 3)])
@pytest.mark.parametrize("mix", [False, True])
def test_get_prediction_pooled_scores_train_step(dev, pool, of, mix):
    from arx import graph as G
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.utils.synthetic import SyntheticHMF
    d, B, S, lr = 32, 16, 32, 0.5
    kw = dict(item_mix=True) if mix else dict(item_mulhot=True)
    syn = SyntheticHMF(n_users=80, n_items=120, logit_size=120, mulhot_vocab=40, avg_len=4, max_len=8, seed=11, **kw)
    syn.u_attr.set_model_size(d)
    syn.i_attr.set_model_size(d)
    params = syn.glorot_params(d, seed=12, scale=0.6)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, 0, False, i2l, l2i, params=params)
    emb.rt.set_learning_rate(lr)
    remb = rg.RefEmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, 0, False, i2l, l2i, params=params,
                                    dtype=np.float64)
    W = S if pool == 'sampled' else len(l2i)
    u_node, _ = emb.get_batch_user(1.0, concat=False)
    logits = emb.get_prediction(u_node, pool, output_feat=of)
    tgt = G.IdsInput(emb.rt, B, 'tgt')
    loss = G.MeanLoss(emb.rt, emb.compute_loss(logits, tgt, 'ce'))
    plan = G.Plan(emb.rt, [loss], True, [])
    rng = np.random.default_rng(3)
    for step in range(3):
        users = rng.integers(0, syn.n_users, size=B).astype(np.int32)
        users[1] = users[0]
        targets = rng.integers(0, W, size=B).astype(np.int32)
        if pool == 'sampled' and step != 1:
            ps = syn.sample_pool(S, rng).astype(np.int32)
            emb.update_sampled_pool(ps)
            remb.update_sampled(ps)
        # ---- oracle: forward, ce, backward through the pooling, Adagrad ----
        u, cu = remb.get_batch_user(list(users), concat=False)
        lg, cp = remb.get_prediction(u, pool, of)
        bl, cl = remb.compute_loss(lg, targets, 'ce')
        grads = rg.Grads()
        dl, _ = remb.compute_loss_bwd(cl, np.full(B, 1.0 / B))
        du = remb.get_prediction_bwd(cp, dl, grads)
        remb.get_batch_user_bwd(cu, du, grads)
        remb.apply_gradients(grads, lr)
        # ---- device ----
        emb.u_indices['input'].feed(users)
        tgt.feed(targets)
        plan.run()
        np.testing.assert_allclose(float(loss.read().item()), bl.mean(), rtol=RTOL, err_msg='loss step %d' % step)
        np.testing.assert_allclose(logits.value.cpu().numpy(), lg, rtol=RTOL, atol=1e-5)
        got = emb.get_params()
        for k, v in got.items():
            np.testing.assert_allclose(v, remb.params[k], rtol=RTOL, atol=ATOL, err_msg='%s step %d' % (k, step))


@pytest.mark.parametrize("of", [1, 2])
def test_get_prediction_latent_list(dev, of):
    """embed_attribute.py:169,178: `latent` may be a list with one latent per output feature;
    logits = mean over features of (that feature's scores with its own latent)."""
    from arx import graph as G
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.utils.synthetic import SyntheticHMF
    d, B, S, lr = 32, 16, 32, 0.5
    syn = SyntheticHMF(n_users=80, n_items=120, logit_size=120, mulhot_vocab=40, avg_len=4, max_len=8, seed=15,
                       item_mulhot=True)
    syn.u_attr.set_model_size(d)
    syn.i_attr.set_model_size(d)
    params = syn.glorot_params(d, seed=16, scale=0.6)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, 0, False, i2l, l2i, params=params)
    emb.rt.set_learning_rate(lr)
    remb = rg.RefEmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, 0, False, i2l, l2i, params=params,
                                    dtype=np.float64)
    u1, _ = emb.get_batch_user(1.0, concat=False)
    u2 = G.EntityEmbed(emb.rt, emb.u_indices['input'], emb.user_feats, with_bias=False, out_scale=0.5)
    logits = emb.get_prediction([u1, u2], 'sampled', output_feat=of)
    tgt = G.IdsInput(emb.rt, B, 'tgt')
    loss = G.MeanLoss(emb.rt, emb.compute_loss(logits, tgt, 'ce'))
    plan = G.Plan(emb.rt, [loss], True, [])
    rng = np.random.default_rng(5)
    for step in range(2):
        users = rng.integers(0, syn.n_users, size=B).astype(np.int32)
        targets = rng.integers(0, S, size=B).astype(np.int32)
        if step == 0:
            ps = syn.sample_pool(S, rng).astype(np.int32)
            emb.update_sampled_pool(ps)
            remb.update_sampled(ps)
        u, cu = remb.get_batch_user(list(users), concat=False)
        lg, cp = remb.get_prediction([u, 0.5 * u], 'sampled', of)
        bl, cl = remb.compute_loss(lg, targets, 'ce')
        grads = rg.Grads()
        dl, _ = remb.compute_loss_bwd(cl, np.full(B, 1.0 / B))
        dus = remb.get_prediction_bwd(cp, dl, grads)
        remb.get_batch_user_bwd(cu, dus[0] + 0.5 * dus[1], grads)
        remb.apply_gradients(grads, lr)
        emb.u_indices['input'].feed(users)
        tgt.feed(targets)
        plan.run()
        np.testing.assert_allclose(float(loss.read().item()), bl.mean(), rtol=RTOL)
        for k, v in emb.get_params().items():
            np.testing.assert_allclose(v, remb.params[k], rtol=RTOL, atol=ATOL, err_msg='%s step %d' % (k, step))
