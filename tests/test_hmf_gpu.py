"""Whole-step HMF parity: arx.hmf.hmf_model.LatentProductModel (HIP, through the
C ABI) against oracle.ref_graph.RefLatentProductModel (numpy restatement of the
reference graph in its own full-table form), identical (user, item, negatives)
batches, N consecutive steps.  fp32 rtol 1e-4 on loss / logits / updated rows."""
import numpy as np
import pytest

from oracle import ref_graph as rg

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-6


def _build(cfg, loss, d, B, S, seed, nonlinear='linear', use_graph=True, loss_func='log', exp_p=1.005,
           mw_eval_unmasked=True):
    from arx.utils.synthetic import SyntheticHMF
    from arx.hmf.hmf_model import LatentProductModel
    syn = SyntheticHMF(seed=seed, **cfg)
    params = syn.glorot_params(d, seed=seed + 1, scale=0.5)
    if nonlinear in ('relu', 'tanh'):
        rng = np.random.default_rng(seed + 2)
        params['w1'] = (rng.standard_normal((d, 48)) * 0.3).astype(np.float32)
        params['b1'] = (rng.standard_normal((48,)) * 0.1).astype(np.float32)
        params['w2'] = (rng.standard_normal((48, d)) * 0.3).astype(np.float32)
        params['b2'] = (rng.standard_normal((d,)) * 0.1).astype(np.float32)
    i2l = syn.item_ind2logit_ind_dict()
    l2i = syn.logit_ind2item_ind
    n_s = S if loss in ('mw', 'mce') else None
    model = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.5, 1.0, syn.u_attr, syn.i_attr,
                               i2l, l2i, loss_function=loss, n_sampled=n_s, params=params,
                               nonlinear=nonlinear, hidden_size=48, top_N_items=10,
                               use_graph=use_graph, loss_func=loss_func, loss_exp_p=exp_p,
                               mw_eval_unmasked=mw_eval_unmasked)
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, i2l, l2i, loss_function=loss,
                                   n_sampled=n_s, params=params, dtype=np.float64, top_N_items=10,
                                   nonlinear=nonlinear, hidden_size=48, loss_func=loss_func,
                                   loss_exp_p=exp_p, mw_eval_unmasked=mw_eval_unmasked)
    pos = syn.positives_dict()
    if loss in ('mw', 'mce', 'warp', 'rs', 'rs-sig', 'rs-sig2', 'bbpr'):
        model.prepare_warp(pos, pos)
        ref.prepare_warp(pos, pos)
    return syn, model, ref


def _bias_sum_bounds(ref, B, lr):
    """What the ORDER of an fp32 sum may change in the bias cells of the scorer after the oracle's last step: a
    bias gradient is the sum of n = B * (logit columns feeding the cell) terms w * d_logits[b, j], w >= 0; any
    fp32 order is within n * 2^-24 * sum|terms| of the exact sum (Higham (4.4)).  sum|terms| comes from the
    oracle's own backward run on |d_logits| (linear, non-negative coefficients).  Returns
    {bias name: (bound on the weight, bound on its Adagrad slot)}."""
    last = ref.last
    m = ref.att_emb
    ga = rg.Grads()
    m.get_prediction_bwd(last['c_pred'], np.abs(last['d_logits']), ga)
    n = {}
    for site in last['c_pred']['sites']:
        p = m.params[site['bias']]
        cnt = np.bincount(np.asarray(site['inds']).ravel().astype(np.int64), minlength=p.shape[0]) * B
        n[site['bias']] = n.get(site['bias'], 0) + cnt.reshape(p.shape).astype(np.float64)
    out = {}
    for name, cnt in n.items():
        p = m.params[name]
        absg = ga.total(name, p.shape, np.float64)
        g = np.abs(last['grads'].total(name, p.shape, np.float64))
        gam = cnt * 2.0 ** -24 * absg
        acc = np.asarray(m.slots[name], dtype=np.float64)
        b_acc = 2 * g * gam + gam * gam
        out[name] = (lr * (gam / np.sqrt(acc) + g * b_acc / (2 * acc ** 1.5)), b_acc)
    return out


def _compare_state(model, ref, rtol=RTOL, atol=ATOL, bounds=None):
    """bounds: {name: (weight bound, slot bound)} added to the tolerance of those tensors (_bias_sum_bounds)."""
    got = model.att_emb.get_params()
    slots = model.att_emb.get_slots()
    for name, val in got.items():
        if bounds and name in bounds:
            for g_, r_, b_, tag in ((val, ref.att_emb.params[name], bounds[name][0], ''),
                                    (slots[name], ref.att_emb.slots[name], bounds[name][1], '/Adagrad')):
                err = np.abs(np.asarray(g_, dtype=np.float64) - r_)
                lim = rtol * np.abs(r_) + atol + b_.reshape(np.shape(r_))
                assert (err <= lim).all(), (name + tag, float((err - lim).max()), float(b_.max()))
            continue
        np.testing.assert_allclose(val, ref.att_emb.params[name], rtol=rtol, atol=atol, err_msg=name)
        np.testing.assert_allclose(slots[name], ref.att_emb.slots[name], rtol=rtol, atol=atol,
                                   err_msg=name + '/Adagrad')
    for name, p in model.rt.dense.items():
        np.testing.assert_allclose(p.w.cpu().numpy(), ref.att_emb.params[name], rtol=rtol, atol=atol,
                                   err_msg=name)


CFG_ID = dict(n_users=500, n_items=700, logit_size=600)
CFG_HET = dict(n_users=500, n_items=700, logit_size=700, item_mulhot=True, mulhot_vocab=300,
               avg_len=6, max_len=18)
CFG_MIX = dict(n_users=400, n_items=600, logit_size=600, item_mulhot=True, user_mulhot=True,
               mulhot_vocab=200, avg_len=5, max_len=12, item_id_feature=False)


@pytest.mark.parametrize("cfg,loss,d,B,S", [
    (CFG_ID, 'mw', 128, 64, 256),
    (CFG_HET, 'mw', 128, 64, 256),
    (CFG_MIX, 'mw', 32, 48, 128),            # d = 32: not a shape of the fused scorer family -> K4 + K6
    (CFG_MIX, 'mw', 64, 48, 128),            # ... and the same layout on it
    (CFG_ID, 'mce', 128, 64, 256),           # build-defined sampled softmax (fused target score)
    (CFG_HET, 'mce', 64, 32, 128),           # ... on the fused 'mce' family (d = 64)
    (CFG_ID, 'mce', 64, 200, 256),
    (CFG_ID, 'ce', 32, 64, None),
    (CFG_HET, 'ce', 32, 64, None),
    (CFG_HET, 'warp', 64, 32, None),
])
@pytest.mark.parametrize("use_graph", [True, False])
def test_hmf_steps_match_oracle(dev, cfg, loss, d, B, S, use_graph):
    syn, model, ref = _build(cfg, loss, d, B, S, seed=3, use_graph=use_graph)
    rng = np.random.default_rng(11)
    for step in range(4):
        users, items = syn.sample_batch(B, rng)
        users[1] = users[0]                      # duplicate user rows in one batch
        items[2] = items[3]
        pool = id2idx = None
        if loss in ('mw', 'mce') and step % 2 == 0:       # resample cadence
            pool = syn.sample_pool(S, rng)
            pool[:4] = items[:4]                 # targets inside the pool -> masked
            pool = np.unique(pool)
            extra = np.setdiff1d(syn.item_population, pool)[:S - len(pool)]
            pool = np.concatenate([pool, extra]).astype(np.int32)
            id2idx = {int(v): i for i, v in enumerate(pool)}
            last_id2idx = id2idx
        l_ref = ref.step(list(users), list(items), pool, id2idx if id2idx else
                         (last_id2idx if loss in ('mw', 'mce') else None), loss=loss)
        l_got = model.step(None, list(users), list(items), None, pool, id2idx, loss=loss)
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
        _compare_state(model, ref)
    if loss == 'mw':
        from conftest import assert_mw_scorer_path
        assert_mw_scorer_path(model._plan('train'), B, S, d)
    if loss == 'mce':                            # evaluates with the full softmax
        from conftest import assert_scorer_path
        from arx import ops
        fused = assert_scorer_path(model._plan('train'), B, S, d, 'mce')
        assert fused or d != 64 or ops.SCORER_F32                     # d = 64: the fused family ran
        e_ref = ref.step(list(users), list(items), forward_only=True, loss=loss)
        e_got = model.step(None, list(users), list(items), forward_only=True, loss=loss)
        np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)


@pytest.mark.parametrize("loss,loss_func,exp_p", [
    ('rs', 'log', 1.005), ('rs', 'exp', 1.3), ('rs', 'poly2', 0.7), ('rs', 'square', 1.005),
    ('rs-sig', 'log', 1.005), ('rs-sig', 'linear', 1.005), ('rs-sig2', 'poly', 1.2),
    ('bbpr', 'log', 1.005),
])
def test_hmf_rs_family_matches_oracle(dev, loss, loss_func, exp_p):
    """a13: rs / rs-sig / rs-sig2 / bbpr with every loss_func transform
    (embed_attribute.py:551-603), whole step vs the oracle, multi-hot item attributes."""
    syn, model, ref = _build(CFG_HET, loss, 32, 48, None, seed=9, loss_func=loss_func, exp_p=exp_p)
    rng = np.random.default_rng(3)
    # 'square' multiplies the gradient by 2s (s ~ 1e2): after one Adagrad step at lr 0.5 every
    # touched weight has jumped by ~0.5 and the hinge activations flip chaotically between fp32
    # and fp64 -- one step is the meaningful comparison there.
    for step in range(1 if loss_func == 'square' else 3):
        users, items = syn.sample_batch(48, rng)
        users[1] = users[0]
        l_ref = ref.step(list(users), list(items), loss=loss)
        l_got = model.step(None, list(users), list(items), loss=loss)
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
        # 'square' (loss_func): per-logit gradients of magnitude ~1e2 with both signs are summed into ONE bias
        # cell per token row; the sum cancels to ~1e0, so fp32 summation ORDER shows up at 1.3e-4 of the
        # squared sum in the Adagrad slot.  Ill-conditioned input, not kernel error: the bias cells get the
        # derived summation bound on top of RTOL (_bias_sum_bounds), every other tensor holds RTOL as is
        bounds = _bias_sum_bounds(ref, 48, 0.5) if loss_func == 'square' else None
        _compare_state(model, ref, rtol=RTOL, atol=2e-5, bounds=bounds)
    e_ref = ref.step(list(users), list(items), forward_only=True, loss=loss)
    e_got = model.step(None, list(users), list(items), forward_only=True, loss=loss)
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)


def test_c1_shape_ce_three_seeds(dev):
    """C1: ML-1m shape (6040 users, 3883 items, V=3100, d=32, B=64, ce, lr=1)."""
    cfg = dict(n_users=6040, n_items=3883, logit_size=3100)
    for seed in (0, 1, 2):
        syn, model, ref = _build(cfg, 'ce', 32, 64, None, seed=seed)
        rng = np.random.default_rng(seed)
        for step in range(3):
            users, items = syn.sample_batch(64, rng)
            l_ref = ref.step(list(users), list(items), loss='ce')
            l_got = model.step(None, list(users), list(items), loss='ce')
            np.testing.assert_allclose(l_got, l_ref, rtol=RTOL)
        _compare_state(model, ref)


def test_eval_recommend_and_logits(dev):
    # (mw_eval_unmasked=False: the evaluation graph with the eval positives masked -- the numeric check of
    # that form; the reference's default, unmasked, is what every other 'mw' test evaluates with)
    syn, model, ref = _build(CFG_HET, 'mw', 64, 32, 128, seed=5, mw_eval_unmasked=False)
    rng = np.random.default_rng(2)
    users, items = syn.sample_batch(32, rng)
    pool = syn.sample_pool(128, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    l_ref = ref.step(list(users), list(items), pool, id2idx, loss='mw')
    l_got = model.step(None, list(users), list(items), None, pool, id2idx, loss='mw')
    np.testing.assert_allclose(l_got, l_ref, rtol=RTOL)
    # sampled logits of that step
    pred = model.batch_loss.inputs[0]
    if pred.value is None:        # fused 'mw' scorer: no [B, S] logits in a train step -- materialise them here
        import torch
        from arx import ops
        lat, pe = pred.inputs
        lg = torch.empty(pred.shape, dtype=torch.float32, device=lat.value.device)
        ops.gemm(lat.value, pe.value, lg, model.rt.ws, transB=True, col_bias=pe.bias_value)
    else:
        lg = pred.value
    np.testing.assert_allclose(lg.cpu().numpy(), ref.last['logits'], rtol=RTOL, atol=1e-5)
    # forward_only -> loss_eval ('warp' over the full vocabulary with the eval positives)
    e_ref = ref.step(list(users), list(items), forward_only=True, loss='warp')
    e_got = model.step(None, list(users), list(items), forward_only=True, loss='warp')
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)
    # recommend -> top-N logit indices
    r_ref = ref.step(list(users), None, recommend=True)
    r_got = model.step(None, list(users), None, recommend=True)
    np.testing.assert_array_equal(r_got, r_ref)


@pytest.mark.parametrize("nonlinear", ['relu', 'tanh'])
def test_mlp_variant(dev, nonlinear):
    syn, model, ref = _build(CFG_ID, 'mw', 32, 32, 128, seed=7, nonlinear=nonlinear)
    rng = np.random.default_rng(4)
    pool = syn.sample_pool(128, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    for step in range(3):
        users, items = syn.sample_batch(32, rng)
        l_ref = ref.step(list(users), list(items), pool if step == 0 else None, id2idx, loss='mw')
        l_got = model.step(None, list(users), list(items), None, pool if step == 0 else None, id2idx,
                           loss='mw')
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL)
    _compare_state(model, ref, rtol=1e-4, atol=1e-5)


def test_checkpoint_roundtrip(dev, tmp_path):
    syn, model, ref = _build(CFG_ID, 'mw', 32, 32, 128, seed=9)
    rng = np.random.default_rng(1)
    pool = syn.sample_pool(128, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    users, items = syn.sample_batch(32, rng)
    model.step(None, list(users), list(items), None, pool, id2idx, loss='mw')
    path = model.saver.save(None, str(tmp_path / 'best.ckpt'), global_step=0)
    before = model.att_emb.get_params()
    users2, items2 = syn.sample_batch(32, rng)
    l2 = model.step(None, list(users2), list(items2), None, None, id2idx, loss='mw')
    model.saver.restore(None, path)
    after = model.att_emb.get_params()
    for k in before:
        np.testing.assert_array_equal(before[k], after[k])
    l2b = model.step(None, list(users2), list(items2), None, None, id2idx, loss='mw')
    assert l2 == l2b     # deterministic kernels: same state + same batch -> same loss bits


@pytest.mark.parametrize("mode", ['fused', 'chunked', 'fused-overflow', 'fused-ties'])
def test_recommend_streaming_topk_equals_materialised(dev, monkeypatch, mode):
    """the streaming full-vocabulary top-k (StreamTopK) returns exactly the recommendation of the
    materialised [mb, V] path (index-exact, same tie rule).  fused (round 5): first chunk -> thresholds, then the
    scorer GEMM over the rest of the vocabulary keeps only the logits above them (arx_gemm_nt_topk_filter), select,
    merge; chunked: GEMM + select + merge per chunk; fused-overflow: item biases RISING along the vocabulary -- every
    later column beats the first chunk's thresholds, the candidate lists overflow, the flag sends the request to the
    chunked path; fused-ties: blocks of items with IDENTICAL rows and biases across the chunk boundary (equal logits:
    the lower index has to win, inside the candidate lists and against the first chunk's)."""
    from arx.hmf import hmf_model as hm
    cfg = dict(n_users=200, n_items=5000, logit_size=5000)
    syn, model_a, ref = _build(cfg, 'ce', 32, 32, None, seed=4)
    monkeypatch.setenv('ARX_STREAM_TOPK_BYTES', '0')             # force the streaming node
    syn2, model_b, _ = _build(cfg, 'ce', 32, 32, None, seed=4)
    assert isinstance(model_b.topk, hm.StreamTopK) and isinstance(model_a.topk, hm.TopK)
    model_b.topk.chunk = 1536                                     # several chunks + a ragged tail
    model_b.topk._buf = model_b.topk._buf[:, :1536].contiguous()
    assert model_b.topk.fused
    if mode == 'chunked':
        model_b.topk.fused = False
    if mode == 'fused-overflow':
        model_b.topk.slack, model_b.topk.min_capp = 0.0, 8        # segments of 8 for tiles that yield 64
    if mode in ('fused-overflow', 'fused-ties'):
        import torch
        for m in (model_a, model_b):
            t = m.att_emb.item_feats[0].table
            if mode == 'fused-overflow':
                rows = m.att_emb._pool_embed('full', 1).feats[0].maps[0].long()      # table row of logit column j
                t.bias[rows] = torch.arange(rows.numel(), device=t.bias.device, dtype=torch.float32)
            else:
                for lo in (100, 1500, 1530, 3000, 4990):          # (1530 .. 1546 straddles the first chunk's end)
                    t.E[lo + 1:lo + 17] = t.E[lo:lo + 1]
                    t.bias[lo + 1:lo + 17] = t.bias[lo:lo + 1]
                t.bias[1500:1517] += 3.0                           # ... and make two of the blocks winners
                t.bias[1530:1547] += 3.0
        P = ref.att_emb.params
        te = model_a.att_emb.item_feats[0].table
        P['itemembed_cat_0'][...] = te.E.cpu().numpy()
        P['item_bias_cat_0'][:, 0] = te.bias.cpu().numpy()
    rng = np.random.default_rng(0)
    users, items = syn.sample_batch(32, rng)
    ra = model_a.step(None, list(users), list(items), recommend=True)
    rb = model_b.step(None, list(users), list(items), recommend=True)
    rr = ref.step(list(users), list(items), recommend=True)
    np.testing.assert_array_equal(ra, rr)
    np.testing.assert_array_equal(rb, rr)
    if mode == 'fused-overflow':
        assert int(model_b.topk.overflow.item()) != 0            # (the fused run of this request did overflow)
    elif mode.startswith('fused'):
        assert int(model_b.topk.overflow.item()) == 0 and model_b.topk.fused
    rb2 = model_b.step(None, list(users), list(items), recommend=True)      # the captured plan, replayed
    np.testing.assert_array_equal(rb2, rr)


@pytest.mark.parametrize("use_graph", [True, False])
def test_hmf_user_dropout_replayed_through_oracle(dev, use_graph):
    """hmf_model.py:78 / run_hmf.py:35 (keep_prob default 0.5): dropout on the user embedding.
    The device draws the masks (new on every step, also under hipGraph replay); the oracle
    replays them."""
    from arx.utils.synthetic import SyntheticHMF
    from arx.hmf.hmf_model import LatentProductModel
    d, B, S, keep = 32, 64, 128, 0.5
    syn = SyntheticHMF(seed=5, **CFG_HET)
    params = syn.glorot_params(d, seed=6, scale=0.5)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    model = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.5, 1.0, syn.u_attr, syn.i_attr, i2l, l2i,
                               loss_function='mw', n_sampled=S, params=params, dropout=keep,
                               use_graph=use_graph)
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, i2l, l2i, loss_function='mw',
                                   n_sampled=S, params=params, dtype=np.float64)
    pos = syn.positives_dict()
    model.prepare_warp(pos, pos)
    ref.prepare_warp(pos, pos)
    rng = np.random.default_rng(8)
    pool = syn.sample_pool(S, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    prev = None
    for step in range(4):
        users, items = syn.sample_batch(B, rng)
        ps = pool if step == 0 else None
        l_got = model.step(None, list(users), list(items), None, ps, id2idx if ps is not None else None, loss='mw')
        mask = model.embedded_user.keep.cpu().numpy().reshape(B, d)
        assert abs(mask.mean() - keep) < 0.05
        if prev is not None:
            assert (mask != prev).any()
        prev = mask
        l_ref = ref.step(list(users), list(items), ps, id2idx, loss='mw', keep_prob=keep, user_mask=mask)
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
        _compare_state(model, ref)
    e_ref = ref.step(list(users), list(items), forward_only=True, loss='mw')
    e_got = model.step(None, list(users), list(items), forward_only=True, loss='mw')
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)


@pytest.mark.parametrize("nonlinear", ['relu', 'tanh'])
def test_hmf_mlp_dropout_replayed_through_oracle(dev, nonlinear):
    """hmf_model.py:80-94 with keep_prob < 1: tf.nn.dropout after each of the three activations
    of the user MLP (none on the raw lookup, :87)."""
    d, B, S, keep = 32, 64, 128, 0.7
    syn, model, ref = None, None, None
    from arx.utils.synthetic import SyntheticHMF
    from arx.hmf.hmf_model import LatentProductModel
    syn = SyntheticHMF(seed=7, **CFG_ID)
    params = syn.glorot_params(d, seed=8, scale=0.5)
    rng = np.random.default_rng(9)
    params['w1'] = (rng.standard_normal((d, 48)) * 0.3).astype(np.float32)
    params['b1'] = (rng.standard_normal((48,)) * 0.1).astype(np.float32)
    params['w2'] = (rng.standard_normal((48, d)) * 0.3).astype(np.float32)
    params['b2'] = (rng.standard_normal((d,)) * 0.1).astype(np.float32)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    model = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.5, 1.0, syn.u_attr, syn.i_attr, i2l, l2i,
                               loss_function='mw', n_sampled=S, params=params, dropout=keep,
                               nonlinear=nonlinear, hidden_size=48)
    ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, i2l, l2i, loss_function='mw',
                                   n_sampled=S, params=params, dtype=np.float64, nonlinear=nonlinear,
                                   hidden_size=48)
    pos = syn.positives_dict()
    model.prepare_warp(pos, pos)
    ref.prepare_warp(pos, pos)
    pool = syn.sample_pool(S, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    mlp = model.embedded_user
    for step in range(3):
        users, items = syn.sample_batch(B, rng)
        ps = pool if step == 0 else None
        l_got = model.step(None, list(users), list(items), None, ps, id2idx if ps is not None else None, loss='mw')
        masks = [mlp.keeps[0].cpu().numpy().reshape(B, d), mlp.keeps[1].cpu().numpy().reshape(B, 48),
                 mlp.keeps[2].cpu().numpy().reshape(B, d)]
        assert all(abs(mk.mean() - keep) < 0.06 for mk in masks)
        if step:
            assert any((a != b).any() for a, b in zip(masks, prev_masks))     # fresh draw per (replayed) step
        prev_masks = masks
        l_ref = ref.step(list(users), list(items), ps, id2idx, loss='mw', keep_prob=keep, mlp_masks=masks)
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
        _compare_state(model, ref)
    e_ref = ref.step(list(users), list(items), forward_only=True, loss='mw')
    e_got = model.step(None, list(users), list(items), forward_only=True, loss='mw')
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)


def test_hmf_mw_eval_unmasked_switch(dev):
    """mw_eval_unmasked=True: the evaluation loss of an 'mw' model is the full-vocabulary warp
    loss with an all-True mask -- what the reference reports, because its step() only runs
    set_mask['mw'] (hmf_model.py:209-210) -- checked against numpy on the model's own logits."""
    from arx.utils.synthetic import SyntheticHMF
    from arx.hmf.hmf_model import LatentProductModel
    d, B, S = 32, 64, 128
    syn = SyntheticHMF(seed=3, **CFG_ID)
    params = syn.glorot_params(d, seed=4, scale=0.5)
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    losses = {}
    for unmasked in (True, False):
        model = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.5, 1.0, syn.u_attr, syn.i_attr,
                                   i2l, l2i, loss_function='mw', n_sampled=S, params=params,
                                   mw_eval_unmasked=unmasked)
        pos = syn.positives_dict()
        model.prepare_warp(pos, pos)
        rng = np.random.default_rng(5)
        users, items = syn.sample_batch(B, rng)
        losses[unmasked] = model.step(None, list(users), list(items), forward_only=True, loss='mw')
        if unmasked:
            x = model.output.value.cpu().numpy().astype(np.float64)
            tcol = np.array([i2l[int(i)] for i in items])
            t = x[np.arange(B), tcol]
            want = np.log1p(np.maximum(x - t[:, None] + 1.0, 0.0).sum(1)).mean()
            np.testing.assert_allclose(losses[True], want, rtol=RTOL)
    assert losses[True] > losses[False]            # the masked form drops the positives' hinge terms


@pytest.mark.parametrize("cfg,d", [(CFG_ID, 128), (CFG_HET, 64)])
def test_hmf_mw_scorer_gemm_with_hinge_epilogue(dev, cfg, d):
    """The default 'mw' train path: the scorer GEMM carries the WMRB hinge in its epilogue (act bits
    instead of [B, S] logits / dlogits), the backward products read the bits -- against the oracle, with
    targets inside the pool (masked: their bits are cleared again) and a repeated user."""
    from arx import graph as G, ops
    B, S = 64, 256
    syn, model, ref = _build(cfg, 'mw', d, B, S, seed=5)
    rng = np.random.default_rng(13)
    last = None
    for step in range(4):
        users, items = syn.sample_batch(B, rng)
        users[1] = users[0]
        pool = id2idx = None
        if step % 2 == 0:
            pool = syn.sample_pool(S, rng)
            pool[:4] = items[:4]                 # targets inside the pool -> masked, bit cleared
            pool = np.unique(pool)
            pool = np.concatenate([pool, np.setdiff1d(syn.item_population, pool)[:S - len(pool)]]).astype(np.int32)
            id2idx = last = {int(v): i for i, v in enumerate(pool)}
        l_ref = ref.step(list(users), list(items), pool, id2idx or last, loss='mw')
        l_got = model.step(None, list(users), list(items), None, pool, id2idx, loss='mw')
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL, err_msg='step %d' % step)
        _compare_state(model, ref)
    from conftest import assert_mw_scorer_path
    assert assert_mw_scorer_path(model._plan('train'), B, S, d) or ops.SCORER_F32    # the path under test ran


@pytest.mark.parametrize("fused", ['1', '0'])
@pytest.mark.parametrize("cfg,loss", [(CFG_ID, 'mw'), (CFG_HET, 'mw'), (CFG_HET, 'mce')])
def test_hmf_streaming_eval_loss(dev, monkeypatch, cfg, loss, fused):
    """Evaluation loss of a sampled-loss model over the FULL vocabulary without [mb, V] logits
    (StreamEvalLoss; fused = 1, round 5: ONE pass of the scorer GEMM whose epilogue keeps the per-row sums,
    arx_gemm_nt_eval_parts; fused = 0: chunked scorer GEMM + running per-row reductions; both: positives taken out
    afterwards): equal to the oracle's forward_only loss and to the materialising path."""
    monkeypatch.setenv('ARX_STREAM_TOPK_BYTES', '1')          # force streaming at this size
    monkeypatch.setenv('ARX_STREAM_EVAL_CHUNK', '96')         # several ragged chunks of the pool
    monkeypatch.setenv('ARX_EVAL_FUSED', fused)
    from arx import graph as G
    d, B, S = 32, 48, 128
    syn, model, ref = _build(cfg, loss, d, B, S, seed=6)
    assert isinstance(model.loss_eval.inputs[0], G.StreamEvalLoss)
    assert model.loss_eval.inputs[0].fused == (fused == '1')
    rng = np.random.default_rng(2)
    pool = syn.sample_pool(S, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    for step in range(2):
        users, items = syn.sample_batch(B, rng)
        users[3] = users[2]
        ps = pool if step == 0 else None
        l_ref = ref.step(list(users), list(items), ps, id2idx, loss=loss)
        l_got = model.step(None, list(users), list(items), None, ps, id2idx if ps is not None else None, loss=loss)
        np.testing.assert_allclose(l_got, l_ref, rtol=RTOL)
    e_ref = ref.step(list(users), list(items), forward_only=True, loss=loss)
    e_got = model.step(None, list(users), list(items), forward_only=True, loss=loss)
    np.testing.assert_allclose(e_got, e_ref, rtol=RTOL)


@pytest.mark.parametrize("cfg", [CFG_ID, CFG_HET])
def test_hmf_prepare_next_ring_mode_bit_identical(dev, monkeypatch, cfg):
    """prepare_next (ring mode: the K7 sort half of step t + 1 runs as a side branch of step t): the same steps with
    and without the announcement leave bit-identical tables and losses -- through graph capture of both parities, a
    pool redraw that is announced, one that is NOT, a step without announcement in the middle and an announced
    buffer that is refilled in place before its step -- and match the oracle."""
    import torch
    monkeypatch.setenv('ARX_K7_EARLY_MIN', '0')           # (the sort branch exists at this small size too)
    B, S, d = 512, 128, 64
    steps = 12
    rng = np.random.default_rng(3)
    syn0, m_plain, ref = _build(cfg, 'mw', d, B, S, seed=9)
    _, m_ring, _ = _build(cfg, 'mw', d, B, S, seed=9)
    dev_ = m_ring.rt.device
    batches = [syn0.sample_batch(B, rng) for _ in range(steps)]
    pools = {0: syn0.sample_pool(S, rng), 6: syn0.sample_pool(S, rng), 9: syn0.sample_pool(S, rng)}
    tb = [(torch.from_numpy(u.astype(np.int32)).to(dev_), torch.from_numpy(i.astype(np.int32)).to(dev_))
          for u, i in batches]
    tp = {k: torch.from_numpy(v.astype(np.int32)).to(dev_) for k, v in pools.items()}
    id2idx = None
    for k in range(steps):
        u, i = batches[k]
        pool = pools.get(k)
        if pool is not None:
            id2idx = {int(v): j for j, v in enumerate(pool)}
        l_ref = ref.step(list(u), list(i), pool, id2idx, loss='mw')
        l_a = m_plain.step(None, tb[k][0], tb[k][1], None, tp.get(k), None, loss='mw')
        if k + 1 < steps and k != 4:                      # (step 4 announces nothing: step 5 sorts for itself)
            nxt_pool = tp.get(k + 1) if k + 1 != 9 else None      # the redraw of step 9 is NOT announced
            m_ring.prepare_next(tb[k + 1][0], tb[k + 1][1], nxt_pool)
            if k == 7:
                # (advisor, round 4) a loader refills the ANNOUNCED buffer in place before step 8 runs: same tensor
                # object, other ids -- the sort done ahead is stale and step 8 has to sort for itself
                u8 = np.roll(batches[8][0], 3)
                batches[8] = (u8, batches[8][1])
                tb[8][0].copy_(torch.from_numpy(u8.astype(np.int32)).to(dev_))
        l_b = m_ring.step(None, tb[k][0], tb[k][1], None, tp.get(k), None, loss='mw')
        assert l_a == l_b, (k, l_a, l_b)
        np.testing.assert_allclose(l_a, l_ref, rtol=RTOL, err_msg='step %d' % k)
    pa, pb = m_plain.att_emb.get_params(), m_ring.att_emb.get_params()
    for name in pa:
        assert np.array_equal(pa[name], pb[name]), name
    plan = m_ring._plan('train')
    assert len(plan._ring_graphs) == 2                    # both parities were captured and replayed
    _compare_state(m_plain, ref)


@pytest.mark.parametrize("cfg", [CFG_ID, CFG_HET])
def test_hmf_feeds_as_graph_nodes_bit_identical(dev, cfg):
    """The step's placeholder feeds as nodes of the captured graph (Runtime.feeds_in_graph, the default: sources
    swapped with hipGraphExecKernelNodeSetParams) against the eager copy in front of every graph launch: the same
    batches -- device tensors, a new one every step, a pool redraw in between -- leave bit-identical losses and
    tables, and the graph of the default model carries feed nodes."""
    import torch
    B, S, d = 512, 128, 64
    steps = 8
    rng = np.random.default_rng(5)
    syn0, m_nodes, _ = _build(cfg, 'mw', d, B, S, seed=11)
    _, m_eager, _ = _build(cfg, 'mw', d, B, S, seed=11)
    m_eager.rt.feeds_in_graph = False
    dev_ = m_nodes.rt.device
    batches = [syn0.sample_batch(B, rng) for _ in range(steps)]
    pools = {0: syn0.sample_pool(S, rng), 5: syn0.sample_pool(S, rng)}
    tb = [(torch.from_numpy(u.astype(np.int32)).to(dev_), torch.from_numpy(i.astype(np.int32)).to(dev_))
          for u, i in batches]
    tp = {k: torch.from_numpy(v.astype(np.int32)).to(dev_) for k, v in pools.items()}
    for k in range(steps):
        l_a = m_nodes.step(None, tb[k][0], tb[k][1], None, tp.get(k), None, loss='mw')
        l_b = m_eager.step(None, tb[k][0], tb[k][1], None, tp.get(k), None, loss='mw')
        assert l_a == l_b, (k, l_a, l_b)
    pa, pb = m_nodes.att_emb.get_params(), m_eager.att_emb.get_params()
    for name in pa:
        assert np.array_equal(pa[name], pb[name]), name
    g = m_nodes._plan('train').graph
    assert g is not None and g.feed_groups, "the captured step carries no feed nodes"
    assert m_eager._plan('train').graph.feed_groups is None


@pytest.mark.parametrize("cfg", [CFG_ID, CFG_HET])
def test_hmf_feed_nodes_unsynchronised_run(dev, cfg):
    """The same comparison over a LONG run without any host synchronisation between the steps (step_async: the host
    enqueues ~0.1 ms per step and runs ahead of the device, so the feed nodes of step t + k are re-pointed while
    step t is still executing -- advisor, round 4): per-step losses and final tables of the feed-node plan equal the
    eager-feed plan's bit for bit."""
    import torch
    B, S, d = 2048, 256, 64
    steps = 40
    rng = np.random.default_rng(6)
    syn0, m_nodes, _ = _build(cfg, 'mw', d, B, S, seed=12)
    _, m_eager, _ = _build(cfg, 'mw', d, B, S, seed=12)
    m_eager.rt.feeds_in_graph = False
    dev_ = m_nodes.rt.device
    batches = [syn0.sample_batch(B, rng) for _ in range(steps)]
    pool = torch.from_numpy(syn0.sample_pool(S, rng).astype(np.int32)).to(dev_)
    tb = [(torch.from_numpy(u.astype(np.int32)).to(dev_), torch.from_numpy(i.astype(np.int32)).to(dev_))
          for u, i in batches]
    out = {}
    for name, m in (('nodes', m_nodes), ('eager', m_eager)):
        losses = torch.zeros(steps, dtype=torch.float32, device=dev_)
        torch.cuda.synchronize()
        for k in range(steps):
            node = m.step_async(None, tb[k][0], tb[k][1], None, pool if k == 0 else None, None, loss='mw')
            losses[k:k + 1].copy_(node.read().reshape(1), non_blocking=True)
        torch.cuda.synchronize()
        out[name] = (losses.cpu().numpy(), m.att_emb.get_params())
    assert np.array_equal(out['nodes'][0], out['eager'][0]), (out['nodes'][0], out['eager'][0])
    for name in out['nodes'][1]:
        assert np.array_equal(out['nodes'][1][name], out['eager'][1][name]), name
    assert m_nodes._plan('train').graph.feed_groups


@pytest.mark.parametrize("use_graph", [True, False])
def test_hmf_host_fed_steps_staged_slab_bit_identical(dev, use_graph):
    """The reference's own hand-over: the ids of a step as HOST arrays (hmf_model.py:162-175 step() puts python lists
    into feed_dict).  Runtime.host_feed packs them into a pinned slab, one asynchronous copy on a side stream, then the
    device feeds' route; four slabs in turn.  40 steps enqueued without a host synchronisation (the slabs are reused
    ten times while earlier steps are still running), a new batch every step, lists and numpy arrays mixed: losses
    and tables equal, bit for bit, the direct pageable copies (ARX_STAGE_FEEDS=0's form) and the device-resident
    batches."""
    import torch
    B, S, d = 2048, 256, 64
    steps = 40
    rng = np.random.default_rng(7)
    syn0, m_stage, _ = _build(CFG_HET, 'mw', d, B, S, seed=13, use_graph=use_graph)
    _, m_direct, _ = _build(CFG_HET, 'mw', d, B, S, seed=13, use_graph=use_graph)
    _, m_dev, _ = _build(CFG_HET, 'mw', d, B, S, seed=13, use_graph=use_graph)
    assert m_stage.rt.stage_host_feeds
    m_direct.rt.stage_host_feeds = False
    dev_ = m_stage.rt.device
    batches = [syn0.sample_batch(B, rng) for _ in range(steps)]
    pools = {0: syn0.sample_pool(S, rng), 17: syn0.sample_pool(S, rng)}
    out = {}
    for name, m in (('stage', m_stage), ('direct', m_direct), ('dev', m_dev)):
        losses = torch.zeros(steps, dtype=torch.float32, device=dev_)
        torch.cuda.synchronize()
        for k in range(steps):
            u, i = batches[k]
            p = pools.get(k)
            if name == 'dev':
                u, i = torch.from_numpy(u.astype(np.int32)).to(dev_), torch.from_numpy(i.astype(np.int32)).to(dev_)
                p = None if p is None else torch.from_numpy(p.astype(np.int32)).to(dev_)
            elif k % 2:
                u, i = u.tolist(), i.tolist()            # (python lists, as run.py hands them over)
            node = m.step_async(None, u, i, None, p, None, loss='mw')
            losses[k:k + 1].copy_(node.read().reshape(1), non_blocking=True)
        torch.cuda.synchronize()
        out[name] = (losses.cpu().numpy(), m.att_emb.get_params())
    assert m_stage.rt._stage is not None and m_direct.rt._stage is None and m_dev.rt._stage is None
    for other in ('direct', 'dev'):
        assert np.array_equal(out['stage'][0], out[other][0]), (other, out['stage'][0], out[other][0])
        for pn in out['stage'][1]:
            assert np.array_equal(out['stage'][1][pn], out[other][1][pn]), (other, pn)


def test_hmf_empty_pool_slot_is_out_of_the_loss(dev):
    """A negative id in the sampled pool (what DeviceSampler.sample leaves where a short capped draw could not fill a
    position) is an EMPTY slot: it looks nothing up -- no read of cat_map[-1] / E[-1] (advisor, round 4) --, receives
    no update, and (advisor, round 5) is OUT of the sampled loss: its bias output is -1e30 (gather.hip
    kEmptySlotBias), so it is never hinge-active and does not enter the rank weighting.  Against a twin model whose
    pool holds, in that slot, a real item X with a zeroed id row and bias -1e30: same losses bit for bit, same
    tables except X's own rows -- and against the oracle over the S - 1 real slots."""
    import torch
    from arx.utils.synthetic import SyntheticHMF
    from arx.hmf.hmf_model import LatentProductModel
    d, B, S = 64, 256, 128
    syn = SyntheticHMF(seed=21, **CFG_ID)
    params = syn.glorot_params(d, seed=22, scale=0.5)
    rng = np.random.default_rng(23)
    batches = [syn.sample_batch(B, rng) for _ in range(3)]
    used = set(int(i) for _, it in batches for i in it)
    pool = syn.sample_pool(S, rng).astype(np.int32)
    X = next(int(i) for i in syn.item_population if int(i) not in used and int(i) not in set(pool.tolist()))
    xrow = int(np.asarray(syn.i_attr.features_cat[0])[X])
    params['itemembed_cat_0'][xrow] = 0
    params['item_bias_cat_0'][xrow] = -1e30
    i2l, l2i = syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind
    models = []
    for _ in range(2):
        m = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.5, 1.0, syn.u_attr, syn.i_attr, i2l, l2i,
                               loss_function='mw', n_sampled=S, params=params)
        m.prepare_warp(None, None)      # no positives: X in the twin's pool would be masked for the users who know it
        models.append(m)
    dev_ = models[0].rt.device
    pa, pb = pool.copy(), pool.copy()
    pa[5] = -1
    pb[5] = X
    for k, (u, it) in enumerate(batches):
        tu = torch.from_numpy(u.astype(np.int32)).to(dev_)
        ti = torch.from_numpy(it.astype(np.int32)).to(dev_)
        la = models[0].step(None, tu, ti, None, torch.from_numpy(pa).to(dev_) if k == 0 else None, None, loss='mw')
        assert np.isfinite(la)
        if k > 0:
            continue                                 # (later steps, eager -> captured: the empty slot stays harmless)
        lb = models[1].step(None, tu, ti, None, torch.from_numpy(pb).to(dev_), None, loss='mw')
        assert la == lb, (la, lb)
        ga, gb = models[0].att_emb.get_params(), models[1].att_emb.get_params()
        keep = np.ones(ga['itemembed_cat_0'].shape[0], dtype=bool)
        keep[xrow] = False
        assert not ga['itemembed_cat_0'][xrow].any() and ga['item_bias_cat_0'][xrow] == np.float32(-1e30)   # nothing written
        assert not gb['itemembed_cat_0'][xrow].any()                  # the twin's X has no active pair: zero gradient
        # the oracle over the S - 1 REAL slots (the reference's np.random.choice never leaves a hole: "not there")
        ref = rg.RefLatentProductModel(d, B, 0.5, syn.u_attr, syn.i_attr, i2l, l2i, loss_function='mw',
                                       n_sampled=S - 1, params=params, dtype=np.float64)
        ref.prepare_warp({}, {})
        real = np.delete(pa, 5)
        l_ref = ref.step(list(u), list(it), real, {int(v): i for i, v in enumerate(real)}, loss='mw')
        assert abs(la - l_ref) <= 1e-4 * abs(l_ref), (la, l_ref)
        assert np.array_equal(ga['userembed_cat_0'], gb['userembed_cat_0'])
        # (the item table's pass sorts one contribution more in the twin: runs of duplicate targets may be cut into
        # sub-sums at other positions -- the same sums in another association, not bit for bit)
        np.testing.assert_allclose(ga['itemembed_cat_0'][keep], gb['itemembed_cat_0'][keep], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ga['item_bias_cat_0'][keep], gb['item_bias_cat_0'][keep], rtol=1e-5, atol=1e-7)
    ga = models[0].att_emb.get_params()
    assert not ga['itemembed_cat_0'][xrow].any() and all(np.isfinite(v).all() for v in ga.values())
