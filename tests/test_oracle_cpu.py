"""CPU suite (no GPU): pins the oracle and the host helpers.

 1. golden vectors produced by the REAL reference helpers (tests/golden/make_golden.py)
    pin arx.utils.prepare_train / eval_metrics / Attributes;
 2. hand-computed known-answer cases pin the oracle's op restatements;
 3. an independent torch-autograd restatement (embedding-space form, written with a
    different code shape) pins the oracle's analytic backward + Adagrad;
 4. libarx.so loads and exports every symbol include/arx.h declares.
"""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import ref_graph as rg
from oracle import ref_lstm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_helpers.json")))


# ---------------------------------------------------------------- 1. goldens
def test_item_frequency_matches_reference_golden():
    from arx.utils import prepare_train as pt
    data_tr = [tuple(x) for x in GOLD["item_frequency"]["data_tr"]]
    for case in GOLD["item_frequency"]["cases"]:
        pop, p = pt.item_frequency(data_tr, case["power"])
        assert sorted(pop) == sorted(case["item_population"])
        got = dict(zip(pop, p))
        exp = dict(zip(case["item_population"], case["p_item"]))
        for k in exp:
            assert got[k] == pytest.approx(exp[k], rel=1e-12)


def test_positive_items_matches_reference_golden():
    from arx.utils import prepare_train as pt
    data_tr = [tuple(x) for x in GOLD["item_frequency"]["data_tr"]]
    data_va = [tuple(x) for x in GOLD["positive_items"]["data_va"]]
    pos, pos_va = pt.positive_items(data_tr, data_va)
    assert {str(k): sorted(v) for k, v in pos.items()} == GOLD["positive_items"]["train"]
    assert {str(k): sorted(v) for k, v in pos_va.items()} == GOLD["positive_items"]["valid"]


def test_sample_items_bit_exact_with_reference_stream():
    """np.random's legacy stream is frozen: same seed => same draws as the reference."""
    from arx.utils import prepare_train as pt
    data_tr = [tuple(x) for x in GOLD["item_frequency"]["data_tr"]]
    ref_pop = GOLD["item_frequency"]["cases"][0]["item_population"]
    ref_p = GOLD["item_frequency"]["cases"][0]["p_item"]
    for s in GOLD["sample_items"]:
        np.random.seed(s["seed"])
        if "uniform_over" in s:
            got, id2idx = pt.sample_items(list(range(s["uniform_over"])), s["n"])
        else:
            got, id2idx = pt.sample_items(ref_pop, s["n"], ref_p)
        assert [int(x) for x in got] == s["sampled"]
        assert {str(int(k)): int(v) for k, v in id2idx.items()} == s["id2idx"]


def test_attributes_container_matches_reference_golden():
    from arx.attributes.attribute import Attributes
    g = GOLD["attributes"]
    a = Attributes(2, [[2, 3, 1], [4, 2, 1]], 1, [[5, 6, 7, 1]], [2], [[0, 2, 3, 4]], [[2, 1, 1]],
                   [5, 6], [9])
    a.set_model_size(12)
    assert {"cat": a._embedding_size_list_cat, "mulhot": a._embedding_size_list_mulhot} == g["after_int"]
    a.set_model_size([3, 4])
    a.set_model_size([7], 1)
    assert {"cat": a._embedding_size_list_cat, "mulhot": a._embedding_size_list_mulhot} == g["after_lists"]
    a.set_target_prediction([[1]], [[2]], [[3]], [[4.0]])
    assert [a.full_cat_tr, a.full_values_tr, a.full_segids_tr, a.full_lengths_tr] == g["full"]
    assert (a.num_features_cat, a.num_features_mulhot) == (g["num_features_cat"], g["num_features_mulhot"])
    with pytest.raises(ValueError):
        a.set_model_size("x")


def test_eval_metrics_matches_reference_golden():
    from arx.utils import eval_metrics as em
    R = {int(k): v for k, v in GOLD["eval_metrics"]["R"].items()}
    T = {int(k): v for k, v in GOLD["eval_metrics"]["T"].items()}
    res = em.metrics(R, T)
    for k, v in GOLD["eval_metrics"]["result"].items():
        np.testing.assert_allclose(res[k], v, rtol=1e-12)


# ------------------------------------------------------- 2. known answers
def test_batch_slice_and_segids_known_answer():
    vals = np.array([10, 11, 12, 13, 14, 15, 16], dtype=np.int32)
    np.testing.assert_array_equal(rg.batch_slice2(vals, [4, 0, 2], [2, 3, 1]), [14, 15, 10, 11, 12, 12])
    np.testing.assert_array_equal(rg.batch_segids2([2, 3, 1]), [0, 0, 1, 1, 1, 2])
    assert rg.batch_slice2(vals, [], []).shape == (0,)
    np.testing.assert_array_equal(rg.unsorted_segment_sum(np.array([[1.], [2.], [4.]]), [1, 1, 0], 3),
                                  [[4.], [3.], [0.]])


def _tiny_attrs():
    from arx.attributes.attribute import Attributes
    # 2 users (+START), 3 items (+START); item: one id feature + one 2-token-vocab bag
    ua = Attributes(1, [np.array([2, 3, 1])], 0, [], None, [], [], [4], [])
    ia = Attributes(1, [np.array([2, 3, 4, 1])], 1, [np.array([2, 3, 2, 3, 3, 1])], None,
                    [np.array([0, 2, 3, 5, 6])], [np.array([2, 1, 2, 1])], [5], [4])
    ua.set_model_size(2)
    ia.set_model_size(2)
    ia.set_target_prediction([np.array([2, 3, 4])], [np.array([2, 3, 2, 3, 3])],
                             [np.array([0, 0, 1, 2, 2])], [np.array([[2.], [1.], [2.]])])
    return ua, ia


def test_scorer_and_mw_loss_known_answer():
    """2-d embeddings with exactly representable values, checked by hand."""
    ua, ia = _tiny_attrs()
    P = {
        'userembed_cat_0': np.array([[0, 0], [0, 0], [1, 0], [0, 2]], float),
        'itemembed_cat_0': np.array([[0, 0], [0, 0], [1, 1], [2, 0], [0, 4]], float),
        'item_bias_cat_0': np.array([[0], [0], [0.5], [0], [1]], float),
        'itemembed_mulhot_0': np.array([[0, 0], [0, 0], [2, 0], [0, 2]], float),
        'item_bias_mulhot_0': np.array([[0], [0], [1], [0]], float),
    }
    e = rg.RefEmbeddingAttribute(ua, ia, 2, 2, 0, False, {0: 0, 1: 1, 2: 2}, [0, 1, 2], params=P,
                                 dtype=np.float64)
    u, _ = e.get_batch_user([0, 1])
    np.testing.assert_array_equal(u, [[1, 0], [0, 2]])
    logits, _ = e.get_prediction(u, 'full')
    # item0: id row [1,1] b .5 ; bag {2,3} mean [1,1] b .5 -> feature scores for u0=[1,0]: 1.5, 1.5
    # item1: id row [2,0] b 0 ; bag {2} -> [2,0] b 1 -> u0: 2, 3 -> 2.5
    # item2: id row [0,4] b 1 ; bag {3,3} -> [0,2] b 0 -> u0: 1, 0 -> .5 ; u1=[0,2]: 9, 4 -> 6.5
    np.testing.assert_allclose(logits, [[1.5, 2.5, 0.5], [2.5, 0.5, 6.5]])
    e.update_sampled([2, 0])
    sl, _ = e.get_prediction(u, 'sampled')
    np.testing.assert_allclose(sl, [[0.5, 1.5], [6.5, 2.5]])
    t, _ = e.get_target_score(u, [1, 2])
    np.testing.assert_allclose(t, [2.5, 6.5])
    mask = np.array([[True, True], [False, True]])
    bl, c = e.compute_loss(sl, t, 'mw', mask)
    # row0: relu(.5-2.5+1)=0, relu(1.5-2.5+1)=0 -> log1 = 0 ; row1: masked, relu(2.5-6.5+1)=0 -> 0
    np.testing.assert_allclose(bl, [0, 0])
    bl2, c2 = e.compute_loss(sl, np.array([0.0, 7.0]), 'mw', mask)
    np.testing.assert_allclose(bl2, [np.log(1 + 1.5 + 2.5), 0.0])
    d, dt = e.compute_loss_bwd(c2, np.array([1.0, 1.0]))
    np.testing.assert_allclose(d, [[0.2, 0.2], [0, 0]])
    np.testing.assert_allclose(dt, [-0.4, 0])
    # ce / warp on the full logits
    ce, _ = e.compute_loss(logits, [1, 2], 'ce')
    np.testing.assert_allclose(ce, [np.log(np.exp([1.5, 2.5, .5]).sum()) - 2.5,
                                    np.log(np.exp([2.5, .5, 6.5]).sum()) - 6.5])
    w, _ = e.compute_loss(logits, [1, 2], 'warp', np.array([[True, False, True], [True, True, False]]))
    np.testing.assert_allclose(w, [np.log(1 + 0 + 0), np.log(1 + 0 + 0)])


def test_adagrad_known_answer():
    p, a = np.array([1.0, 2.0]), np.array([0.1, 0.1])
    rg.adagrad_apply(p, a, np.array([0.3, 0.0]), 0.5)
    np.testing.assert_allclose(a, [0.19, 0.1])
    np.testing.assert_allclose(p, [1.0 - 0.5 * 0.3 / np.sqrt(0.19), 2.0])


def test_mask_indices_follow_reference_feed_logic():
    ua, ia = _tiny_attrs()
    e = rg.RefEmbeddingAttribute(ua, ia, 3, 4, 0, False, {0: 2, 1: 0, 2: 1}, [1, 2, 0], params={})
    e.prepare_warp({0: [0, 2], 1: [1]}, {0: [1]})
    idx, V = e.mask_indices([1, 0, 5], 'mw', {2: 3, 1: 0})
    assert V == 4 and idx == [0, 4 + 3]              # user1: item1->slot0 ; user0: item2->slot3 ; user5 absent
    idx, V = e.mask_indices([0, 1], 'warp')
    assert V == 3 and idx == [2, 1, 3 + 0]
    idx, _ = e.mask_indices([0], 'warp', forward_only=True)
    assert idx == [0]


# ------------------------------------------------ 3. torch autograd witness
def _torch_hmf_loss(P, syn_maps, users, items, pool, mask, loss, targets, B):
    """Embedding-space HMF forward in torch (independent of ref_graph's code shape)."""
    ucat, icat, ivals, istarts, ilens = syn_maps

    def item_embed(ids):
        feats, biases = [], []
        if icat is not None:
            rows = torch.as_tensor(icat[ids]).long()
            feats.append(P['itemembed_cat_0'][rows])
            biases.append(P['item_bias_cat_0'][rows, 0])
        if ivals is not None:
            es, bs = [], []
            for i in ids:
                tok = torch.as_tensor(ivals[istarts[i]:istarts[i] + ilens[i]]).long()
                es.append(P['itemembed_mulhot_0'][tok].mean(0))
                bs.append(P['item_bias_mulhot_0'][tok, 0].mean())
            feats.append(torch.stack(es))
            biases.append(torch.stack(bs))
        return torch.stack(feats).mean(0), torch.stack(biases).mean(0)

    u = P['userembed_cat_0'][torch.as_tensor(ucat[users]).long()]
    Ip, bp = item_embed(pool)
    logits = u @ Ip.T + bp
    m = torch.as_tensor(mask)
    if loss == 'mw':
        It, bt = item_embed(items)
        t = (u * It).sum(1) + bt
        bl = torch.log(1 + (torch.relu(logits - t[:, None] + 1) * m).sum(1))
    elif loss == 'mce':      # build-defined sampled softmax: CE over [target score || kept sampled logits]
        It, bt = item_embed(items)
        t = (u * It).sum(1) + bt
        cat = torch.cat([t[:, None], logits.masked_fill(~m, float('-inf'))], 1)
        bl = torch.nn.functional.cross_entropy(cat, torch.zeros(B, dtype=torch.long), reduction='none')
    elif loss == 'warp':
        t = logits[torch.arange(B), torch.as_tensor(targets).long()]
        bl = torch.log(1 + (torch.relu(logits - t[:, None] + 1) * m).sum(1))
    else:
        bl = torch.nn.functional.cross_entropy(logits, torch.as_tensor(targets).long(), reduction='none')
    return bl.mean()


@pytest.mark.parametrize("loss,mulhot,id_feature", [('mw', False, True), ('mw', True, True),
                                                    ('mw', True, False), ('ce', True, True),
                                                    ('warp', True, True), ('mce', True, True),
                                                    ('mce', False, True)])
def test_oracle_step_equals_torch_autograd_plus_adagrad(loss, mulhot, id_feature):
    from arx.utils.synthetic import SyntheticHMF
    d, B, S = 8, 12, 16
    syn = SyntheticHMF(n_users=40, n_items=50, logit_size=50, item_mulhot=mulhot, mulhot_vocab=20,
                       avg_len=3, max_len=6, seed=3, item_id_feature=id_feature, n_pos=5)
    params = syn.glorot_params(d, seed=4, scale=0.5)
    i2l = syn.item_ind2logit_ind_dict()
    ref = rg.RefLatentProductModel(d, B, 0.7, syn.u_attr, syn.i_attr, i2l, syn.logit_ind2item_ind,
                                   loss_function=loss, n_sampled=S if loss in ('mw', 'mce') else None,
                                   params=params, dtype=np.float64)
    pos = syn.positives_dict()
    ref.prepare_warp(pos, pos)
    rng = np.random.default_rng(0)
    users, items = syn.sample_batch(B, rng)
    users[1] = users[0]
    pool = syn.sample_pool(S, rng)
    pool[0] = items[0]
    pool = np.unique(pool)
    pool = np.concatenate([pool, np.setdiff1d(syn.item_population, pool)[:S - len(pool)]])
    id2idx = {int(v): i for i, v in enumerate(pool)}
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    ia = syn.i_attr
    maps = (np.asarray(syn.u_attr.features_cat[0]),
            np.asarray(ia.features_cat[0]) if id_feature else None,
            np.asarray(ia.features_mulhot[0]) if mulhot else None,
            np.asarray(ia.mulhot_starts[0]) if mulhot else None,
            np.asarray(ia.mulhot_lengths[0]) if mulhot else None)
    if loss in ('mw', 'mce'):
        mask = ref.att_emb.mask(list(users), loss, id2idx)
        cols = pool
        targets = None
    else:
        cols = np.asarray(syn.logit_ind2item_ind)
        targets = [i2l[int(v)] for v in items]
        mask = np.ones((B, len(cols)), bool) if loss == 'ce' else ref.att_emb.mask(list(users), 'warp')
    L = _torch_hmf_loss(P, maps, users, items, cols, mask, loss, targets, B)
    L.backward()
    l_ref = ref.step(list(users), list(items), pool if loss in ('mw', 'mce') else None, id2idx, loss=loss)
    assert float(L) == pytest.approx(float(l_ref), rel=1e-10)
    for name, t in P.items():
        g = t.grad.numpy() if t.grad is not None else np.zeros(t.shape)
        acc = 0.1 + g * g
        exp = params[name].astype(np.float64) - 0.7 * g / np.sqrt(acc)
        np.testing.assert_allclose(ref.att_emb.params[name], exp, rtol=1e-9, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(ref.att_emb.slots[name], acc, rtol=1e-9, atol=1e-12, err_msg=name)


def test_oracle_lstm_equals_torch_autograd():
    rng = np.random.default_rng(1)
    L, B, din, h = 4, 3, 5, 6
    x = rng.standard_normal((L, B, din))
    W = rng.standard_normal((din + h, 4 * h)) * 0.3
    b = rng.standard_normal(4 * h) * 0.1
    dhs = rng.standard_normal((L, B, h))
    hs, cs, gates = ref_lstm.lstm_fwd(x, W, b, 1.0)
    dz, dx, dW, db = ref_lstm.lstm_bwd(x, W, hs, cs, gates, dhs)
    tx = torch.tensor(x, requires_grad=True)
    tW = torch.tensor(W, requires_grad=True)
    tb = torch.tensor(b, requires_grad=True)
    hp, cp, outs = torch.zeros(B, h, dtype=torch.float64), torch.zeros(B, h, dtype=torch.float64), []
    for t in range(L):
        z = torch.cat([tx[t], hp], 1) @ tW + tb
        i, j, f, o = z.split(h, 1)
        cp = torch.sigmoid(f + 1.0) * cp + torch.sigmoid(i) * torch.tanh(j)
        hp = torch.sigmoid(o) * torch.tanh(cp)
        outs.append(hp)
    H = torch.stack(outs)
    np.testing.assert_allclose(H.detach().numpy(), hs, rtol=1e-12)
    (H * torch.tensor(dhs)).sum().backward()
    np.testing.assert_allclose(tx.grad.numpy(), dx, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tW.grad.numpy(), dW, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tb.grad.numpy(), db, rtol=1e-9, atol=1e-12)


def test_clip_norm_aggregation_rule():
    """tf.gradients aggregation (gradients_impl._AggregatedGrads): all-dense => add_n;
    any IndexedSlices => concatenation, norm over un-merged values."""
    g = rg.Grads()
    g.add_dense('a', np.array([[3.0, 0.0]]))
    g.add_dense('a', np.array([[0.0, 4.0]]))
    assert g.sq_norm_unmerged('a') == pytest.approx(25.0)          # ||[3,4]||^2
    g.add_sparse('a', [0], np.array([[1.0, 0.0]]))
    assert g.sq_norm_unmerged('a') == pytest.approx(9 + 16 + 1)    # concatenated, un-merged
    g.add_sparse('b', [2, 2], np.array([[1.0], [1.0]]))
    assert g.sq_norm_unmerged('b') == pytest.approx(2.0)           # duplicates NOT merged
    np.testing.assert_allclose(g.total('b', (3, 1), np.float64), [[0], [0], [2.0]])


# ------------------------------------------------------------ 4. the C ABI
def test_library_exports_every_symbol_of_arx_h():
    from arx import _lib
    hdr = open(os.path.join(ROOT, "include", "arx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(arx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 45
    missing = [n for n in sorted(names) if not hasattr(_lib.lib, n)]
    assert not missing, "not exported by libarx.so: %s" % missing
    unbound = [n for n in sorted(names) if n not in _lib.PROTOTYPES]
    assert not unbound, "declared in arx.h but not bound in arx/_lib.py: %s" % unbound
    extra = [n for n in _lib.PROTOTYPES if n not in names]
    assert not extra, "bound but not declared in arx.h: %s" % extra
    assert _lib.lib.arx_version() >= 100


def test_scorer_family_shape_predicates_and_state_sizes():
    """Host logic of the fused scorer families (no compute, no GPU): which shapes they take and what state they ask
    of the caller -- arx_mw_scorer_* and arx_mce_scorer_* (d in {64, 128}; 'mce' at 128 since round 6), arx_gemm_bt_bx6
    (the LSTM's dx)."""
    from arx import _lib
    L = _lib.lib
    assert L.arx_mw_scorer_supported(16384, 1024, 128) and L.arx_mw_scorer_supported(51200, 1024, 64)
    assert not L.arx_mw_scorer_supported(64, 1000, 128) and not L.arx_mw_scorer_supported(64, 1024, 32)
    assert L.arx_mce_scorer_supported(51200, 1024, 64) and L.arx_mce_scorer_supported(1, 128, 64)
    assert L.arx_mce_scorer_supported(16384, 1024, 128)
    assert L.arx_mce_scorer_state_bytes(16384, 1024, 128) > L.arx_mce_scorer_state_bytes(16384, 1024, 64)
    for B, S, d in ((64, 1024, 32), (64, 1024, 256), (64, 1000, 64), (64, 4096, 64), (0, 1024, 64)):
        assert not L.arx_mce_scorer_supported(B, S, d)
        assert L.arx_mce_scorer_state_bytes(B, S, d) == 0
    # the state holds the O partials ([splits][Bp][64] f32) and four plane sets but NOTHING of size B x S x 4
    n = L.arx_mce_scorer_state_bytes(51200, 1024, 64)
    assert 0 < n < 51200 * 1024 * 4 // 2 and n % 256 == 0
    assert L.arx_mce_scorer_state_bytes(51201, 1024, 64) >= n
    ws = L.arx_mce_scorer_bwd_di_workspace_bytes(51200, 1024, 64, 1024)
    assert ws >= 50 * 1024 * 64 * 4 + 50 * 1024 * 4               # per-step partial products + bias partials
    assert L.arx_gemm_bt_bx6_supported(51200, 64, 256) and L.arx_gemm_bt_bx6_supported(7, 128, 64)
    assert not L.arx_gemm_bt_bx6_supported(100, 160, 256) and not L.arx_gemm_bt_bx6_supported(100, 64, 192)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from arx import graph
    with pytest.raises(RuntimeError):
        graph.Runtime()


def test_synthetic_layout_follows_reference_preprocessing():
    from arx.utils.synthetic import SyntheticHMF
    syn = SyntheticHMF(n_users=30, n_items=40, logit_size=25, item_mulhot=True, mulhot_vocab=12,
                       avg_len=3, max_len=5, seed=2)
    ia = syn.i_attr
    cat = np.asarray(ia.features_cat[0])
    assert len(cat) == 41 and cat[-1] == 1                         # START row (preprocess.py:198)
    starts, lens, vals = map(np.asarray, (ia.mulhot_starts[0], ia.mulhot_lengths[0], ia.features_mulhot[0]))
    assert len(starts) == 42 and len(lens) == 41 and len(vals) == lens.sum()
    np.testing.assert_array_equal(starts[1:], np.cumsum(lens))
    assert lens.min() >= 1 and vals[-1] == 1 and lens[-1] == 1     # trailing START bag (:223-226)
    l2i = syn.logit_ind2item_ind
    np.testing.assert_array_equal(np.asarray(ia.full_cat_tr[0]), cat[l2i])
    seg = np.asarray(ia.full_segids_tr[0])
    assert np.all(np.diff(seg) >= 0) and seg[-1] == len(l2i) - 1   # sorted segids (:302-323)
    fl = np.asarray(ia.full_lengths_tr[0]).reshape(-1)
    np.testing.assert_array_equal(fl, lens[l2i])
    exp_vals = np.concatenate([vals[starts[i]:starts[i] + lens[i]] for i in l2i])
    np.testing.assert_array_equal(np.asarray(ia.full_values_tr[0]), exp_vals)


def test_oracle_seq_use_concat_gradient_by_finite_differences():
    """use_concat input projection of the LSTM restatement (seqModel.py:130-146): the gradient
    implied by one un-clipped Adagrad step equals central differences of the loss."""
    from arx.utils.synthetic import SyntheticHMF
    size, B, L = 8, 4, 3
    syn = SyntheticHMF(n_users=30, n_items=40, logit_size=40, item_mulhot=True, user_mulhot=True,
                       mulhot_vocab=20, avg_len=3, max_len=6, seed=3)
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    base = syn.glorot_params(size, seed=4, scale=0.5)
    rng = np.random.default_rng(5)
    base['lstm_w'] = rng.standard_normal((2 * size, 4 * size)) * 0.3
    base['lstm_b'] = rng.standard_normal(4 * size) * 0.1
    base['w_input_user'] = rng.standard_normal((2 * size, size)) * 0.4
    base['w_input_item'] = rng.standard_normal((2 * size, size)) * 0.4
    i2l = syn.item_ind2logit_ind_dict()
    i2l[syn.n_items] = 0
    users = rng.integers(0, syn.n_users, size=B)
    tg = rng.integers(0, syn.n_items, size=(L, B))
    inp = np.concatenate([np.full((1, B), syn.n_items), tg[:-1]], 0)
    w = np.ones((L, B))

    def make(params):
        remb = rg.RefEmbeddingAttribute(syn.u_attr, syn.i_attr, B, None, L, False, i2l, syn.logit_ind2item_ind,
                                        params={k: v for k, v in params.items() if not k.startswith('lstm')},
                                        dtype=np.float64)
        return remb, ref_lstm.RefSeqModel(L, size, 1e12, B, 0.5, remb, loss='ce', params=params,
                                          use_concat=True)

    def loss_at(params):
        _, ref = make(params)
        return ref.step(list(users), inp.tolist(), tg.tolist(), w.tolist(), forward_only=True)

    remb, ref = make(base)
    ref.step(list(users), inp.tolist(), tg.tolist(), w.tolist())
    emb_name = [k for k in remb.params if k.startswith('user') and 'mulhot' in k][0]
    for name, pick in (('w_input_item', (3, 2)), ('w_input_item', (size + 1, 5)), ('w_input_user', (2, 1)),
                       ('w_input_user', (size + 3, 0)), (emb_name, None)):
        p0 = np.array(base[name], dtype=np.float64)
        if pick is None:                     # a multi-hot token row that the batch touched
            moved = np.argwhere(np.abs(remb.params[name] - p0) > 0)
            pick = tuple(moved[0])
        r = (remb.params[name][pick] - p0[pick]) / 0.5
        g_step = -r * np.sqrt(0.1 / (1.0 - r * r))
        eps = 1e-5
        hi, lo = dict(base), dict(base)
        hi[name] = p0.copy(); hi[name][pick] += eps
        lo[name] = p0.copy(); lo[name][pick] -= eps
        g_fd = (loss_at(hi) - loss_at(lo)) / (2 * eps)
        np.testing.assert_allclose(g_step, g_fd, rtol=2e-5, atol=1e-8, err_msg='%s%s' % (name, pick))


@pytest.mark.parametrize("kind", ['skipgram', 'cbow'])
def test_oracle_w2v_gradient_by_finite_differences(kind):
    """word2vec-style restatement (oracle/ref_w2v.py): the gradient implied by one Adagrad step
    on a context-item row and on a user row equals central differences of the training loss."""
    from oracle import ref_w2v
    from arx.utils.synthetic import SyntheticHMF
    d, B, n_in = 6, 5, 3
    syn = SyntheticHMF(n_users=20, n_items=30, logit_size=30, seed=2)
    syn.u_attr.set_model_size(d)
    syn.i_attr.set_model_size(d)
    base = {k: v.astype(np.float64) for k, v in syn.glorot_params(d, seed=3, item_output=True, scale=0.6).items()}
    rng = np.random.default_rng(4)
    users = rng.integers(0, 20, size=B)
    ctx = rng.integers(0, 30, size=(n_in, B))
    tgt = rng.integers(0, 30, size=B)

    def make(params):
        return ref_w2v.RefW2VModel(kind, d, B, 0.5, syn.u_attr, syn.i_attr, syn.item_ind2logit_ind_dict(),
                                   syn.logit_ind2item_ind, n_input_items=n_in, loss_function='ce',
                                   params=params)

    def train_loss(params):
        # the loss of a step is computed before the update; lr does not enter it
        return float(make(params).step(list(users), ctx.tolist(), list(tgt)))

    ref = make(base)
    ref.step(list(users), ctx.tolist(), list(tgt))
    for name, row in (('itemembed_cat_0', int(syn.i_attr.features_cat[0][ctx[0, 0]])),
                      ('itemembed_cat_0', int(syn.i_attr.features_cat[0][ctx[2, 1]])),
                      ('userembed_cat_0', int(syn.u_attr.features_cat[0][users[0]]))):
        pick = (row, 1)
        r = (ref.att_emb.params[name][pick] - base[name][pick]) / 0.5
        g_step = -r * np.sqrt(0.1 / (1.0 - r * r)) if r != 0 else 0.0
        hi, lo = dict(base), dict(base)
        hi[name] = base[name].copy(); hi[name][pick] += 1e-5
        lo[name] = base[name].copy(); lo[name][pick] -= 1e-5
        g_fd = (train_loss(hi) - train_loss(lo)) / 2e-5
        np.testing.assert_allclose(g_step, g_fd, rtol=3e-5, atol=1e-9, err_msg='%s %s %s' % (kind, name, pick))


# ---------------------------------------- 5. the embedding-space (sparse) oracle == the dense one
def _assert_sparse_state(emb_model, ref_params, ref_slots, params0, rtol=1e-10):
    """Rows the sparse oracle updated equal the dense oracle's; every other row is untouched."""
    for name, st in emb_model.t.items():
        P = np.asarray(ref_params[name], dtype=np.float64).reshape(st.base.shape)
        A = np.asarray(ref_slots[name], dtype=np.float64).reshape(st.base.shape)
        np.testing.assert_allclose(st.val, P[st.idx], rtol=rtol, atol=1e-13, err_msg=name)
        np.testing.assert_allclose(st.acc, A[st.idx], rtol=rtol, atol=1e-13, err_msg=name + '/Adagrad')
        rest = np.ones(P.shape[0], dtype=bool)
        rest[st.idx] = False
        base = np.asarray(params0[name], dtype=np.float64).reshape(st.base.shape)
        assert np.array_equal(P[rest], base[rest]), name        # the dense update left them alone
        assert np.all(A[rest] == 0.1), name


@pytest.mark.parametrize("loss", ['mw', 'mce'])
@pytest.mark.parametrize("cfg", [dict(), dict(item_mulhot=True), dict(item_mix=True, user_mulhot=True)])
def test_embedding_space_oracle_equals_dense_oracle_hmf(loss, cfg):
    """oracle/ref_embed.py (what the full-size GPU parity tests use) against ref_graph's
    reference-form step: id-only, HET and MIX layouts, three consecutive steps, pool resampled."""
    from arx.utils.synthetic import SyntheticHMF
    from oracle import ref_embed
    d, B, S = 8, 12, 16
    syn = SyntheticHMF(n_users=40, n_items=50, logit_size=50, mulhot_vocab=20, avg_len=3, max_len=6,
                       seed=5, n_pos=5, **cfg)
    params = syn.glorot_params(d, seed=6, scale=0.5)
    i2l = syn.item_ind2logit_ind_dict()
    ref = rg.RefLatentProductModel(d, B, 0.7, syn.u_attr, syn.i_attr, i2l, syn.logit_ind2item_ind,
                                   loss_function=loss, n_sampled=S, params=params, dtype=np.float64)
    pos = syn.positives_dict()
    ref.prepare_warp(pos, pos)
    emb = ref_embed.EmbedSpaceHMF(syn.u_attr, syn.i_attr, params, 0.7, loss=loss)
    ptr, items_csr = syn.positives_csr()
    rng = np.random.default_rng(1)
    id2idx = None
    for step in range(3):
        users, items = syn.sample_batch(B, rng)
        users[1] = users[0]
        items[2] = items[3]
        pool = None
        if step != 1:
            pool = syn.sample_pool(S, rng)
            pool[0] = items[0]
            pool = np.unique(pool)
            pool = np.concatenate([pool, np.setdiff1d(syn.item_population, pool)[:S - len(pool)]])
            id2idx = {int(v): i for i, v in enumerate(pool)}
        l_ref = ref.step(list(users), list(items), pool, id2idx, loss=loss)
        l_emb = emb.step(users, items, pool, ptr, items_csr)
        assert l_emb == pytest.approx(float(l_ref), rel=1e-12)
        _assert_sparse_state(emb, ref.att_emb.params, ref.att_emb.slots, params)


@pytest.mark.parametrize("loss,cfg,no_uid", [('mw', dict(), True), ('mce', dict(), False),
                                             ('mw', dict(item_mulhot=True), True)])
def test_embedding_space_oracle_equals_dense_oracle_lstm(loss, cfg, no_uid):
    """ref_embed.EmbedSpaceSeq against ref_lstm.RefSeqModel: loss, clip norm (TF-1.0 aggregation
    rule, active clipping), every table row, LSTM weights -- three steps."""
    from arx.utils.synthetic import SyntheticHMF
    from oracle import ref_embed
    size, B, L, S = 6, 4, 5, 8
    syn = SyntheticHMF(n_users=30, n_items=40, logit_size=40, mulhot_vocab=15, avg_len=3, max_len=5,
                       seed=7, n_pos=4, **cfg)
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    params = syn.glorot_params(size, seed=8, scale=0.5)
    rng = np.random.default_rng(2)
    lw = (rng.standard_normal((2 * size, 4 * size)) * 0.3)
    lb = (rng.standard_normal((4 * size,)) * 0.1)
    full = dict(params, lstm_w=lw, lstm_b=lb)
    i2l = syn.item_ind2logit_ind_dict()
    i2l[syn.n_items] = 0
    remb = rg.RefEmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, L, False, i2l, syn.logit_ind2item_ind,
                                    params=dict(params), dtype=np.float64)
    ref = ref_lstm.RefSeqModel(L, size, 0.05, B, 0.5, remb, loss=loss, no_user_id=no_uid, params=full)
    pos = syn.positives_dict()
    remb.prepare_warp(pos, pos)
    emb = ref_embed.EmbedSpaceSeq(syn.u_attr, syn.i_attr, params, lw, lb, 0.5, 0.05, loss=loss,
                                  no_user_id=no_uid)
    ptr, items_csr = syn.positives_csr()
    pool = syn.sample_pool(S, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    for step in range(3):
        users = rng.integers(0, syn.n_users, size=B)
        tg = np.stack([syn.sample_batch(B, rng)[1] for _ in range(L)], 0)
        inp = np.concatenate([np.full((1, B), syn.n_items), tg[:-1]], 0)
        lens = rng.integers(1, L + 1, size=B)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float64)
        ps = pool if step == 0 else None
        l_ref = ref.step(list(users), inp.tolist(), tg.tolist(), w.tolist(), ps, id2idx)
        l_emb = emb.step(users, inp, tg, w, ps, ptr, items_csr)
        assert l_emb == pytest.approx(float(l_ref), rel=1e-11)
        assert emb.last['gnorm'] == pytest.approx(ref.last['gnorm'], rel=1e-11)
        assert ref.last['gnorm'] > 0.05                              # clipping is active
        tabs = {k: v for k, v in remb.params.items() if not k.startswith('lstm')}
        _assert_sparse_state(emb, tabs, remb.slots, params, rtol=1e-9)
        np.testing.assert_allclose(emb.W, ref.W, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(emb.b, ref.b, rtol=1e-9, atol=1e-13)


def test_bench_line_is_compact_and_parses():
    """The driver keeps an 8 KB tail of bench.py's stdout (round 5: a 20 KB line left BENCH_r05 unparsed).  The final
    line built from a detail dict with kilobytes of prose and sub-results must stay under the limit, parse, carry the
    contract's fields and name the DOMINANT pass (by time) as `roofline`."""
    import json
    import bench
    prose = "x" * 3000
    k7 = {"kernel": "K7 " + prose, "bound": "hbm", "achieved": 800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1,
          "bytes_per_launch": 147028324, "ms_per_launch": 0.17, "traffic": 1.8e8, "unique_rows": 5, "contributions": 9}
    mf = {"kernel": "gemm_logits_hinge", "bound": "mfma", "achieved": 190.0, "peak": 416.7, "unit": "TFLOP/s",
          "frac": 0.46, "ms_per_launch": 0.022, "flops_per_launch": 4.29e9, "traffic": None, "peak_note": prose}
    ga = {"kernel": "K1 " + prose, "bound": "hbm", "achieved": 5000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.63,
          "ms_per_launch": 0.14, "bytes_per_launch": 7e8, "traffic": 7.2e8, "in_step": {"note": prose}}
    out = {"metric": bench.METRIC, "value": 7.3e7, "unit": "interactions/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 0.22, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "dtype_detail": prose, "data": "synthetic",
           "config": {"workload": "C3 " + prose, "batch": 16384, "n_sampled": 1024, "dim": 128, "setup_s": 3.0},
           "roofline": mf, "roofline_hbm": k7, "roofline_gather": ga,
           "kernels": {("k%d" % i): {"ms": 0.01, "note": prose} for i in range(12)},
           "sub": {("s%d" % i): {"ms_per_step": 0.3, "config": {"workload": prose}} for i in range(10)},
           "cpu_baseline": {"value": 77.0, "unit": "interactions/s", "cores": 128, "kind": "port", "sample": prose}}
    assert len(json.dumps(out)) > 40000
    txt = bench.compact_line(out)
    assert len(txt) < bench.LINE_LIMIT and "\n" not in txt
    j = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["config"]["workload"].startswith("C3") and j["config"]["batch"] == 16384
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["ms_per_launch"] == 0.17 and r["frac"] == 0.1 and r["traffic"] == 1.8e8
    assert r["mfma"]["frac"] == 0.46 and r["gather"]["achieved"] <= r["gather"]["peak"]
    assert j["cpu_baseline"]["cores"] == 128 and j["cpu_baseline"]["kind"] == "port"
    # a sharded (N > 1) line: one `roofline` only
    out2 = {k: v for k, v in out.items() if k not in ("roofline_hbm", "roofline_gather")}
    out2["roofline_comm_predicted"] = {"this_run": {"a": prose}, "at_8_ranks": {"b": prose * 3}}
    j2 = json.loads(bench.compact_line(out2))
    assert j2["roofline"]["bound"] == "mfma" and len(json.dumps(j2)) < bench.LINE_LIMIT


# ------------------------------------------------------------ 5. third-party witnesses (round-5 verdict, missing #6)
def test_oracle_lstm_equals_torch_nn_lstmcell():
    """oracle.ref_lstm (TF-1.0 LSTMCell: gate order i, j, f, o, forget_bias 1.0 added inside the sigmoid;
    lstm/seqModel.py:99-103) against a THIRD-PARTY cell, torch.nn.LSTMCell (gate order i, f, g, o, no forget bias):
    TF's columns permuted into torch's row blocks and forget_bias folded into the f bias must give the same
    hidden states, cell states and BPTT gradients in fp64."""
    rng = np.random.default_rng(7)
    L, B, din, h = 6, 5, 7, 9
    x = rng.standard_normal((L, B, din))
    W = rng.standard_normal((din + h, 4 * h)) * 0.3
    b = rng.standard_normal(4 * h) * 0.1
    dhs = rng.standard_normal((L, B, h))
    fb = 1.0
    hs, cs, gates = ref_lstm.lstm_fwd(x, W, b, fb)
    dz, dx, dW, db = ref_lstm.lstm_bwd(x, W, hs, cs, gates, dhs)
    cell = torch.nn.LSTMCell(din, h, bias=True).double()
    blk = lambda M, g: M[..., g * h:(g + 1) * h]
    tf2torch = (0, 2, 1, 3)            # torch block order (i, f, g, o) <- TF column blocks (i, j, f, o)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.tensor(np.concatenate([blk(W[:din], g).T for g in tf2torch], 0)))
        cell.weight_hh.copy_(torch.tensor(np.concatenate([blk(W[din:], g).T for g in tf2torch], 0)))
        bt = np.concatenate([blk(b, g) for g in tf2torch]).copy()
        bt[h:2 * h] += fb
        cell.bias_ih.copy_(torch.tensor(bt))
        cell.bias_hh.zero_()
    tx = torch.tensor(x, requires_grad=True)
    hp = torch.zeros(B, h, dtype=torch.float64)
    cp = torch.zeros(B, h, dtype=torch.float64)
    Hs, Cs = [], []
    for t in range(L):
        hp, cp = cell(tx[t], (hp, cp))
        Hs.append(hp)
        Cs.append(cp)
    H = torch.stack(Hs)
    np.testing.assert_allclose(H.detach().numpy(), hs, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(torch.stack(Cs).detach().numpy(), cs, rtol=1e-12, atol=1e-14)
    (H * torch.tensor(dhs)).sum().backward()
    np.testing.assert_allclose(tx.grad.numpy(), dx, rtol=1e-10, atol=1e-13)
    gWi, gWh = cell.weight_ih.grad.numpy(), cell.weight_hh.grad.numpy()
    gb = cell.bias_ih.grad.numpy()
    for k, g in enumerate(tf2torch):   # torch block k holds TF block g
        np.testing.assert_allclose(gWi[k * h:(k + 1) * h].T, blk(dW[:din], g), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(gWh[k * h:(k + 1) * h].T, blk(dW[din:], g), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(gb[k * h:(k + 1) * h], blk(db, g), rtol=1e-10, atol=1e-13)


def test_oracle_adagrad_equals_torch_optim_adagrad_dense_and_sparse_duplicates():
    """oracle.ref_graph.adagrad_apply / RefEmbeddingAttribute.apply_gradients (tf.train.AdagradOptimizer,
    hmf/hmf_model.py:147: accumulators start at 0.1, no epsilon, duplicate IndexedSlices rows summed BEFORE the
    single application) against torch.optim.Adagrad(initial_accumulator_value=0.1, eps=0): dense gradients over
    three steps, and sparse gradients with duplicate indices (torch coalesces an uncoalesced sparse gradient,
    i.e. sums duplicates first -- the same rule)."""
    rng = np.random.default_rng(11)
    V, d, lr = 13, 4, 0.7
    p0 = rng.standard_normal((V, d))
    # dense
    p, acc = p0.copy(), np.full((V, d), 0.1)
    tp = torch.nn.Parameter(torch.tensor(p0.copy()))
    opt = torch.optim.Adagrad([tp], lr=lr, initial_accumulator_value=0.1, eps=0.0)
    for _ in range(3):
        g = rng.standard_normal((V, d))
        rg.adagrad_apply(p, acc, g, lr)
        tp.grad = torch.tensor(g)
        opt.step()
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(acc, opt.state[tp]['sum'].numpy(), rtol=1e-13)
    # sparse with duplicates, through the oracle's own optimiser entry (two lookup sites of one table)
    class _Holder(rg.RefEmbeddingAttribute):
        def __init__(self, params):          # only what apply_gradients touches
            self.params = params
            self.slots = {k: np.full(v.shape, 0.1) for k, v in params.items()}
            self.dt = np.dtype(np.float64)
    hold = _Holder({'E': p0.copy()})
    tp = torch.nn.Parameter(torch.tensor(p0.copy()))
    opt = torch.optim.Adagrad([tp], lr=lr, initial_accumulator_value=0.1, eps=0.0)
    for _ in range(3):
        i1 = np.array([3, 3, 5, 0, 3]); v1 = rng.standard_normal((5, d))
        i2 = np.array([5, 12, 0]); v2 = rng.standard_normal((3, d))
        grads = rg.Grads()
        grads.add_sparse('E', i1, v1)
        grads.add_sparse('E', i2, v2)
        hold.apply_gradients(grads, lr)
        idx = torch.tensor(np.concatenate([i1, i2]))[None]
        tp.grad = torch.sparse_coo_tensor(idx, torch.tensor(np.concatenate([v1, v2], 0)), (V, d))
        assert not tp.grad.is_coalesced()
        opt.step()
        np.testing.assert_allclose(hold.params['E'], tp.detach().numpy(), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(hold.slots['E'], opt.state[tp]['sum'].numpy(), rtol=1e-13)
    # untouched rows: exactly unchanged (a dense Adagrad over the summed gradient would also leave them)
    untouched = np.setdiff1d(np.arange(V), [0, 3, 5, 12])
    assert np.array_equal(hold.params['E'][untouched], p0[untouched])


def test_tf1_witness_hook():
    """"auto-upgrade to real TF if importable".  oracle.tf1_witness.available() probes for
    tensorflow.compat.v1; absent here (and on the GPU box), the hook reports so and nothing else changes.  When
    present, the real TF-1 graph ops (LSTMCell / AdagradOptimizer / clip_by_global_norm) run as a further witness."""
    from oracle import tf1_witness
    if not tf1_witness.available():
        assert tf1_witness.kind() == "port"
        pytest.skip("tensorflow not importable: parity stays 'unpinned' (third-party torch witnesses above)")
    assert tf1_witness.kind() == "tf1"
    tf1_witness.check_lstm(ref_lstm)
    tf1_witness.check_adagrad(rg)


# ------------------------------------------------------------ 6. the C ABI without a GPU: behaviour, not only names
def test_abi_argument_validation_without_gpu():
    """SURVEY 8(b) asked for a host build (libarx_host.so) so that the ABI can be exercised on a GPU-less box.  It is
    not needed: libarx.so itself loads without a device (hipcc's host half links against the HIP runtime only), and
    every entry point validates its arguments BEFORE its first HIP call -- so the error half of the contract (arx.h:
    "returns 0 or a negative ARX_E* code; the message through arx_last_error") is testable here: null pointers, bad
    sizes, unsupported shapes and short workspaces come back as codes + messages, no launch is attempted, nothing
    crashes.  (The compute half needs the GPU: tests/*_gpu.py.)"""
    from arx import _lib
    lib = _lib.lib
    EINVAL, EWS, EUNS = -1, -3, -4

    def err():
        m = lib.arx_last_error()
        return m.decode() if m else ""
    assert lib.arx_version() > 0
    # null pointers / negative sizes
    assert lib.arx_gather_onehot_fwd(None, None, None, None, 8, 64, 1.0, 0, None, 64, None, None) == EINVAL
    assert "arx_gather_onehot" in err()
    assert lib.arx_loss_mce_fwdbwd(None, 0, None, None, 0, 0, 1.0, None, 4, 8, None, None, 0, None, None) == EINVAL
    assert "arx_loss_mce_fwdbwd" in err()
    assert lib.arx_shard_route(None, 4, 2, 0, 0, None, None, None) == EINVAL
    assert lib.arx_shard_route(1, 4, 2, 5, 0, None, None, None) == EINVAL          # rank >= world
    assert "arx_shard_route" in err()
    # an empty problem is NOT an error (ragged / empty inputs are legal)
    assert lib.arx_shard_route(1, 0, 2, 0, 0, None, None, None) == 0
    # size queries and shape predicates are pure host functions
    assert lib.arx_sparse_adagrad_workspace_bytes(0) > 0
    assert lib.arx_sparse_adagrad_workspace_bytes(1 << 20) > lib.arx_sparse_adagrad_workspace_bytes(1 << 10)
    assert lib.arx_mw_scorer_supported(16384, 1024, 128) == 1 and lib.arx_mw_scorer_supported(16384, 1000, 128) == 0
    assert lib.arx_mce_scorer_supported(51200, 1024, 64) == 1 and lib.arx_mce_scorer_supported(51200, 1024, 96) == 0
    assert lib.arx_mw_scorer_state_bytes(16384, 1024, 128) > 16384 * 1024 // 8       # at least the activity bits
    # a workspace that is too small is refused before any launch
    need = lib.arx_sparse_adagrad_workspace_bytes(4096)
    # (pointers are only compared with NULL before the workspace check: small integers stand in for them)
    rc = lib.arx_sparse_adagrad(1, 1, None, None, 64, 1, 1, 1, 4096, 1, 64, None, 1, None, 12, 1, need - 1, None)
    assert rc == EWS, (rc, err())
    assert "workspace" in err()
    assert lib.arx_sparse_adagrad(1, 1, None, None, 62, 1, 1, 1, 4096, 1, 64, None, 1, None, 12, 1, need, None) == EUNS
    assert "d=62" in err()
    assert lib.arx_sparse_adagrad(1, 1, None, None, 64, 1, 1, 1, 0, 1, 64, None, 1, None, 12, None, 0, None) == 0   # n = 0: nothing to do
