"""BASELINE.json's full sizes (configs[1] / configs[2]: 1 M users, 1 M items, d = 128, S = 1024
negatives, B = 16384; multi-hot item attributes with ~20 tokens; configs[3]: LSTM L = 50, B = 1024).

PARITY at these sizes is checked against oracle/ref_embed.py -- the embedding-space fp64
restatement of the step, proven equal to the reference-form oracle (ref_graph / ref_lstm) at small
sizes in tests/test_oracle_cpu.py: loss per step rtol 1e-4, every touched table row and Adagrad
slot rtol 1e-4, over one and three consecutive steps (test_fullsize_*_matches_embedding_space_oracle).

The dense reference-form oracle itself is O(table) per step and cannot run here, so the remaining
tests check size-independent properties:
  * bit-reproducibility: same seed, same batches -> identical tables (graph replay vs eager too);
  * locality: rows outside (batch u pool) are untouched, their Adagrad slots still at 0.1, touched
    slots strictly grew;
  * conservation of the WMRB gradient: d loss / d target score = - sum of the row's logit gradient;
  * the loss reported = mean of the per-row losses, and it falls on a repeated batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, D, B, S = 1000000, 128, 16384, 1024


def _model(mulhot, use_graph=True, seed=0, **kw):
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.synthetic import SyntheticHMF
    syn = SyntheticHMF(n_users=N, n_items=N, item_mulhot=mulhot, permute_logits=False, seed=seed)
    model = LatentProductModel(N, N, D, 1, B, 0.1, 1.0, syn.u_attr, syn.i_attr, syn.item2logit[:N],
                               syn.logit_ind2item_ind, loss_function='mw', n_sampled=S,
                               use_graph=use_graph, seed=seed)
    model.prepare_warp(syn.positives_csr(), syn.positives_csr())
    return syn, model


def _batches(syn, dev, n, seed=1):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        u, i = syn.sample_batch(B, rng)
        out.append((torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev)))
    pool = torch.from_numpy(syn.sample_pool(S, rng).astype(np.int32)).to(dev)
    return out, pool


def _tables(model):
    return {t.name: (t.E.clone(), t.acc.clone(), None if t.bias is None else t.bias.clone())
            for t in model.att_emb.tables.values()}


@pytest.mark.parametrize("mulhot", [False, True])
def test_fullsize_bit_reproducible_and_graph_equals_eager(dev, mulhot):
    res = []
    for use_graph in (True, True, False):
        syn, model = _model(mulhot, use_graph=use_graph)
        batches, pool = _batches(syn, model.rt.device, 4)
        losses = []
        for k, (u, i) in enumerate(batches):
            losses.append(model.step(None, u, i, None, pool if k == 0 else None, None, loss='mw'))
        res.append((losses, _tables(model)))
        del model
    for other in res[1:]:
        assert other[0] == res[0][0]                                   # same floats, bit for bit
        for name, (E, acc, bias) in res[0][1].items():
            assert torch.equal(E, other[1][name][0]) and torch.equal(acc, other[1][name][1]), name
            if bias is not None:
                assert torch.equal(bias, other[1][name][2]), name


@pytest.mark.parametrize("mulhot", [False, True])
def test_fullsize_locality_conservation_and_update(dev, mulhot):
    from arx import graph as G
    syn, model = _model(mulhot)
    d_ = model.rt.device
    batches, pool = _batches(syn, d_, 3)
    u, i = batches[0]
    model.step(None, u, i, None, pool, None, loss='mw')                # warm (eager) step
    before = _tables(model)
    u, i = batches[1]
    loss = model.step(None, u, i, None, None, None, loss='mw')          # captured step
    plan = model._plan('train')
    bl = [n for n in plan.order if isinstance(n, G.BatchLoss)][0]
    pred = [n for n in plan.order if type(n) is G.Prediction][0]
    ts = [n for n in plan.order if isinstance(n, G.TargetScore)][0]
    # loss = mean of the row losses; WMRB gradient conservation (dt = - sum_s dlogits)
    np.testing.assert_allclose(loss, float(bl.value.double().mean().item()), rtol=1e-6)
    if pred.fused_into_loss:      # hinge in the scorer GEMM's epilogue: dlogits = g_r * act bits
        nB = pred.shape[0]
        w = np.ascontiguousarray(pred.scorer.act_bits[:, :nB].cpu().numpy().T).view(np.uint32)   # word-major -> [B][S/32]
        act = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(w.shape[0], -1)
        dl = torch.from_numpy(act.astype(np.float64)).to(d_) * pred.scorer.g[:nB].double()[:, None]
    else:
        dl = pred.grad.double()
    dt = ts.grad.double()
    np.testing.assert_allclose(dl.sum(1).cpu().numpy(), -dt.cpu().numpy(), rtol=1e-5, atol=1e-12)
    assert float(dl.min().item()) >= 0.0 and float(dt.max().item()) <= 0.0
    # which table rows may move
    m = model.att_emb
    touched = {}
    for table, sites, bufs, total in plan.tables:
        rows = []
        for s_ in sites:
            ids = s_.ids_node.value.long()
            if s_.kind == 'cat':
                rows.append(s_.maps[0][ids].long() if s_.maps[0] is not None else ids)
            else:
                vals, starts, lens = s_.maps
                st, ln = starts[ids].long(), lens[ids].long()
                idx = torch.repeat_interleave(st, ln) + (torch.arange(int(ln.sum()), device=d_) -
                                                         torch.repeat_interleave(torch.cumsum(ln, 0) - ln, ln))
                rows.append(vals[idx].long())
        touched[table.name] = torch.unique(torch.cat(rows))
    after = _tables(model)
    for name, (E0, a0, b0) in before.items():
        E1, a1, b1 = after[name]
        mask = torch.ones(E0.shape[0], dtype=torch.bool, device=d_)
        mask[touched[name]] = False
        assert torch.equal(E0[mask], E1[mask]) and torch.equal(a0[mask], a1[mask]), name   # locality
        assert float((a1[touched[name]] - a0[touched[name]]).min().item()) >= 0.0
        assert bool((a1 >= 0.1 - 1e-7).all())
        assert int((E0 != E1).any(1).sum().item()) > 0.2 * len(touched[name])
    # the same batch again: the loss falls
    l1 = model.step(None, u, i, None, None, None, loss='mw')
    l2 = model.step(None, u, i, None, None, None, loss='mw')
    assert l2 < l1 < loss


def test_fullsize_lstm_reproducible_clipped_and_learning(dev):
    """configs[3] at full size (d = h = 64, L = 50, 1 M items, S = 1024, B = 1024 sequences):
    graph replay and eager execution agree bit for bit, the global norm the step clipped with is
    the norm of what it applied (clip_by_global_norm: applied update = coef * gradient, coef =
    clip / max(norm, clip)), and a repeated batch gets cheaper."""
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.lstm.seqModel import SeqModel
    from arx.utils.synthetic import SyntheticHMF
    Bq, L, size, Sq = 1024, 50, 64, 1024
    runs = []
    for use_graph in (True, False):
        syn = SyntheticHMF(n_users=N, n_items=N, permute_logits=False, seed=0)
        syn.u_attr.set_model_size(size)
        syn.i_attr.set_model_size(size)
        emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, Bq, Sq, L, False, None, syn.logit_ind2item_ind)
        emb.rt.use_graph = use_graph
        model = SeqModel([L], size, 1, 5.0, Bq, 0.5, 0.99, emb, loss='mw', use_concat=False, START_ID=N)
        emb.prepare_warp(syn.positives_csr(), syn.positives_csr())
        d_ = model.rt.device
        rng = np.random.default_rng(1)
        users = torch.from_numpy(rng.integers(0, N, size=Bq).astype(np.int32)).to(d_)
        tg = np.stack([syn.sample_batch(Bq, rng)[1] for _ in range(L)], 0).astype(np.int32)
        inp = np.concatenate([np.full((1, Bq), N, dtype=np.int32), tg[:-1]], 0)
        lens = rng.integers(10, L + 1, size=Bq)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float32)
        pool = torch.from_numpy(syn.sample_pool(Sq, rng).astype(np.int32)).to(d_)
        args = (users, torch.from_numpy(inp).to(d_), torch.from_numpy(tg).to(d_), torch.from_numpy(w).to(d_))
        W0 = model.W.w.clone()
        losses, norms = [], []
        for k in range(4):
            losses.append(model.step(None, *args, 0, pool if k == 0 else None, None))
            norms.append(float(model._gnorm.item()))
        runs.append((losses, norms, model.W.w.clone(), model.b.w.clone(), W0, model))
    (l_g, n_g, W_g, b_g, W0, m_g), (l_e, n_e, W_e, b_e, _, m_e) = runs
    assert l_g == l_e and n_g == n_e
    assert torch.equal(W_g, W_e) and torch.equal(b_g, b_e)
    for (name, t1), (_, t2) in zip(sorted(m_g.att_emb.get_params().items()), sorted(m_e.att_emb.get_params().items())):
        assert np.array_equal(t1, t2), name
    assert all(np.isfinite(n_g)) and max(n_g) > 0
    assert l_g[-1] < l_g[1] < l_g[0]                                 # same batch, four steps
    # the clip coefficient of the last step is consistent with its norm
    coef = float(m_g.rt.clip_coef_dev.item())
    np.testing.assert_allclose(coef, 5.0 / max(n_g[-1], 5.0), rtol=1e-5)
    assert not torch.equal(W_g, W0)


def test_fullsize_integer_paths_bit_exact(dev):
    """Index work at full size, bit-exact against vectorised numpy / torch: ragged CSR expansion of
    65536 bags of a 1 M-item multi-hot attribute (compact and padded forms), the weighted sampler
    over 1 M items, top-k over a 1 M-column score row."""
    from arx import ops
    from arx.utils.synthetic import SyntheticHMF
    syn = SyntheticHMF(n_users=1000, n_items=N, item_mulhot=True, permute_logits=False, seed=3)
    ia = syn.i_attr
    vals = np.asarray(ia.features_mulhot[0], dtype=np.int32)
    starts = np.asarray(ia.mulhot_starts[0], dtype=np.int32)
    lens = np.asarray(ia.mulhot_lengths[0], dtype=np.int32)
    rng = np.random.default_rng(0)
    ids = rng.integers(0, N, size=65536).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ws = ops.Workspace(dev)
    ln = lens[ids].astype(np.int64)
    total = int(ln.sum())
    cap = total + 1000
    tok, seg, offs, tot, _ = ops.csr_expand(t(vals), t(starts), t(lens), t(ids), cap, ws, pad_token=-7, pad_seg=-1)
    exp_offs = np.concatenate([[0], np.cumsum(ln)])
    exp_seg = np.repeat(np.arange(len(ids)), ln)
    exp_tok = vals[np.repeat(starts[ids].astype(np.int64), ln) + (np.arange(total) - np.repeat(exp_offs[:-1], ln))]
    assert int(tot.item()) == total
    assert np.array_equal(offs.cpu().numpy(), exp_offs)
    assert np.array_equal(seg.cpu().numpy()[:total], exp_seg) and np.array_equal(tok.cpu().numpy()[:total], exp_tok)
    assert np.all(tok.cpu().numpy()[total:] == -7) and np.all(seg.cpu().numpy()[total:] == -1)
    # padded form: the live entries, in order, are the compact form
    mx = int(lens.max())
    k = torch.empty(len(ids) * mx, dtype=torch.int32, device=dev)
    s_ = torch.empty_like(k)
    c = torch.empty(len(ids) * mx, dtype=torch.float32, device=dev)
    ops.bag_expand_padded(t(vals), t(starts), t(lens), t(ids), mx, 5, 1.0, k, s_, c)
    kk, ss, cc = k.cpu().numpy(), s_.cpu().numpy(), c.cpu().numpy()
    live = kk != ops.KEY_NONE
    assert int(live.sum()) == total and np.array_equal(kk[live], exp_tok) and np.array_equal(ss[live], exp_seg + 5)
    np.testing.assert_array_equal(cc[live], (1.0 / np.repeat(ln, ln).astype(np.float32)).astype(np.float32))
    assert np.all(cc[~live] == 0.0)
    # sampler: S distinct positions with positive weight, reproducible
    w = rng.random(N).astype(np.float32) ** 2
    out = torch.empty(1024, dtype=torch.int32, device=dev)
    ops.sample_wor(t(w), 1024, seed=11, counter=5, out=out, ws=ws)
    a = out.cpu().numpy()
    assert len(np.unique(a)) == 1024 and a.min() >= 0 and a.max() < N and np.all(w[a] > 0)
    # top-k over 1 M columns: exactly torch's (values and, no ties in random floats, indices)
    x = torch.randn(8, N, device=dev)
    v = torch.empty((8, 100), dtype=torch.float32, device=dev)
    ix = torch.empty((8, 100), dtype=torch.int32, device=dev)
    ops.topk(x, 100, v, ix)
    tv, ti = torch.topk(x, 100, dim=1, largest=True, sorted=True)
    assert torch.equal(v, tv) and torch.equal(ix.long(), ti)


def _cmp_rows(name, st, E_gpu, acc_gpu, rtol=1e-4, atol=2e-6):
    """Rows the oracle updated: GPU value / slot == oracle value / slot."""
    idx = torch.from_numpy(st.idx).to(E_gpu.device)
    got = E_gpu[idx].double().cpu().numpy().reshape(st.val.shape)
    np.testing.assert_allclose(got, st.val, rtol=rtol, atol=atol, err_msg=name)
    got = acc_gpu[idx].double().cpu().numpy().reshape(st.acc.shape)
    np.testing.assert_allclose(got, st.acc, rtol=rtol, atol=atol, err_msg=name + '/Adagrad')


def _check_tables(model, oracle):
    for t in model.att_emb.tables.values():
        st = oracle.t[t.name]
        assert len(st.idx) > 0, t.name
        _cmp_rows(t.name, st, t.E, t.acc)
        if t.bias is not None:
            _cmp_rows(t.bias_name, oracle.t[t.bias_name], t.bias, t.bias_acc)


@pytest.mark.parametrize("layout,loss", [('id', 'mw'), ('het', 'mw'), ('mix', 'mw'), ('id', 'mce'), ('het', 'mce')])
def test_fullsize_hmf_matches_embedding_space_oracle(dev, layout, loss):
    """C2 / C3 (HET and MIX layouts) at 1 M x 1 M, d = 128, B = 16384, S = 1024: three consecutive
    'mw' steps (pool drawn at step 0, fresh batch every step) against oracle/ref_embed.py; round 6: the same with
    the build-defined sampled softmax 'mce' on its fused family at d = 128 (no [B, S] array on the device).
    Checked after step 1 and after step 3: the loss of every step (rtol 1e-4), every touched row
    of every table and bias with its Adagrad slot (rtol 1e-4); rows the oracle did not touch are
    covered by the locality test above."""
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.synthetic import SyntheticHMF
    from oracle import ref_embed
    kw = {'id': {}, 'het': dict(item_mulhot=True), 'mix': dict(item_mix=True)}[layout]
    syn = SyntheticHMF(n_users=N, n_items=N, permute_logits=False, seed=0, **kw)
    model = LatentProductModel(N, N, D, 1, B, 0.1, 1.0, syn.u_attr, syn.i_attr, syn.item2logit[:N],
                               syn.logit_ind2item_ind, loss_function=loss, n_sampled=S, seed=0)
    model.prepare_warp(syn.positives_csr(), syn.positives_csr())
    params0 = model.att_emb.get_params()
    oracle = ref_embed.EmbedSpaceHMF(syn.u_attr, syn.i_attr, params0, 0.1, loss=loss)
    ptr, pit = syn.positives_csr()
    d_ = model.rt.device
    rng = np.random.default_rng(1)
    pool = syn.sample_pool(S, rng).astype(np.int32)
    for step in range(3):
        u, i = syn.sample_batch(B, rng)
        if step == 0:
            pool[:8] = i[:8]                      # targets inside the pool: masked columns
            pool = np.unique(pool)
            pool = np.concatenate([pool, np.setdiff1d(syn.item_population[:4 * S], pool)[:S - len(pool)]]).astype(np.int32)
        ps = pool if step == 0 else None
        l_ref = oracle.step(u, i, ps, ptr, pit)
        l_got = model.step(None, torch.from_numpy(u).to(d_), torch.from_numpy(i).to(d_), None,
                           torch.from_numpy(ps).to(d_) if ps is not None else None, None, loss=loss)
        np.testing.assert_allclose(l_got, l_ref, rtol=1e-4, err_msg='loss, step %d' % step)
        if step in (0, 2):
            _check_tables(model, oracle)
    from conftest import assert_scorer_path
    assert_scorer_path(model._plan('train'), B, S, D, loss)             # the fused family of the loss ran (unless switched off)


@pytest.mark.parametrize("loss", ['mw', 'mce'])
def test_fullsize_lstm_matches_embedding_space_oracle(dev, loss):
    """configs[3] at full size (d = h = 64, L = 50, B = 1024 sequences, 1 M items, S = 1024, 'mw' and the
    build-defined sampled softmax 'mce', clip 5.0 -- active): two consecutive steps against ref_embed.EmbedSpaceSeq: summed sequence
    loss, the global norm the step clipped with, LSTM weights / biases, every touched table row."""
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.lstm.seqModel import SeqModel
    from arx.utils.synthetic import SyntheticHMF
    from oracle import ref_embed
    Bq, L, size, Sq = 1024, 50, 64, 1024
    syn = SyntheticHMF(n_users=N, n_items=N, permute_logits=False, seed=0)
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, Bq, Sq, L, False, None, syn.logit_ind2item_ind)
    model = SeqModel([L], size, 1, 5.0, Bq, 0.5, 0.99, emb, loss=loss, use_concat=False, START_ID=N)
    emb.prepare_warp(syn.positives_csr(), syn.positives_csr())
    params0 = emb.get_params()
    oracle = ref_embed.EmbedSpaceSeq(syn.u_attr, syn.i_attr, params0, model.W.w.cpu().numpy(),
                                     model.b.w.cpu().numpy(), 0.5, 5.0, loss=loss, no_user_id=True)
    ptr, pit = syn.positives_csr()
    d_ = model.rt.device
    rng = np.random.default_rng(1)
    pool = syn.sample_pool(Sq, rng).astype(np.int32)
    for step in range(2):
        users = rng.integers(0, N, size=Bq).astype(np.int32)
        tg = np.stack([syn.sample_batch(Bq, rng)[1] for _ in range(L)], 0).astype(np.int32)
        inp = np.concatenate([np.full((1, Bq), N, dtype=np.int32), tg[:-1]], 0)
        lens = rng.integers(10, L + 1, size=Bq)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float32)
        ps = pool if step == 0 else None
        l_ref = oracle.step(users, inp, tg, w, ps, ptr, pit)
        t = lambda a: torch.from_numpy(a).to(d_)
        l_got = model.step(None, t(users), t(inp), t(tg), t(w), 0, t(ps) if ps is not None else None, None)
        np.testing.assert_allclose(l_got, l_ref, rtol=1e-4, err_msg='loss, step %d' % step)
        np.testing.assert_allclose(float(model._gnorm.item()), oracle.last['gnorm'], rtol=1e-4)
        assert oracle.last['gnorm'] > 5.0                             # the clip is active
        np.testing.assert_allclose(model.W.w.cpu().numpy(), oracle.W, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(model.b.w.cpu().numpy(), oracle.b, rtol=1e-4, atol=2e-6)
        for tb in emb.tables.values():
            st = oracle.t[tb.name]
            if len(st.idx):
                _cmp_rows(tb.name, st, tb.E, tb.acc)
                if tb.bias is not None:
                    _cmp_rows(tb.bias_name, oracle.t[tb.bias_name], tb.bias, tb.bias_acc)


def test_fullsize_streaming_eval(dev):
    """Evaluation loss of the C3 model ('warp' over all 1 M logits, B = 16384): streamed -- the 64 GB
    [B, V] logits never exist -- and, for a sample of rows, equal to an fp64 numpy evaluation of
    the same rows against the model's tables (positives of the row's user masked)."""
    from arx import graph as G
    syn, model = _model(True, mw_eval_unmasked=False)     # the masked form: positives taken out of the streamed sums
    assert isinstance(model.loss_eval.inputs[0], G.StreamEvalLoss)
    d_ = model.rt.device
    batches, pool = _batches(syn, d_, 1)
    u, i = batches[0]
    model.step(None, u, i, None, pool, None, loss='mw')
    e = model.step(None, u, i, None, None, None, forward_only=True, loss='mw')
    bl = model.loss_eval.inputs[0].value.cpu().numpy()
    assert np.isfinite(e) and abs(e - bl.mean()) <= 1e-5 * abs(e)
    # fp64 check of 6 rows
    P = model.att_emb.get_params()
    ia = syn.i_attr
    icat = np.asarray(ia.features_cat[0])
    vals, st, ln = (np.asarray(x) for x in (ia.features_mulhot[0], ia.mulhot_starts[0], ia.mulhot_lengths[0]))
    Eid, bid = P['itemembed_cat_0'].astype(np.float64), P['item_bias_cat_0'].astype(np.float64)[:, 0]
    Em, bm = P['itemembed_mulhot_0'].astype(np.float64), P['item_bias_mulhot_0'].astype(np.float64)[:, 0]
    Eu = P['userembed_cat_0']
    ucat = np.asarray(syn.u_attr.features_cat[0])
    # full pool embedding (logit j = item j here): 0.5 * (id row + mean of the bag rows)
    lens = ln[:N].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    bag = np.add.reduceat(Em[vals[:lens.sum()]], offs, axis=0) / lens[:, None]
    bagb = np.add.reduceat(bm[vals[:lens.sum()]], offs) / lens
    pool_emb = 0.5 * (Eid[icat[:N]] + bag)
    pool_b = 0.5 * (bid[icat[:N]] + bagb)
    un, it = u.cpu().numpy(), i.cpu().numpy()
    ptr, pit = syn.positives_csr()
    for r in (0, 1, 77, 4095, 9999, 16383):
        uu = Eu[ucat[un[r]]].astype(np.float64)
        x = pool_emb @ uu + pool_b
        t = x[it[r]]
        m = np.ones(N, dtype=bool)
        m[pit[ptr[un[r]]:ptr[un[r] + 1]]] = False
        want = np.log1p(np.maximum(x - t + 1.0, 0.0)[m].sum())
        np.testing.assert_allclose(bl[r], want, rtol=1e-4, err_msg='row %d' % r)
