"""On-device negative-pool sampler (arx_sample_wor / DeviceSampler) against the distribution of
the reference's host call  np.random.choice(items, S, replace=False, p)  (utils/prepare_train.py:7-17):
same law (sequential weighted draws without replacement), different random stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(dev, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_sample_wor_basic_properties(dev):
    from arx import ops
    import torch
    rng = np.random.default_rng(0)
    n, S = 100000, 1024
    w = rng.random(n).astype(np.float32) ** 3
    w[rng.choice(n, size=n // 4, replace=False)] = 0.0           # never drawn
    ws = ops.Workspace(dev)
    out = torch.empty(S, dtype=torch.int32, device=dev)
    ops.sample_wor(_t(dev, w), S, seed=7, counter=3, out=out, ws=ws)
    a = out.cpu().numpy()
    assert a.min() >= 0 and a.max() < n
    assert len(np.unique(a)) == S                                 # without replacement
    assert np.all(w[a] > 0)
    out2 = torch.empty(S, dtype=torch.int32, device=dev)
    ops.sample_wor(_t(dev, w), S, seed=7, counter=3, out=out2, ws=ws)
    assert torch.equal(out, out2)                                 # deterministic in (seed, counter)
    ops.sample_wor(_t(dev, w), S, seed=7, counter=4, out=out2, ws=ws)
    assert not torch.equal(out, out2)
    # heavier items are drawn far more often than light ones
    assert w[a].mean() > 2.0 * w[w > 0].mean()
    # fewer positive weights than S -> the tail is -1
    w2 = np.zeros(50, dtype=np.float32)
    w2[:7] = 1.0
    o3 = torch.empty(10, dtype=torch.int32, device=dev)
    ops.sample_wor(_t(dev, w2), 10, seed=1, counter=0, out=o3, ws=ws)
    b = o3.cpu().numpy()
    assert sorted(b[:7].tolist()) == list(range(7)) and np.all(b[7:] == -1)


def test_sample_wor_matches_numpy_choice_distribution(dev):
    """Inclusion frequencies and first-draw marginals over many draws vs numpy's sequential
    sampler (the reference's host call) on a small skewed item set."""
    from arx import ops
    import torch
    n, S, draws = 40, 6, 6000
    p = (1.0 / np.arange(1, n + 1) ** 1.1)
    p = (p / p.sum()).astype(np.float64)
    rs = np.random.RandomState(123)
    ref_incl = np.zeros(n)
    ref_first = np.zeros(n)
    for _ in range(draws):
        s = rs.choice(n, S, replace=False, p=p)
        ref_incl[s] += 1
        ref_first[s[0]] += 1
    w = _t(dev, p.astype(np.float32))
    ws = ops.Workspace(dev)
    outs = torch.empty((draws, S), dtype=torch.int32, device=dev)
    for k in range(draws):
        ops.sample_wor(w, S, seed=99, counter=k, out=outs[k], ws=ws)
    got = outs.cpu().numpy()
    got_incl = np.bincount(got.reshape(-1), minlength=n).astype(float)
    got_first = np.bincount(got[:, 0], minlength=n).astype(float)
    # first draw is exactly ~ p
    exp_first = p * draws
    z = (got_first - exp_first) / np.sqrt(exp_first * (1 - p) + 1e-9)
    assert np.abs(z).max() < 4.5, z
    # inclusion counts: two independent Monte-Carlo estimates of the same probabilities
    pi = (ref_incl + got_incl) / (2.0 * draws)
    sd = np.sqrt(2.0 * draws * pi * (1 - pi) + 1e-9)
    z2 = (got_incl - ref_incl) / sd
    assert np.abs(z2).max() < 4.5, z2
    assert abs(got_incl.sum() - draws * S) < 1e-6


def test_device_sampler_feeds_model_pool(dev):
    """DeviceSampler -> LatentProductModel.step(item_sampled=<device tensor>): the pool never
    touches the host."""
    from arx.utils.prepare_train import DeviceSampler
    from arx.utils.synthetic import SyntheticHMF
    from arx.hmf.hmf_model import LatentProductModel
    syn = SyntheticHMF(n_users=300, n_items=400, logit_size=400, seed=2)
    S, B, d = 64, 32, 32
    model = LatentProductModel(syn.n_users, syn.n_items, d, 1, B, 0.3, 1.0, syn.u_attr, syn.i_attr,
                               syn.item_ind2logit_ind_dict(), syn.logit_ind2item_ind,
                               loss_function='mw', n_sampled=S)
    pos = syn.positives_dict()
    model.prepare_warp(pos, pos)
    counts = np.bincount(syn.pos_items, minlength=syn.n_items).astype(np.float64) + 1.0
    pw = (counts / counts.sum()) ** 0.5
    sampler = DeviceSampler(np.arange(syn.n_items), pw / pw.sum(), device=dev, seed=5)
    rng = np.random.default_rng(1)
    losses = []
    for step in range(6):
        users, items = syn.sample_batch(B, rng)
        pool = sampler.sample(S) if step % 2 == 0 else None
        losses.append(model.step(None, list(users), list(items), None, pool, None, loss='mw'))
    assert np.all(np.isfinite(losses))
    p = sampler.sample(S).cpu().numpy()
    assert len(np.unique(p)) == S and p.min() >= 0 and p.max() < syn.n_items


def test_item_frequency_on_device_matches_reference_helper(dev):
    """arx_item_frequency (counts by integer atomics, p ~ (count/total)^power) against the host
    helper that is pinned bit for bit to the reference (utils/prepare_train.py:19-35): same
    probabilities for the seen items, zero for the unseen ones; DeviceSampler.from_interactions
    draws from it."""
    import torch
    from arx.utils.prepare_train import DeviceSampler, item_frequency
    rng = np.random.default_rng(4)
    n_items, n = 5000, 60000
    p = 1.0 / np.arange(1, n_items + 1) ** 1.1
    items = rng.permutation(n_items)[rng.choice(n_items, size=n, p=p / p.sum())].astype(np.int32)
    data = [(0, int(i), 0) for i in items]
    pop, p_ref = item_frequency(data, 0.5)
    s = DeviceSampler.from_interactions(torch.from_numpy(items).to(dev), n_items, power=0.5, device=dev, seed=3)
    w = s.w.cpu().numpy().astype(np.float64)
    cnt = s.counts.cpu().numpy()
    assert np.array_equal(cnt, np.bincount(items, minlength=n_items))            # integer path: exact
    np.testing.assert_allclose(w[pop] / w.sum(), np.asarray(p_ref), rtol=1e-5)
    unseen = np.setdiff1d(np.arange(n_items), pop)
    assert np.all(w[unseen] == 0.0)
    draw = s.sample(256).cpu().numpy()
    assert len(np.unique(draw)) == 256 and np.all(cnt[draw] > 0)                 # only seen items


def test_capped_race_equals_uncapped(dev):
    """arx_sample_wor_capped: dropping the keys above 8 S / sum(w) before the sort does not change the
    draw (the S smallest keys are below the cap), for skewed weights over 3 M items."""
    import torch
    from arx import ops
    rng = np.random.default_rng(9)
    n, S = 3000000, 1024
    w = (rng.random(n) ** 6).astype(np.float32)
    w[rng.integers(0, n, 1000)] = 0.0
    tw = torch.from_numpy(w).to(dev)
    ws = ops.Workspace(dev)
    a = torch.empty(S, dtype=torch.int32, device=dev)
    b = torch.empty(S, dtype=torch.int32, device=dev)
    for counter in (0, 7):
        ops.sample_wor(tw, S, 5, counter, a, ws)
        ops.sample_wor(tw, S, 5, counter, b, ws, key_cap=8.0 * S / float(w.astype(np.float64).sum()))
        assert torch.equal(a, b) and int(a.min().item()) >= 0


@pytest.mark.parametrize("n,S", [(1000003, 1024), (262147, 64), (70001, 2048), (500002, 4096)])
def test_capped_race_survivor_list_edges(dev, n, S):
    """The capped race of S <= 2048 never stores the n keys (k_race_compact -> k_bitonic_take): item
    counts that are not multiples of 4, survivors in the tail, keys out == the un-capped race's, a cap
    that lets more than the list's 16384 entries through -> all -1; S = 4096 takes the sort path."""
    import torch
    from arx import ops
    rng = np.random.default_rng(n)
    w = (rng.random(n) ** 3).astype(np.float32)
    w[-3:] = 1e7                                                    # the tail items are (almost) always drawn
    tw = torch.from_numpy(w).to(dev)
    ws, ws2 = ops.Workspace(dev), ops.Workspace(dev)
    a, b = (torch.empty(S, dtype=torch.int32, device=dev) for _ in range(2))
    ka, kb = (torch.empty(S, dtype=torch.float32, device=dev) for _ in range(2))
    cap = 7.5 * S / float(w[:-3].astype(np.float64).sum())         # (w t << 1 for all but the three heavy items)
    for counter in (1, 2):
        ops.sample_wor(tw, S, 11, counter, a, ws, out_keys=ka)
        ops.sample_wor(tw, S, 11, counter, b, ws2, key_cap=cap, out_keys=kb)
        assert torch.equal(a, b) and torch.equal(ka, kb) and int(a.min().item()) >= 0
        assert set(range(n - 3, n)) <= set(a.cpu().numpy().tolist())
        assert bool((ka[1:] >= ka[:-1]).all())
    if S <= 2048:
        assert ws2.buf.numel() == (1 << 20)                              # no 28 n-byte workspace on this path
        ops.sample_wor(tw, S, 11, 3, b, ws2, key_cap=1e30)           # every positive weight survives the cap
        assert int(b.max().item()) == -1


def test_device_sampler_heavy_tailed_weights(dev):
    """DeviceSampler on Zipf(1.5)-like weights (round-2 advisor finding): with the cap 8 S / sum(w)
    only ~600 of 1 M keys survived, the draw of 1000 came back short and the missing positions
    indexed items[-1].  The cap now solves sum_i (1 - exp(-w_i t)) = 8 S: the draw is complete,
    duplicate-free and equal to the un-capped race; too few positive weights raise."""
    import torch
    from arx import ops
    from arx.utils.prepare_train import DeviceSampler
    n, S = 1000000, 1000
    w = (1.0 / np.arange(1, n + 1) ** 1.5).astype(np.float32)
    items = np.arange(n, dtype=np.int32)[::-1].copy()            # item ids != positions
    s = DeviceSampler(items, w, device=dev, seed=3)
    assert s._cap_for(S) > 8.0 * S / float(w.astype(np.float64).sum())      # the old cap was too tight
    tw = torch.from_numpy(w).to(dev)
    ws = ops.Workspace(dev)
    ref = torch.empty(S, dtype=torch.int32, device=dev)
    for counter in range(3):
        got = s.sample(S).cpu().numpy()
        ops.sample_wor(tw, S, 3, counter, ref, ws)                # un-capped race, same seed / counter
        assert len(np.unique(got)) == S
        np.testing.assert_array_equal(got, items[ref.cpu().numpy()])
    few = np.zeros(200000, dtype=np.float32)
    few[:10] = 1.0
    with pytest.raises(ValueError):
        DeviceSampler(np.arange(200000, dtype=np.int32), few, device=dev).sample(64)


@pytest.mark.parametrize("world,S", [(2, 60), (8, 1024), (3, 1000), (16, 1024)])
def test_merge_keyed_take_equals_stable_argsort(dev, world, S):
    """arx_merge_keyed_take (the merge of the ranks' race lists in arx.dist.draw_global_pool): the S smallest of the
    world * S (key, id) pairs in (key, position) order == torch's stable argsort, incl. tied keys across ranks,
    +inf / id -1 tails of short shards and a zero key."""
    import torch
    from arx import ops
    rng = np.random.default_rng(world * 1000 + S)
    keys = np.sort(rng.exponential(size=(world, S)).astype(np.float32), axis=1)
    ids = rng.permutation(world * S).astype(np.int32).reshape(world, S)
    keys[1, :5] = keys[0, :5]                      # ties between ranks: the lower rank first
    keys[0, 0] = 0.0
    keys[-1, S // 2:] = np.inf                     # a short shard
    ids[-1, S // 2:] = -1
    tk, ti = torch.from_numpy(keys.reshape(-1)).to(dev), torch.from_numpy(ids.reshape(-1)).to(dev)
    out = torch.full((S,), -7, dtype=torch.int32, device=dev)
    ops.merge_keyed_take(tk, ti, S, out)
    want = ti[torch.argsort(tk, stable=True)[:S]]
    assert torch.equal(out, want)
