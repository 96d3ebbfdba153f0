"""Per-kernel parity: every C-ABI entry point against a numpy restatement of the
TF op it replaces (oracle.ref_graph helpers / plain numpy), on seeded inputs.
Integer outputs bit-exact; fp32 rtol 1e-4 (north_star tolerance)."""
import numpy as np
import pytest

from oracle import ref_graph as rg

pytestmark = pytest.mark.gpu

RTOL = 1e-4
ATOL = 1e-5


def _t(dev, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _csr(rng, n_rows, vocab, max_len, zipf=False, min_len=1):
    lens = rng.integers(min_len, max_len + 1, size=n_rows).astype(np.int32)
    starts = np.zeros(n_rows + 1, dtype=np.int32)
    starts[1:] = np.cumsum(lens)
    if zipf:
        p = 1.0 / np.arange(1, vocab + 1)
        p /= p.sum()
        vals = rng.choice(vocab, size=int(starts[-1]), p=p).astype(np.int32)
    else:
        vals = rng.integers(0, vocab, size=int(starts[-1])).astype(np.int32)
    return vals, starts[:-1].copy(), lens


@pytest.mark.parametrize("B,max_len", [(1, 1), (7, 5), (256, 64), (1000, 20), (5000, 8)])
def test_csr_expand_bit_exact(dev, B, max_len):
    from arx import ops
    rng = np.random.default_rng(B)
    n_rows = 3000
    vals, starts, lens = _csr(rng, n_rows, 50000, max_len)
    ids = rng.integers(0, n_rows, size=B).astype(np.int32)
    exp_tok = rg.batch_slice2(vals, starts[ids], lens[ids])
    exp_seg = rg.batch_segids2(lens[ids])
    T = len(exp_tok)
    cap = T + 37
    ws = ops.Workspace(dev)
    tok, seg, offs, tot, coef = ops.csr_expand(_t(dev, vals), _t(dev, starts), _t(dev, lens),
                                               _t(dev, ids), cap, ws, pad_token=-7, pad_seg=-9,
                                               seg_base=100, coef_scale=0.5, want_coef=True)
    tok, seg, offs, tot, coef = [x.cpu().numpy() for x in (tok, seg, offs, tot, coef)]
    assert int(tot[0]) == T
    np.testing.assert_array_equal(tok[:T], exp_tok)
    np.testing.assert_array_equal(seg[:T], exp_seg + 100)
    np.testing.assert_array_equal(tok[T:], -7)
    np.testing.assert_array_equal(seg[T:], -9)
    np.testing.assert_array_equal(offs, np.concatenate([[0], np.cumsum(lens[ids])]))
    np.testing.assert_allclose(coef[:T], 0.5 / lens[ids][exp_seg].astype(np.float32), rtol=1e-7)
    assert np.all(coef[T:] == 0)


def test_csr_expand_empty_and_zero_len(dev):
    from arx import ops
    import torch
    ws = ops.Workspace(dev)
    vals = _t(dev, np.arange(10, dtype=np.int32))
    starts = _t(dev, np.array([0, 3, 3, 6], dtype=np.int32))
    lens = _t(dev, np.array([3, 0, 3, 0], dtype=np.int32))
    ids = _t(dev, np.array([1, 0, 3, 2, 1], dtype=np.int32))
    tok, seg, offs, tot, _ = ops.csr_expand(vals, starts, lens, ids, 8, ws, pad_token=-1, pad_seg=-1)
    assert int(tot.item()) == 6
    np.testing.assert_array_equal(tok.cpu().numpy(), [0, 1, 2, 3, 4, 5, -1, -1])
    np.testing.assert_array_equal(seg.cpu().numpy(), [1, 1, 1, 3, 3, 3, -1, -1])
    empty = torch.empty(0, dtype=torch.int32, device=dev)
    tok, seg, offs, tot, _ = ops.csr_expand(vals, starts, lens, empty, 4, ws, pad_token=-1, pad_seg=-1)
    assert int(tot.item()) == 0 and int(offs[0].item()) == 0
    np.testing.assert_array_equal(tok.cpu().numpy(), [-1] * 4)


@pytest.mark.parametrize("d", [4, 20, 32, 64, 128, 256])
def test_gather_onehot(dev, d):
    from arx import ops
    import torch
    rng = np.random.default_rng(d)
    Vf, N, B = 1000, 700, 333
    E = rng.standard_normal((Vf, d)).astype(np.float32)
    bias = rng.standard_normal((Vf,)).astype(np.float32)
    cmap = rng.integers(0, Vf, size=N).astype(np.int32)
    ids = rng.integers(0, N, size=B).astype(np.int32)
    out = torch.full((B, d), 3.0, dtype=torch.float32, device=dev)
    bout = torch.full((B,), 2.0, dtype=torch.float32, device=dev)
    ops.gather_onehot(_t(dev, E), _t(dev, bias), _t(dev, cmap), _t(dev, ids), out, scale=0.5,
                      accumulate=True, bias_out=bout)
    np.testing.assert_allclose(out.cpu().numpy(), 3.0 + 0.5 * E[cmap[ids]], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bout.cpu().numpy(), 2.0 + 0.5 * bias[cmap[ids]], rtol=RTOL, atol=ATOL)
    out2 = torch.empty((B, d), dtype=torch.float32, device=dev)
    ops.gather_onehot(_t(dev, E), None, None, _t(dev, cmap[ids]), out2)
    np.testing.assert_array_equal(out2.cpu().numpy(), E[cmap[ids]])


@pytest.mark.parametrize("d,sizes", [(128, [700, 0, 33, 1]), (32, [5, 130]), (16, [257, 64, 3, 0, 19]),
                                     (128, [16384, 1024, 16384])])
def test_gather_onehot_multi(dev, d, sizes):
    """One launch for several one-hot lookups (embed_attribute.py:371-380 per feature): every
    workgroup serves one site -- ragged and EMPTY sites, with / without cat_map and bias."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + len(sizes))
    sites, want = [], []
    for k, n in enumerate(sizes):
        V = 50 + 37 * k
        E = rng.standard_normal((V, d)).astype(np.float32)
        bias = rng.standard_normal((V,)).astype(np.float32) if k % 2 == 0 else None
        N = V + 11
        cmap = rng.integers(0, V, size=N).astype(np.int32) if k % 3 != 1 else None
        ids = rng.integers(0, N if cmap is not None else V, size=n).astype(np.int32)
        rows = cmap[ids] if cmap is not None else ids
        scale = 1.0 if k == 0 else 0.25 * (k + 1)
        out = torch.full((max(n, 1), d + 4), 7.0, dtype=torch.float32, device=dev)[:n, :d]
        bout = torch.full((max(n, 1),), 5.0, dtype=torch.float32, device=dev)[:n] if bias is not None else None
        sites.append((_t(dev, E), _t(dev, bias) if bias is not None else None,
                      _t(dev, cmap) if cmap is not None else None, _t(dev, ids), out, scale, bout))
        want.append((scale * E[rows], scale * bias[rows] if bias is not None else None))
    ops.gather_onehot_multi(ops.GatherSet(sites))
    for (E_, b_, c_, i_, out, sc, bout), (w, wb) in zip(sites, want):
        np.testing.assert_allclose(out.cpu().numpy(), w, rtol=1e-6, atol=0)
        if wb is not None:
            np.testing.assert_allclose(bout.cpu().numpy(), wb, rtol=1e-6, atol=0)


@pytest.mark.parametrize("d,sizes", [(128, [300, 77, 1030]), (64, [5, 0, 9])])
def test_gather_onehot_multi_packed_sites(dev, d, sizes):
    """arx_gather_onehot_multi_ld: sites whose bias goes to column d of their packed out rows ('packed'),
    and a site that reorders PACKED rows and splits off their bias column (bias = column d of the table
    itself) -- the sharded step's fused lookups (arx.dist.ShardedHMF._step_static)."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d)
    V = 211
    E = rng.standard_normal((V, d)).astype(np.float32)
    bias = rng.standard_normal((V,)).astype(np.float32)
    tE, tb = _t(dev, E), _t(dev, bias)
    ids = [rng.integers(0, V, size=n).astype(np.int32) for n in sizes]
    plain = torch.full((max(sizes[0], 1), d), 3.0, dtype=torch.float32, device=dev)[:sizes[0]]
    vec = torch.full((max(sizes[0], 1),), 3.0, dtype=torch.float32, device=dev)[:sizes[0]]
    packed = [torch.full((max(n, 1), d + 4), 9.0, dtype=torch.float32, device=dev)[:n] for n in sizes[1:]]
    ops.gather_onehot_multi(ops.GatherSet(
        [(tE, tb, None, _t(dev, ids[0]), plain, 1.0, vec)] +
        [(tE, tb, None, _t(dev, i), o, 1.0, 'packed') for i, o in zip(ids[1:], packed)]))
    np.testing.assert_array_equal(plain.cpu().numpy(), E[ids[0]])
    np.testing.assert_array_equal(vec.cpu().numpy(), bias[ids[0]])
    for i, o in zip(ids[1:], packed):
        got = o.cpu().numpy()
        np.testing.assert_array_equal(got[:, :d], E[i])
        np.testing.assert_array_equal(got[:, d], bias[i])
        assert np.all(got[:, d + 1:] == 9.0)                              # the padding columns are not written
    # packed table in, reordered packed rows + bias vector out
    P = rng.standard_normal((V, d + 4)).astype(np.float32)
    order = rng.integers(0, V, size=sizes[2]).astype(np.int32)
    out = torch.zeros((max(sizes[2], 1), d + 4), dtype=torch.float32, device=dev)[:sizes[2]]
    bvec = torch.zeros((max(sizes[2], 1),), dtype=torch.float32, device=dev)[:sizes[2]]
    ops.gather_onehot_multi(ops.GatherSet([(_t(dev, P), d, None, _t(dev, order), out, 1.0, bvec)]))
    np.testing.assert_array_equal(out.cpu().numpy(), P[order])
    np.testing.assert_array_equal(bvec.cpu().numpy(), P[order, d])


@pytest.mark.parametrize("d,B", [(128, 1000), (32, 7), (64, 0)])
def test_gather_onehot_packed_and_strided_copy(dev, d, B):
    """Packed rows of the sharded exchanges: [row | bias | pad]; the bias column <-> vector copies."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + B)
    V = 300
    E = rng.standard_normal((V, d)).astype(np.float32)
    bias = rng.standard_normal((V,)).astype(np.float32)
    ids = rng.integers(0, V, size=B).astype(np.int32)
    out = torch.full((max(B, 1), d + 4), 9.0, dtype=torch.float32, device=dev)[:B]
    ops.gather_onehot_packed(_t(dev, E), _t(dev, bias), None, _t(dev, ids), out)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, :d], E[ids])
    np.testing.assert_array_equal(got[:, d], bias[ids])
    np.testing.assert_array_equal(got[:, d + 1:], np.full((B, 3), 9.0, np.float32))   # pad untouched
    if B:
        vec = torch.zeros((B,), dtype=torch.float32, device=dev)
        ops.copy_strided(out[:, d], vec)
        np.testing.assert_array_equal(vec.cpu().numpy(), bias[ids])
        ops.copy_strided(vec, out[:, d + 2])
        np.testing.assert_array_equal(out.cpu().numpy()[:, d + 2], bias[ids])


@pytest.mark.parametrize("d,max_len,B", [(128, 64, 1000), (32, 18, 64), (64, 5, 257), (20, 3, 10),
                                         (128, 200, 50)])
def test_gather_mulhot_mean(dev, d, max_len, B):
    from arx import ops
    import torch
    rng = np.random.default_rng(d + max_len)
    Vf, N = 5000, 2000
    E = rng.standard_normal((Vf, d)).astype(np.float32)
    bias = rng.standard_normal((Vf,)).astype(np.float32)
    vals, starts, lens = _csr(rng, N, Vf, max_len, zipf=True)
    ids = rng.integers(0, N, size=B).astype(np.int32)
    tok = rg.batch_slice2(vals, starts[ids], lens[ids])
    seg = rg.batch_segids2(lens[ids])
    ref = rg.unsorted_segment_sum(E.astype(np.float64)[tok], seg, B) / lens[ids].reshape(-1, 1)
    refb = rg.unsorted_segment_sum(bias.astype(np.float64)[tok].reshape(-1, 1), seg, B)[:, 0] / lens[ids]
    out = torch.empty((B, d), dtype=torch.float32, device=dev)
    bout = torch.empty((B,), dtype=torch.float32, device=dev)
    ops.gather_mulhot_mean(_t(dev, E), _t(dev, bias), _t(dev, vals), _t(dev, starts), _t(dev, lens),
                           _t(dev, ids), out, bias_out=bout)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bout.cpu().numpy(), refb, rtol=RTOL, atol=ATOL)
    # accumulate + scale into a strided (concat) destination
    wide = torch.ones((B, d + 8), dtype=torch.float32, device=dev)
    ops.gather_mulhot_mean(_t(dev, E), None, _t(dev, vals), _t(dev, starts), _t(dev, lens),
                           _t(dev, ids), wide[:, 4:4 + d], scale=0.25, accumulate=True)
    w = wide.cpu().numpy()
    np.testing.assert_allclose(w[:, 4:4 + d], 1 + 0.25 * ref, rtol=RTOL, atol=ATOL)
    assert np.all(w[:, :4] == 1) and np.all(w[:, 4 + d:] == 1)


def test_dot_score(dev):
    from arx import ops
    import torch
    rng = np.random.default_rng(0)
    for d in (32, 128):
        B = 517
        U = rng.standard_normal((B, d)).astype(np.float32)
        T = rng.standard_normal((B, d)).astype(np.float32)
        tb = rng.standard_normal((B,)).astype(np.float32)
        s = torch.empty(B, dtype=torch.float32, device=dev)
        ops.dot_score(_t(dev, U), _t(dev, T), _t(dev, tb), s)
        np.testing.assert_allclose(s.cpu().numpy(), (U.astype(np.float64) * T).sum(1) + tb,
                                   rtol=RTOL, atol=1e-4)
        ds = rng.standard_normal((B,)).astype(np.float32)
        dU = torch.ones((B, d), dtype=torch.float32, device=dev)
        dT = torch.empty((B, d), dtype=torch.float32, device=dev)
        ops.dot_score_bwd(_t(dev, U), _t(dev, T), _t(dev, ds), dU, True, dT)
        np.testing.assert_allclose(dU.cpu().numpy(), 1 + ds[:, None] * T, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(dT.cpu().numpy(), ds[:, None] * U, rtol=RTOL, atol=ATOL)


GEMM_CASES = [
    # transA, transB, M, N, K
    (False, True, 64, 1024, 32),      # C1-like logits
    (False, True, 4096, 1024, 128),   # C2 logits (big tiles)
    (False, False, 4096, 128, 1024),  # dU (split-K)
    (True, False, 1024, 128, 4096),   # dI (split-K)
    (False, True, 64, 3100, 32),      # C1 full logits
    (True, False, 3100, 32, 64),      # C1 dI, N=32 < tile
    (False, False, 64, 32, 3100),     # C1 dU, K not multiple of 16
    (False, True, 33, 77, 19),        # ragged everything (scalar load path)
    (True, True, 129, 65, 130),
    (False, False, 3200, 64, 1024),   # LSTM dH
]


@pytest.mark.parametrize("tA,tB,M,N,K", GEMM_CASES)
def test_gemm_f32(dev, tA, tB, M, N, K):
    from arx import ops
    import torch
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    bias = rng.standard_normal((N,)).astype(np.float32)
    opA = A.T if tA else A
    opB = Bm.T if tB else Bm
    ref = 0.5 * (opA.astype(np.float64) @ opB.astype(np.float64)) + 2.0 * C0 + bias
    C = _t(dev, C0)
    ws = ops.Workspace(dev)
    ops.gemm(_t(dev, A), _t(dev, Bm), C, ws, transA=tA, transB=tB, alpha=0.5, beta=2.0,
             col_bias=_t(dev, bias))
    torch.cuda.synchronize()
    scale = np.abs(opA).astype(np.float64) @ np.abs(opB).astype(np.float64)
    err = np.abs(C.cpu().numpy() - ref)
    assert np.all(err <= 2e-6 * scale + 1e-5), float((err / (scale + 1e-9)).max())
    # transpose detection: asymmetric inputs already; also check plain alpha=1,beta=0
    C2 = torch.empty((M, N), dtype=torch.float32, device=dev)
    ops.gemm(_t(dev, A), _t(dev, Bm), C2, ws, transA=tA, transB=tB)
    ref2 = opA.astype(np.float64) @ opB.astype(np.float64)
    err2 = np.abs(C2.cpu().numpy() - ref2)
    assert np.all(err2 <= 2e-6 * scale + 1e-5)


@pytest.mark.parametrize("M,N,K", [(16384, 1024, 128), (300, 1000, 64), (129, 65, 32), (1, 1, 128),
                                   (4096, 70, 128)])
def test_gemm_nt_scorer_shape(dev, M, N, K):
    """logits = U.I^T + b with K = embedding width: the register-resident-A kernel
    (gemm_nt.hip), ragged M / N, bias and alpha in the epilogue."""
    from arx import ops
    import torch
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal((N,)).astype(np.float32)
    C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    ops.gemm(_t(dev, A), _t(dev, Bm), C, ops.Workspace(dev), transB=True, alpha=0.25,
             col_bias=_t(dev, bias))
    ref = 0.25 * (A.astype(np.float64) @ Bm.astype(np.float64).T) + bias
    scale = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64).T
    err = np.abs(C.cpu().numpy() - ref)
    assert np.all(err <= 2e-6 * scale + 1e-5), float((err / (scale + 1e-9)).max())


def _mask(rng, B, W, p=0.05):
    return (rng.random((B, W)) > p)


@pytest.mark.parametrize("kind", ['mw', 'mce'])
@pytest.mark.parametrize("B,S", [(64, 1024), (5, 100), (33, 3100)])
def test_loss_mw(dev, B, S, kind):
    """'mw' (embed_attribute.py:641-649) and the build-defined sampled softmax 'mce', mask-array form:
    wave-per-row kernel (S <= 2048, aligned) and workgroup-per-row fallback."""
    from arx import ops
    import torch
    rng = np.random.default_rng(B + S)
    logits = (rng.standard_normal((B, S)) * (3.0 if kind == 'mce' else 1.0)).astype(np.float32)
    t = rng.standard_normal((B,)).astype(np.float32)
    mask = _mask(rng, B, S)
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.dt = np.dtype(np.float64)
    bl, cache = e.compute_loss(logits.astype(np.float64), t.astype(np.float64), kind, mask)
    dl, dt = e.compute_loss_bwd(cache, np.full(B, 1.0 / B))
    L = _t(dev, logits)
    out_l = torch.empty(B, dtype=torch.float32, device=dev)
    out_dt = torch.empty(B, dtype=torch.float32, device=dev)
    dlog = torch.empty((B, S), dtype=torch.float32, device=dev)
    ops.loss_mw(L, _t(dev, t), _t(dev, mask.astype(np.uint8)), out_l, dlog, out_dt, 1.0 / B, kind=kind)
    np.testing.assert_allclose(out_l.cpu().numpy(), bl, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dlog.cpu().numpy(), dl, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(out_dt.cpu().numpy(), dt, rtol=RTOL, atol=1e-7)
    # in place + no mask + row weights
    rw = rng.random(B).astype(np.float32)
    bl2, cache2 = e.compute_loss(logits.astype(np.float64), t.astype(np.float64), kind,
                                 np.ones((B, S), bool))
    dl2, dt2 = e.compute_loss_bwd(cache2, rw.astype(np.float64) * 0.3)
    ops.loss_mw(L, _t(dev, t), None, out_l, L, out_dt, 0.3, row_w=_t(dev, rw), kind=kind)
    np.testing.assert_allclose(L.cpu().numpy(), dl2, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(out_dt.cpu().numpy(), dt2, rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("B,S", [(64, 1024), (33, 3100)])
def test_loss_mce_large_logits_saturate(dev, B, S):
    """Round-5 advisor (medium): logits that lead the target score by more than 88.7 overflowed the fused 'mce'
    family.  The build-defined loss saturates its exponent at 64 on EVERY path (csrc/common.h kMceSat, the oracle's
    MCE_SAT): the materialising kernels (wave-per-row and workgroup-per-row) against the oracle on logits up to
    +-300 -- finite everywhere, rows far beyond the cap included."""
    from arx import ops
    import torch
    rng = np.random.default_rng(7 * B + S)
    logits = (rng.standard_normal((B, S)) * 60.0).astype(np.float32)
    t = (rng.standard_normal((B,)) * 20.0).astype(np.float32)
    logits[0] = -150.0                                        # a row far BELOW its target score
    t[0] = 100.0
    mask = _mask(rng, B, S)
    assert ((logits - t[:, None]) * mask > 88.7).any()
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.dt = np.dtype(np.float64)
    bl, cache = e.compute_loss(logits.astype(np.float64), t.astype(np.float64), 'mce', mask)
    dl, dt = e.compute_loss_bwd(cache, np.full(B, 1.0 / B))
    assert np.isfinite(bl).all() and np.isfinite(dl).all()
    out_l = torch.empty(B, dtype=torch.float32, device=dev)
    out_dt = torch.empty(B, dtype=torch.float32, device=dev)
    dlog = torch.empty((B, S), dtype=torch.float32, device=dev)
    ops.loss_mw(_t(dev, logits), _t(dev, t), _t(dev, mask.astype(np.uint8)), out_l, dlog, out_dt, 1.0 / B, kind='mce')
    assert bool(torch.isfinite(out_l).all()) and bool(torch.isfinite(dlog).all()) and bool(torch.isfinite(out_dt).all())
    np.testing.assert_allclose(out_l.cpu().numpy(), bl, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dlog.cpu().numpy(), dl, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(out_dt.cpu().numpy(), dt, rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("B,V", [(64, 3100), (7, 50)])
def test_loss_warp_and_ce(dev, B, V):
    from arx import ops
    import torch
    rng = np.random.default_rng(B * V)
    logits = rng.standard_normal((B, V)).astype(np.float32)
    tgt = rng.integers(0, V, size=B).astype(np.int32)
    mask = _mask(rng, B, V)
    mask[np.arange(B)[::2], tgt[::2]] = False     # half the targets masked (training case)
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.dt = np.dtype(np.float64)
    for loss in ('warp', 'ce'):
        bl, cache = e.compute_loss(logits.astype(np.float64), tgt, loss, mask)
        dl, _ = e.compute_loss_bwd(cache, np.full(B, 1.0 / B))
        out_l = torch.empty(B, dtype=torch.float32, device=dev)
        dlog = torch.empty((B, V), dtype=torch.float32, device=dev)
        if loss == 'warp':
            ops.loss_warp(_t(dev, logits), _t(dev, tgt), _t(dev, mask.astype(np.uint8)), out_l,
                          dlog, 1.0 / B)
        else:
            ops.loss_ce(_t(dev, logits), _t(dev, tgt), out_l, dlog, 1.0 / B)
        np.testing.assert_allclose(out_l.cpu().numpy(), bl, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(dlog.cpu().numpy(), dl, rtol=RTOL, atol=1e-7)
    mr, tr = e.warp_eval(logits.astype(np.float64), tgt, mask)
    o_mr = torch.empty(B, dtype=torch.float32, device=dev)
    o_tr = torch.empty(B, dtype=torch.int32, device=dev)
    ops.loss_warp_eval(_t(dev, logits), _t(dev, tgt), _t(dev, mask.astype(np.uint8)), o_mr, o_tr)
    np.testing.assert_allclose(o_mr.cpu().numpy(), mr, rtol=RTOL, atol=1e-4)
    np.testing.assert_array_equal(o_tr.cpu().numpy(), tr)


def test_pos_mask_bit_exact(dev):
    from arx import ops
    import torch
    rng = np.random.default_rng(5)
    Nu, Ni, S, B = 300, 2000, 128, 64
    pos = {u: list(rng.choice(Ni, size=rng.integers(0, 30), replace=False)) for u in range(0, Nu, 2)}
    ptr = np.zeros(Nu + 1, dtype=np.int32)
    items = []
    for u in range(Nu):
        items.extend(pos.get(u, []))
        ptr[u + 1] = len(items)
    items = np.asarray(items, dtype=np.int32)
    pool = rng.choice(Ni, size=S, replace=False).astype(np.int32)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    users = rng.integers(0, Nu, size=B).astype(np.int32)
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.n_sampled, e.logit_size, e.item_ind2logit_ind = S, Ni, None
    e.pos_item_set, e.pos_item_set_eval = pos, pos
    ref = e.mask(list(users), 'mw', id2idx)
    m2s = torch.full((Ni,), -1, dtype=torch.int32, device=dev)
    ops.slot_map_set(m2s, _t(dev, pool))
    mask = torch.ones((B, S), dtype=torch.uint8, device=dev)
    ops.pos_mask_scatter(_t(dev, users), _t(dev, ptr), _t(dev, items), m2s, mask, 0)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), ref)
    ops.pos_mask_scatter(_t(dev, users), _t(dev, ptr), _t(dev, items), m2s, mask, 1)
    assert bool(mask.all().item())
    ops.slot_map_set(m2s, _t(dev, pool), clear=True)
    assert bool((m2s == -1).all().item())


def _ref_sparse_adagrad(E, acc, bias, bacc, keys, src, coef, G, Gb, lr, gs=1.0):
    E, acc, bias, bacc = [x.astype(np.float64).copy() for x in (E, acc, bias, bacc)]
    g = np.zeros_like(E)
    gb = np.zeros_like(bias)
    for k, s, c in zip(keys, src, coef):
        if k == 0x7FFFFFFF:
            continue
        g[k] += c * G[s]
        gb[k] += c * Gb[s]
    g *= gs
    gb *= gs
    touched = np.zeros(E.shape[0], bool)
    touched[keys[keys != 0x7FFFFFFF]] = True
    acc[touched] += g[touched] ** 2
    E[touched] -= lr * g[touched] / np.sqrt(acc[touched])
    bacc[touched] += gb[touched] ** 2
    bias[touched] -= lr * gb[touched] / np.sqrt(bacc[touched])
    return E, acc, bias, bacc


def _sparse_adagrad_sum_bound(E, acc0, keys, src, coef, G, lr, gs=1.0):
    """First-order bound of what f32 summation ORDER may change in a sparse Adagrad update (the merged row is a
    sum of n_k terms c*G[s]; any f32 order is within n_k * 2^-24 * sum|c*G[s]| of the exact sum -- Higham,
    Accuracy and Stability, (4.4)), carried through acc += g^2 and E -= lr * g / sqrt(acc).  Returns
    (bound on acc, bound on E), both [rows, d]: zero for rows whose run is one entry."""
    live = keys != 0x7FFFFFFF
    k, s_, c = keys[live].astype(np.int64), src[live], coef[live].astype(np.float64)
    rows, d = E.shape
    absg = np.zeros((rows, d))
    g = np.zeros((rows, d))
    np.add.at(absg, k, np.abs(c[:, None] * G[s_].astype(np.float64)))
    np.add.at(g, k, c[:, None] * G[s_].astype(np.float64))
    n_k = np.bincount(k, minlength=rows).astype(np.float64)[:, None]
    gam = np.where(n_k > 1, n_k, 0.0) * 2.0 ** -24 * absg * gs
    g = np.abs(g) * gs
    acc = acc0.astype(np.float64) + g * g
    b_acc = 2 * g * gam + gam * gam
    b_E = lr * (gam / np.sqrt(acc) + g * b_acc / (2 * acc ** 1.5))
    return b_acc, b_E


@pytest.mark.parametrize("d,n,Vf,hot", [(128, 5000, 300, 0), (128, 20000, 5000, 3000),
                                        (32, 777, 50, 400), (64, 64, 1000, 0), (128, 130, 2, 0),
                                        # n > 8192: device-wide LSD radix sort (2 / 3 passes,
                                        # 256 sort blocks with multi-round slices)
                                        (4, 70000, 2100000, 500), (32, 600000, 40000, 20000),
                                        (16, 9000, 1, 0)])
def test_sparse_adagrad(dev, d, n, Vf, hot):
    """Duplicates summed first, one update per touched row; long runs (hot keys,
    > kPiece duplicates) exercise the two-pass path."""
    from arx import ops
    import torch
    rng = np.random.default_rng(n + d)
    E = rng.standard_normal((Vf, d)).astype(np.float32)
    acc = np.full((Vf, d), 0.1, dtype=np.float32) + rng.random((Vf, d)).astype(np.float32)
    bias = rng.standard_normal((Vf,)).astype(np.float32)
    bacc = np.full((Vf,), 0.1, dtype=np.float32)
    m = 97
    G = rng.standard_normal((m, d)).astype(np.float32)
    Gb = rng.standard_normal((m,)).astype(np.float32)
    keys = rng.integers(0, Vf, size=n).astype(np.int32)
    if hot:
        keys[rng.choice(n, size=hot, replace=False)] = 1 % Vf
        keys[rng.choice(n, size=hot // 4, replace=False)] = Vf - 1
    keys[rng.choice(n, size=n // 10, replace=False)] = 0x7FFFFFFF
    src = rng.integers(0, m, size=n).astype(np.int32)
    coef = rng.random(n).astype(np.float32)
    lr, gs = 0.3, 0.7
    rE, racc, rb, rbacc = _ref_sparse_adagrad(E, acc, bias, bacc, keys, src, coef, G, Gb, lr, gs)
    tE, tacc, tb, tbacc = _t(dev, E), _t(dev, acc), _t(dev, bias), _t(dev, bacc)
    ws = ops.Workspace(dev)
    lr_dev = torch.tensor([lr], dtype=torch.float32, device=dev)
    gs_dev = torch.tensor([gs], dtype=torch.float32, device=dev)
    ops.sparse_adagrad(tE, tacc, tb, tbacc, _t(dev, keys), _t(dev, src), _t(dev, coef), _t(dev, G),
                       _t(dev, Gb), lr_dev, ws, gscale_dev=gs_dev)
    torch.cuda.synchronize()
    np.testing.assert_allclose(tacc.cpu().numpy(), racc, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tE.cpu().numpy(), rE, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(tbacc.cpu().numpy(), rbacc, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tb.cpu().numpy(), rb, rtol=1e-4, atol=2e-5)
    # determinism: the same call on the same inputs is bit-identical
    tE2, tacc2 = _t(dev, E), _t(dev, acc)
    tb2, tbacc2 = _t(dev, bias), _t(dev, bacc)
    ops.sparse_adagrad(tE2, tacc2, tb2, tbacc2, _t(dev, keys), _t(dev, src), _t(dev, coef),
                       _t(dev, G), _t(dev, Gb), lr_dev, ws, gscale_dev=gs_dev)
    assert torch.equal(tE, tE2) and torch.equal(tacc, tacc2)


def test_one_launch_reductions_are_reentrant_across_streams(dev):
    """arx_sq_norm_clip_multi and arx_max_argmax on TWO streams at once: block partials and the
    arrival ticket live in the caller's reduce scratch (arx_reduce_scratch_bytes), one per stream here, so the
    launches cannot meet in library-owned memory.  Many rounds of overlapping launches with different data;
    every result is checked, the norms bit for bit against a quiet single-stream run."""
    from arx import ops
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n = 1 << 20
    xs = [torch.randn(n, device=dev, generator=g) * (1.0 + k) for k in range(2)]
    ys = [torch.randn(4, n // 4, device=dev, generator=g) for _ in range(2)]
    for k in range(2):
        ys[k][k, 1000 + k] = 50.0 + k
    sc = [ops.new_reduce_scratch(dev) for _ in range(2)]
    quiet = []
    for k in range(2):
        sq, coef, gn = (torch.zeros(1, device=dev) for _ in range(3))
        ops.sq_norm_clip_multi([(xs[k], 1, None, None)], sq, 5.0, coef, gn, scratch=sc[k])
        quiet.append((float(sq.item()), float(coef.item())))
        assert quiet[k][0] == pytest.approx(float((xs[k].double() ** 2).sum().item()), rel=1e-5)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [[(torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev),
              torch.zeros(1, device=dev), torch.zeros(2, dtype=torch.int32, device=dev)) for _ in range(40)]
            for _ in range(2)]
    torch.cuda.synchronize()
    for r in range(40):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                sq, coef, gn, best, bidx = outs[k][r]
                ops.sq_norm_clip_multi([(xs[k], 1, None, None)], sq, 5.0, coef, gn, scratch=sc[k])
                ops.max_argmax(ys[k], 0, True, best, bidx, scratch=sc[k])
    torch.cuda.synchronize()
    for k in range(2):
        for sq, coef, gn, best, bidx in outs[k]:
            assert (float(sq.item()), float(coef.item())) == quiet[k]
            assert float(best.item()) == 50.0 + k and bidx.tolist() == [k, 1000 + k]
        assert int(sc[k].view(torch.int32)[0].item()) == 0             # the ticket is back at zero
    with pytest.raises(Exception):
        from arx._lib import call
        call("arx_max_argmax", ys[0].data_ptr(), 4, n // 4, n // 4, 0, 1, outs[0][0][3].data_ptr(),
             outs[0][0][4].data_ptr(), None, None)                     # no scratch: an error, not a global


def test_dense_adagrad_norm_clip(dev):
    from arx import ops
    import torch
    rng = np.random.default_rng(1)
    n = 100003
    w = rng.standard_normal(n).astype(np.float32)
    acc = np.full(n, 0.1, np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    tw, ta = _t(dev, w), _t(dev, acc)
    lr = torch.tensor([0.5], dtype=torch.float32, device=dev)
    sq = torch.zeros(1, dtype=torch.float32, device=dev)
    ops.sq_norm_accum(_t(dev, g), sq)
    np.testing.assert_allclose(sq.item(), (g.astype(np.float64) ** 2).sum(), rtol=1e-5)
    coef = torch.empty(1, dtype=torch.float32, device=dev)
    gn = torch.empty(1, dtype=torch.float32, device=dev)
    ops.clip_coef(sq, 5.0, coef, gn)
    nrm = np.sqrt((g.astype(np.float64) ** 2).sum())
    np.testing.assert_allclose(coef.item(), 5.0 / max(nrm, 5.0), rtol=1e-5)
    np.testing.assert_allclose(gn.item(), nrm, rtol=1e-5)
    ops.adagrad_dense(tw, ta, _t(dev, g), lr, gscale_dev=coef)
    gg = g.astype(np.float64) * (5.0 / max(nrm, 5.0))
    racc = acc + gg * gg
    np.testing.assert_allclose(ta.cpu().numpy(), racc, rtol=RTOL)
    np.testing.assert_allclose(tw.cpu().numpy(), w - 0.5 * gg / np.sqrt(racc), rtol=RTOL, atol=ATOL)


def test_topk_matches_tf_order(dev):
    from arx import ops
    import torch
    rng = np.random.default_rng(2)
    B, V, k = 9, 3100, 100
    logits = rng.standard_normal((B, V)).astype(np.float32)
    logits[:, 100:110] = logits[:, 99:100]       # ties -> lower index first
    ref = np.argsort(-logits, axis=1, kind='stable')[:, :k]
    vals = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    ops.topk(_t(dev, logits), k, vals, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(vals.cpu().numpy(), np.take_along_axis(logits, ref, 1))


@pytest.mark.parametrize("B,V,k", [(5, 100000, 100), (3, 4096, 1024), (7, 2048, 5), (2, 1000003, 30)])
def test_topk_radix_select(dev, B, V, k):
    """Radix-select top-k (topk.hip): index-exact against a stable argsort, with heavy ties
    (quantised values, a constant row, negative values, +-0)."""
    from arx import ops
    import torch
    rng = np.random.default_rng(V + k)
    logits = rng.standard_normal((B, V)).astype(np.float32)
    logits[0] = np.round(logits[0] * 4) / 4               # many exact ties
    if B > 1:
        logits[1] = -1.5                                   # all equal: lowest k columns win
    if B > 2:
        logits[2, ::3] = 0.0
        logits[2, 1::3] = -0.0
    ref = np.argsort(-logits, axis=1, kind='stable')[:, :k]
    vals = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    ops.topk(_t(dev, logits), k, vals, idx)
    got = idx.cpu().numpy()
    gv = vals.cpu().numpy()
    # numpy's stable argsort treats -0.0 == 0.0; the kernel orders +0 above -0: compare values
    # exactly and indices wherever the value is not a signed zero
    np.testing.assert_array_equal(np.abs(gv), np.abs(np.take_along_axis(logits, ref, 1)))
    nz = np.take_along_axis(logits, ref, 1) != 0
    np.testing.assert_array_equal(got[nz], ref[nz])
    for r in range(B):
        assert len(np.unique(got[r])) == k


def test_topk_streaming_chunks_equal_full(dev):
    """Chunked top-k + merge (the streaming recommend path) == top-k of the full row."""
    from arx import ops
    import torch
    rng = np.random.default_rng(9)
    B, V, k, Vc = 6, 50000, 64, 8192
    logits = np.round(rng.standard_normal((B, V)) * 8).astype(np.float32) / 8     # ties across chunks
    ref = np.argsort(-logits, axis=1, kind='stable')[:, :k]
    tl = _t(dev, logits)
    run_v = torch.empty((B, k), dtype=torch.float32, device=dev)
    run_i = torch.empty((B, k), dtype=torch.int32, device=dev)
    tmp_v, tmp_i = torch.empty_like(run_v), torch.empty_like(run_i)
    out_v, out_i = torch.empty_like(run_v), torch.empty_like(run_i)
    first = True
    for c0 in range(0, V, Vc):
        c1 = min(V, c0 + Vc)
        kc = min(k, c1 - c0)
        cv = tmp_v[:, :kc] if kc == k else torch.empty((B, kc), dtype=torch.float32, device=dev)
        ci = tmp_i[:, :kc] if kc == k else torch.empty((B, kc), dtype=torch.int32, device=dev)
        ops.topk_chunk(tl[:, c0:c1], kc, c0, cv, ci)
        if first:
            run_v.copy_(cv)
            run_i.copy_(ci)
            first = False
        else:
            ops.topk_merge(run_v, run_i, cv, ci, k, out_v, out_i)
            run_v, out_v = out_v, run_v
            run_i, out_i = out_i, run_i
    np.testing.assert_array_equal(run_i.cpu().numpy(), ref)
    np.testing.assert_array_equal(run_v.cpu().numpy(), np.take_along_axis(logits, ref, 1))


def test_small_utils(dev):
    from arx import ops
    import torch
    rng = np.random.default_rng(3)
    x = rng.standard_normal((50, 64)).astype(np.float32)
    y = rng.standard_normal((200, 64)).astype(np.float32)
    ty = _t(dev, y)
    ops.add_rows_bcast(0.5, _t(dev, x), 0.25, ty)
    np.testing.assert_allclose(ty.cpu().numpy(), 0.5 * np.tile(x, (4, 1)) + 0.25 * y, rtol=1e-6)
    rs = torch.empty(200, dtype=torch.float32, device=dev)
    ops.row_sum(_t(dev, y), rs)
    np.testing.assert_allclose(rs.cpu().numpy(), y.sum(1), rtol=1e-5, atol=1e-5)
    cs = torch.empty(64, dtype=torch.float32, device=dev)
    ops.col_sum(_t(dev, y), cs, ops.Workspace(dev))
    np.testing.assert_allclose(cs.cpu().numpy(), y.sum(0), rtol=1e-5, atol=1e-4)
    s = torch.empty(1, dtype=torch.float32, device=dev)
    ops.sum_scaled(_t(dev, y), 0.125, s)
    np.testing.assert_allclose(s.item(), y.sum() * 0.125, rtol=1e-4, atol=1e-4)
    w = np.array([[1, 1, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32)  # [L=4,B=3]
    o = torch.empty((4, 3), dtype=torch.float32, device=dev)
    ops.seq_weights(_t(dev, w), 4, 3, o)
    np.testing.assert_allclose(o.cpu().numpy(), w / (w.sum(0) + 1e-12), rtol=1e-6)
    # dropout: keep_prob=1 is the identity; keep_prob<1 keeps ~p and rescales
    xd = _t(dev, y)
    yd = torch.empty_like(xd)
    km = torch.empty(xd.numel(), dtype=torch.uint8, device=dev)
    ops.dropout_fwd(xd, 1.0, 7, yd, km)
    assert torch.equal(xd, yd)
    ops.dropout_fwd(xd, 0.5, 7, yd, km)
    frac = km.float().mean().item()
    assert 0.45 < frac < 0.55
    kept = km.view_as(xd).bool()
    np.testing.assert_allclose(yd[kept].cpu().numpy(), (xd[kept] * 2).cpu().numpy(), rtol=1e-6)
    assert float(yd[~kept].abs().sum().item()) == 0.0


@pytest.mark.parametrize("L,B,din,h", [(3, 5, 64, 64), (50, 64, 64, 64), (7, 33, 128, 128),
                                       (4, 6, 20, 24), (5, 16, 32, 64)])
def test_lstm_fwd_bwd(dev, L, B, din, h):
    from arx import ops
    import torch
    from oracle import ref_lstm
    rng = np.random.default_rng(L * B + h)
    x = (rng.standard_normal((L, B, din)) * 0.5).astype(np.float32)
    W = (rng.standard_normal((din + h, 4 * h)) * 0.2).astype(np.float32)
    b = (rng.standard_normal((4 * h,)) * 0.1).astype(np.float32)
    dhs = rng.standard_normal((L, B, h)).astype(np.float32)
    r_hs, r_cs, r_g = ref_lstm.lstm_fwd(x.astype(np.float64), W.astype(np.float64),
                                        b.astype(np.float64), 1.0)
    r_dz, r_dx, r_dW, r_db = ref_lstm.lstm_bwd(x.astype(np.float64), W.astype(np.float64), r_hs,
                                               r_cs, r_g, dhs.astype(np.float64))
    hs = torch.empty((L, B, h), dtype=torch.float32, device=dev)
    cs = torch.empty_like(hs)
    gates = torch.empty((L, B, 4 * h), dtype=torch.float32, device=dev)
    tW = _t(dev, W)
    ops.lstm_fwd(_t(dev, x), tW, _t(dev, b), L, B, din, h, 1.0, hs, cs, gates)
    np.testing.assert_allclose(hs.cpu().numpy(), r_hs, rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), r_cs, rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(gates.cpu().numpy(), r_g, rtol=RTOL, atol=2e-5)
    dz = torch.empty((L, B, 4 * h), dtype=torch.float32, device=dev)
    wxt = torch.full((4 * h, din), 7.0, dtype=torch.float32, device=dev)
    ops.lstm_bwd(tW, hs, cs, gates, _t(dev, dhs), L, B, din, h, dz, wxt=wxt)
    np.testing.assert_allclose(dz.cpu().numpy(), r_dz, rtol=1e-4, atol=5e-5)
    assert np.array_equal(wxt.cpu().numpy(), W[:din].T)         # W_x^T rides along (a copy: exact)
    # dx / dW / db are GEMMs + a column sum over dz (what the model issues)
    ws = ops.Workspace(dev)
    dz2 = dz.view(L * B, 4 * h)
    dx = torch.empty((L * B, din), dtype=torch.float32, device=dev)
    ops.gemm(dz2, tW[:din], dx, ws, transB=True)
    np.testing.assert_allclose(dx.cpu().numpy(), r_dx.reshape(L * B, din), rtol=1e-4, atol=1e-4)
    dW = torch.zeros((din + h, 4 * h), dtype=torch.float32, device=dev)
    ops.gemm(_t(dev, x).view(L * B, din), dz2, dW[:din], ws, transA=True)
    if L > 1:
        ops.gemm(hs.view(L * B, h)[:(L - 1) * B], dz2[B:], dW[din:], ws, transA=True)
    np.testing.assert_allclose(dW.cpu().numpy(), r_dW, rtol=1e-4, atol=2e-4)
    db = torch.empty(4 * h, dtype=torch.float32, device=dev)
    ops.col_sum(dz2, db, ws)
    np.testing.assert_allclose(db.cpu().numpy(), r_db, rtol=1e-4, atol=2e-4)
    if L > 1 and ops.gemm_tn_pair_supported(4 * h, din, h, L * B):     # the model's one-pass form of dW, db
        dW2 = torch.empty_like(dW)
        db2 = torch.empty_like(db)
        ops.gemm_tn_pair(dz2, _t(dev, x).view(L * B, din), hs.view(L * B, h), B, dW2, ws, a_rowsum=db2)
        np.testing.assert_allclose(dW2.cpu().numpy(), r_dW, rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(db2.cpu().numpy(), r_db, rtol=1e-4, atol=2e-4)


def test_loss_pos_variants_equal_mask_array(dev):
    """mw / warp with the mask derived in-kernel from the positives CSR must be
    bit-identical to the mask-array form fed by arx_pos_mask_scatter."""
    from arx import ops
    import torch
    rng = np.random.default_rng(8)
    Nu, Ni, B = 200, 5000, 48
    for W, kind in ((1024, 'mw'), (5000, 'warp'), (70000, 'warp')):
        Ni2 = max(Ni, W)
        ptr = np.zeros(Nu + 1, dtype=np.int32)
        cnt = rng.integers(0, 40, size=Nu)
        ptr[1:] = np.cumsum(cnt)
        items = rng.integers(0, Ni2, size=int(ptr[-1])).astype(np.int32)
        i2s = np.full(Ni2, -1, dtype=np.int32)
        sel = rng.choice(Ni2, size=min(W, Ni2), replace=False)
        i2s[sel] = rng.permutation(len(sel)).astype(np.int32)
        users = rng.integers(0, Nu, size=B).astype(np.int32)
        logits = rng.standard_normal((B, W)).astype(np.float32)
        t = rng.standard_normal((B,)).astype(np.float32)
        tgt = rng.integers(0, W, size=B).astype(np.int32)
        mask = torch.ones((B, W), dtype=torch.uint8, device=dev)
        d = [_t(dev, x) for x in (users, ptr, items, i2s)]
        ops.pos_mask_scatter(d[0], d[1], d[2], d[3], mask, 0)
        L = _t(dev, logits)
        bl_a = torch.empty(B, dtype=torch.float32, device=dev); bl_b = torch.empty_like(bl_a)
        dl_a = torch.empty_like(L); dl_b = torch.empty_like(L)
        dt_a = torch.empty_like(bl_a); dt_b = torch.empty_like(bl_a)
        if kind == 'mw':
            ops.loss_mw(L, _t(dev, t), mask, bl_a, dl_a, dt_a, 0.1)
            ops.loss_mw_pos(L, _t(dev, t), d[0], d[1], d[2], d[3], bl_b, dl_b, dt_b, 0.1)
            assert torch.equal(dt_a, dt_b)
        else:
            ops.loss_warp(L, _t(dev, tgt), mask, bl_a, dl_a, 0.1)
            ops.loss_warp_pos(L, _t(dev, tgt), d[0], d[1], d[2], d[3], bl_b, dl_b, 0.1)
        assert torch.equal(bl_a, bl_b) and torch.equal(dl_a, dl_b)
        assert int((mask == 0).sum().item()) > 0


@pytest.mark.parametrize("tA,M,N,K", [(True, 1024, 128, 4096), (True, 300, 32, 64),
                                      (False, 200, 64, 1000)])
def test_gemm_rowsum(dev, tA, M, N, K):
    from arx import ops
    import torch
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    opA = A.T if tA else A
    C = torch.empty((M, N), dtype=torch.float32, device=dev)
    rs = torch.full((M,), 7.0, dtype=torch.float32, device=dev)
    ops.gemm(_t(dev, A), _t(dev, Bm), C, ops.Workspace(dev), transA=tA, a_rowsum=rs)
    np.testing.assert_allclose(rs.cpu().numpy(), opA.astype(np.float64).sum(1), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(C.cpu().numpy(), opA.astype(np.float64) @ Bm, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,N1,N2,K,shift", [(256, 64, 64, 51200, 1024), (256, 64, 64, 640, 128),
                                             (128, 32, 64, 4096, 0), (512, 100, 28, 320, 320),
                                             (64, 8, 40, 96, 5)])
def test_gemm_tn_pair(dev, M, N1, N2, K, shift):
    """The LSTM cell's dW and db in one pass over dz (arx_gemm_f32_tn_pair): (A^T . [B1 | B2 moved down by
    `shift` rows])^T against a float64 product, the bound scaled by |A|^T |B| (f32 products, f32 sums)."""
    from arx import ops
    import torch
    rng = np.random.default_rng(M + N1 + K + shift)
    A = rng.standard_normal((K, M)).astype(np.float32)
    B1 = rng.standard_normal((K, N1)).astype(np.float32)
    B2 = rng.standard_normal((K, N2)).astype(np.float32)       # rows [0, K - shift) are read
    assert ops.gemm_tn_pair_supported(M, N1, N2, K)
    Ct = torch.full((N1 + N2, M), 7.0, dtype=torch.float32, device=dev)
    rs = torch.full((M,), 7.0, dtype=torch.float32, device=dev)
    ops.gemm_tn_pair(_t(dev, A), _t(dev, B1), _t(dev, B2), shift, Ct, ops.Workspace(dev), a_rowsum=rs)
    Bcat = np.zeros((K, N1 + N2))
    Bcat[:, :N1] = B1
    Bcat[shift:, N1:] = B2[:K - shift]
    ref = (A.astype(np.float64).T @ Bcat).T
    scale = (np.abs(A).astype(np.float64).T @ np.abs(Bcat)).T
    err = np.abs(Ct.cpu().numpy() - ref)
    assert np.all(err <= 2e-6 * scale + 1e-5), float((err / (scale + 1e-9)).max())
    np.testing.assert_allclose(rs.cpu().numpy(), A.astype(np.float64).sum(0), rtol=1e-4, atol=2e-3)
    # deterministic: a second launch gives the same bits
    Ct2 = torch.empty_like(Ct)
    ops.gemm_tn_pair(_t(dev, A), _t(dev, B1), _t(dev, B2), shift, Ct2, ops.Workspace(dev))
    assert torch.equal(Ct, Ct2)


@pytest.mark.parametrize("d,Vf,ns", [(128, 5000, (4096, 1024)), (32, 40, (300, 17, 64)), (64, 100000, (64,))])
def test_sparse_adagrad_cat_fast_path(dev, d, Vf, ns):
    """Sort-free one-hot path == reference semantics (duplicates summed, one update per
    row), bit-deterministic, leaves its aux arrays clean."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + Vf)
    E = rng.standard_normal((Vf, d)).astype(np.float32)
    acc = (0.1 + rng.random((Vf, d))).astype(np.float32)
    bias = rng.standard_normal((Vf,)).astype(np.float32)
    bacc = np.full((Vf,), 0.1, dtype=np.float32)
    N = 3 * Vf
    sites, keys, src, coef = [], [], [], []
    base = 0
    m = sum(ns)
    G = rng.standard_normal((m, d)).astype(np.float32)
    Gb = rng.standard_normal((m,)).astype(np.float32)
    for k, n in enumerate(ns):
        cmap = rng.integers(0, Vf, size=N).astype(np.int32) if k % 2 == 0 else None
        hi = N if cmap is not None else Vf
        ids = rng.integers(0, hi, size=n).astype(np.int32)
        if n > 40:
            ids[rng.choice(n, size=n // 3, replace=False)] = ids[0]       # a hot row
        c = float(rng.random() + 0.5)
        sites.append((_t(dev, cmap) if cmap is not None else None, None, _t(dev, ids), base, c))
        keys.append(cmap[ids] if cmap is not None else ids)
        src.append(base + np.arange(n))
        coef.append(np.full(n, c, dtype=np.float32))
        base += n
    keys, src, coef = np.concatenate(keys), np.concatenate(src).astype(np.int32), np.concatenate(coef)
    rE, racc, rb, rbacc = _ref_sparse_adagrad(E, acc, bias, bacc, keys, src, coef, G, Gb, 0.3, 0.7)
    lr = torch.tensor([0.3], dtype=torch.float32, device=dev)
    gs = torch.tensor([0.7], dtype=torch.float32, device=dev)
    first = torch.full((Vf,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
    cnt = torch.zeros((Vf,), dtype=torch.int32, device=dev)
    args = ops.CatSiteArgs(sites)
    hot = torch.zeros(m // 16 + 4, dtype=torch.int32, device=dev)
    outs = []
    ws = ops.Workspace(dev)
    for rep in range(4):                     # mode 1 (atomic election) twice, mode 0 (fused sort) twice
        mode = 1 if rep < 2 else 0
        tE, tacc, tb, tbacc = _t(dev, E), _t(dev, acc), _t(dev, bias), _t(dev, bacc)
        kb = torch.empty(m, dtype=torch.int32, device=dev)
        sb = torch.empty(m, dtype=torch.int32, device=dev)
        cb = torch.empty(m, dtype=torch.float32, device=dev)
        ops.sparse_adagrad_cat(tE, tacc, tb, tbacc, args, _t(dev, G), _t(dev, Gb), lr, first, cnt,
                               hot, kb, sb, cb, ws, gscale_dev=gs, mode=mode)
        torch.cuda.synchronize()
        outs.append((tE, tacc, tb))
        assert int((first != 2 ** 31 - 1).sum().item()) == 0 and int(cnt.abs().sum().item()) == 0
        assert int(hot[:2].abs().sum().item()) == 0
    for o in (outs[0], outs[2]):
        np.testing.assert_allclose(o[1].cpu().numpy(), racc, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(o[0].cpu().numpy(), rE, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(o[2].cpu().numpy(), rb, rtol=1e-4, atol=2e-5)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[2][0], outs[3][0]) and torch.equal(outs[2][1], outs[3][1])


@pytest.mark.parametrize("n,Vf,dist", [(60000, 3000, 'zipf'), (20000, 50, 'zipf'),
                                       (100000, 1, 'zipf'),          # ONE run of 80 k entries: 313 work items
                                       (50001, 200000, 'uniform'),   # almost every run is one entry
                                       (16385, 7, 'blocks')])        # runs that end exactly on tile borders
def test_sparse_adagrad_ticket_large_n(dev, n, Vf, dist):
    """Device radix-sort path (n > 8192) with the run records + run-centric apply: Zipf-heavy keys
    (runs of thousands of duplicates spanning many work items), one giant run, all-distinct
    keys, runs cut at the extraction tile size == reference, deterministic."""
    from arx import ops
    import torch
    rng = np.random.default_rng(n)
    d = 128
    E = rng.standard_normal((Vf, d)).astype(np.float32)
    acc = (0.1 + rng.random((Vf, d))).astype(np.float32)
    bias = rng.standard_normal((Vf,)).astype(np.float32)
    bacc = np.full((Vf,), 0.1, dtype=np.float32)
    m = 400
    G = rng.standard_normal((m, d)).astype(np.float32)
    Gb = rng.standard_normal((m,)).astype(np.float32)
    if dist == 'zipf':
        p = 1.0 / np.arange(1, Vf + 1)
        keys = rng.choice(Vf, size=n, p=p / p.sum()).astype(np.int32)
        keys[rng.choice(n, size=n // 5, replace=False)] = 0x7FFFFFFF
    elif dist == 'uniform':
        keys = rng.integers(0, Vf, size=n).astype(np.int32)
    else:   # sorted order = runs of exactly 2048 (the extraction tile), then a tail of one
        keys = rng.permutation((np.arange(n) // 2048).astype(np.int32) % Vf)
        keys[keys == 0] = np.where(rng.random((keys == 0).sum()) < 0.5, 0, Vf - 1)
    src = rng.integers(0, m, size=n).astype(np.int32)
    coef = rng.random(n).astype(np.float32)
    rE, racc, rb, rbacc = _ref_sparse_adagrad(E, acc, bias, bacc, keys, src, coef, G, Gb, 0.3, 1.0)
    lr = torch.tensor([0.3], dtype=torch.float32, device=dev)
    cnt = torch.zeros((Vf,), dtype=torch.int32, device=dev)
    ws = ops.Workspace(dev)
    outs = []
    for rep in range(2):
        tE, tacc, tb, tbacc = _t(dev, E), _t(dev, acc), _t(dev, bias), _t(dev, bacc)
        ops.sparse_adagrad(tE, tacc, tb, tbacc, _t(dev, keys), _t(dev, src), _t(dev, coef), _t(dev, G),
                           _t(dev, Gb), lr, ws, aux_cnt=cnt)
        torch.cuda.synchronize()
        assert int(cnt.abs().sum().item()) == 0
        outs.append((tE, tacc, tb))
    # the tolerance of every other K7 test (rtol 1e-4, atol 1e-5 / 2e-5) plus a DERIVED term for the summation
    # order of long runs -- up to 80 k entries of mixed sign here, where cancellation, not the kernel, sets the
    # error: n_k * 2^-24 * sum|c*g| carried through the Adagrad formula (_sparse_adagrad_sum_bound)
    b_acc, b_E = _sparse_adagrad_sum_bound(E, acc, keys, src, coef, G, 0.3)
    bb_acc, bb_E = _sparse_adagrad_sum_bound(bias[:, None], bacc[:, None], keys, src, coef, Gb[:, None], 0.3)
    for got, ref, bound, atol in ((outs[0][1], racc, b_acc, 1e-5), (outs[0][0], rE, b_E, 2e-5),
                                  (outs[0][2], rb, bb_E[:, 0], 2e-5)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref)
        lim = 1e-4 * np.abs(ref) + atol + bound
        assert (err <= lim).all(), (float((err - lim).max()), float(bound.max()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B", [900, 60])      # 11.7 k / 780 contribution slots: both sides of the rank-sort limit
def test_sparse_adagrad_cat_multi_with_multihot_segments(dev, B):
    """One fused pass over a one-hot table and a multi-hot token table: the multi-hot lookups
    arrive as arx_csr_expand output (table-local token keys, ARX_KEY_NONE pads) behind the
    one-hot contributions; result == one reference update per table."""
    from arx import ops
    import torch
    rng = np.random.default_rng(5)
    d, V0, V1, n_rows = 64, 3000, 500, 1200
    G = rng.standard_normal((2 * B, d)).astype(np.float32)
    Gb = rng.standard_normal((2 * B,)).astype(np.float32)
    E0 = rng.standard_normal((V0, d)).astype(np.float32)
    A0 = (0.1 + rng.random((V0, d))).astype(np.float32)
    E1 = rng.standard_normal((V1, d)).astype(np.float32)
    A1 = (0.1 + rng.random((V1, d))).astype(np.float32)
    b1 = rng.standard_normal((V1,)).astype(np.float32)
    ba1 = np.full((V1,), 0.1, dtype=np.float32)
    ids0 = rng.integers(0, V0, size=B).astype(np.int32)                    # one-hot site, G rows [0, B)
    vals, starts, lens = _csr(rng, n_rows, V1, 12, zipf=True)              # multi-hot site, G rows [B, 2B)
    bag = rng.integers(0, n_rows, size=B).astype(np.int32)
    tok = rg.batch_slice2(vals, starts[bag], lens[bag])
    seg = rg.batch_segids2(lens[bag])
    coef1 = (0.5 / lens[bag][seg]).astype(np.float32)
    r0 = _ref_sparse_adagrad(E0, A0, np.zeros(V0, np.float32), np.full(V0, 0.1, np.float32),
                             ids0.astype(np.int64), np.arange(B, dtype=np.int32),
                             np.ones(B, np.float32), G, Gb, 0.3, 1.0)
    r1 = _ref_sparse_adagrad(E1, A1, b1, ba1, tok.astype(np.int64), (B + seg).astype(np.int32), coef1,
                             G, Gb, 0.3, 1.0)
    cap = B * 12
    t0 = (_t(dev, E0), _t(dev, A0), None, None, torch.zeros(V0, dtype=torch.int32, device=dev))
    t1 = (_t(dev, E1), _t(dev, A1), _t(dev, b1), _t(dev, ba1), torch.zeros(V1, dtype=torch.int32, device=dev))
    args = ops.MultiCatArgs([t0, t1], [(0, None, _t(dev, ids0), 0, 1.0)], extra=[(1, cap)])
    kb = torch.empty(args.total, dtype=torch.int32, device=dev)
    sb = torch.empty(args.total, dtype=torch.int32, device=dev)
    cb = torch.empty(args.total, dtype=torch.float32, device=dev)
    ws = ops.Workspace(dev)
    off = args.extra_off[0]
    offs = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    tot = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.csr_expand(_t(dev, vals), _t(dev, starts), _t(dev, lens), _t(dev, bag), cap, ws,
                   pad_token=ops.KEY_NONE, pad_seg=0, seg_base=B, coef_scale=0.5, want_coef=True,
                   out=(kb[off:off + cap], sb[off:off + cap], offs, tot, cb[off:off + cap]))
    lr = torch.tensor([0.3], dtype=torch.float32, device=dev)
    ops.sparse_adagrad_cat_multi(args, _t(dev, G), _t(dev, Gb), lr, kb, sb, cb, ws)
    torch.cuda.synchronize()
    np.testing.assert_allclose(t0[0].cpu().numpy(), r0[0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(t0[1].cpu().numpy(), r0[1], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(t1[0].cpu().numpy(), r1[0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(t1[1].cpu().numpy(), r1[1], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(t1[2].cpu().numpy(), r1[2], rtol=1e-4, atol=2e-5)
    assert int(t0[4].abs().sum().item()) == 0 and int(t1[4].abs().sum().item()) == 0


@pytest.mark.parametrize("d,rows,ns", [(128, (5000, 7000), (4096, 1024, 4096)),
                                       (32, (40, 1000, 17), (300, 64, 50, 2000)),
                                       (64, (100000, 300000), (20000, 30000)),
                                       (32, (1 << 20, 70000, 300), (16384, 17000, 5000, 1024)),   # one-launch LDS sort, 2 passes
                                       (16, (3000000, 50), (12000, 9000))])                       # ... 22 key bits: 3 passes
def test_sparse_adagrad_cat_multi(dev, d, rows, ns):
    """Several one-hot tables in one pass (table index in the sort key) == one reference
    update per table; deterministic; counters left clean.  Small n: LDS rank sort + ticket
    apply; up to 20 k keys per table: one-launch LDS radix sort; larger: global radix sort; window apply."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + sum(ns))
    nt = len(rows)
    m = sum(ns)
    G = rng.standard_normal((m, d)).astype(np.float32)
    Gb = rng.standard_normal((m,)).astype(np.float32)
    tabs = []
    for t, V in enumerate(rows):
        E = rng.standard_normal((V, d)).astype(np.float32)
        acc = (0.1 + rng.random((V, d))).astype(np.float32)
        has_bias = (t != 1)
        bias = rng.standard_normal((V,)).astype(np.float32)
        bacc = np.full((V,), 0.1, dtype=np.float32)
        tabs.append(dict(E=E, acc=acc, bias=bias, bacc=bacc, has_bias=has_bias, keys=[], src=[], coef=[]))
    sites = []
    base = 0
    for k, n in enumerate(ns):
        t = k % nt
        V = rows[t]
        N = 3 * V
        cmap = rng.integers(-1, V, size=N).astype(np.int32) if k % 2 == 0 else None   # -1: dropped
        ids = rng.integers(0, N if cmap is not None else V, size=n).astype(np.int32)
        if n > 40:
            ids[rng.choice(n, size=n // 3, replace=False)] = ids[0]       # a hot row
        c = float(rng.random() + 0.5)
        sites.append((t, cmap, ids, base, c))
        key = cmap[ids] if cmap is not None else ids
        tabs[t]['keys'].append(np.where(key < 0, 0x7FFFFFFF, key))
        tabs[t]['src'].append(base + np.arange(n))
        tabs[t]['coef'].append(np.full(n, c, dtype=np.float32))
        base += n
    refs = []
    for tb in tabs:
        keys = np.concatenate(tb['keys']) if tb['keys'] else np.zeros(0, np.int64)
        src = np.concatenate(tb['src']).astype(np.int32) if tb['src'] else np.zeros(0, np.int32)
        coef = np.concatenate(tb['coef']) if tb['coef'] else np.zeros(0, np.float32)
        rE, racc, rb, rbacc = _ref_sparse_adagrad(tb['E'], tb['acc'], tb['bias'], tb['bacc'],
                                                  keys.astype(np.int64), src, coef, G, Gb, 0.3, 0.7)
        if not tb['has_bias']:
            rb, rbacc = tb['bias'], tb['bacc']
        refs.append((rE, racc, rb, rbacc))
    lr = torch.tensor([0.3], dtype=torch.float32, device=dev)
    gs = torch.tensor([0.7], dtype=torch.float32, device=dev)
    outs = []
    ws = ops.Workspace(dev)
    for rep in range(2):
        dtab = []
        for tb in tabs:
            cnt = torch.zeros((tb['E'].shape[0],), dtype=torch.int32, device=dev)
            dtab.append((_t(dev, tb['E']), _t(dev, tb['acc']),
                         _t(dev, tb['bias']) if tb['has_bias'] else None,
                         _t(dev, tb['bacc']) if tb['has_bias'] else None, cnt))
        dsites = [(t, _t(dev, cmap) if cmap is not None else None, _t(dev, ids), b, c)
                  for (t, cmap, ids, b, c) in sites]
        args = ops.MultiCatArgs(dtab, dsites)
        kb = torch.empty(m, dtype=torch.int32, device=dev)
        sb = torch.empty(m, dtype=torch.int32, device=dev)
        cb = torch.empty(m, dtype=torch.float32, device=dev)
        ops.sparse_adagrad_cat_multi(args, _t(dev, G), _t(dev, Gb), lr, kb, sb, cb, ws, gscale_dev=gs)
        torch.cuda.synchronize()
        for tt in dtab:
            assert int(tt[4].abs().sum().item()) == 0
        outs.append(dtab)
    for t in range(nt):
        rE, racc, rb, rbacc = refs[t]
        np.testing.assert_allclose(outs[0][t][1].cpu().numpy(), racc, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(outs[0][t][0].cpu().numpy(), rE, rtol=1e-4, atol=2e-5)
        if tabs[t]['has_bias']:
            np.testing.assert_allclose(outs[0][t][2].cpu().numpy(), rb, rtol=1e-4, atol=2e-5)
        assert torch.equal(outs[0][t][0], outs[1][t][0]) and torch.equal(outs[0][t][1], outs[1][t][1])


@pytest.mark.parametrize("n,rows,d,L", [(700, 90, 64, 5), (20000, 1500, 32, 3), (300, 300, 8, 1)])
def test_merged_sq_norm(dev, n, rows, d, L):
    """arx_merged_sq_norm: norm of a table gradient after contributions to the same row are summed
    (the dense per-step matmul gradient of embed_attribute.py:171,188 under seqModel.py:180)."""
    import torch
    from arx import ops
    rng = np.random.default_rng(n + d)
    S = 257
    keys = rng.integers(0, rows, size=n).astype(np.int32)
    pads = rng.random(n) < 0.2
    keys[pads] = ops.KEY_NONE
    src = rng.integers(0, S, size=n).astype(np.int32)
    coef = rng.random(n).astype(np.float32) + 0.1
    X = rng.standard_normal((L, S, d)).astype(np.float32)
    Xb = rng.standard_normal((L, S)).astype(np.float32)
    exp = 0.0
    for t in range(L):
        M = np.zeros((rows, d))
        mb = np.zeros(rows)
        live = ~pads
        np.add.at(M, keys[live], coef[live, None].astype(np.float64) * X[t][src[live]])
        np.add.at(mb, keys[live], coef[live].astype(np.float64) * Xb[t][src[live]])
        exp += (M ** 2).sum() + (mb ** 2).sum()
    out = torch.full((1,), 3.0, dtype=torch.float32, device=dev)
    ws = ops.Workspace(dev)
    ops.merged_sq_norm(_t(dev, keys), _t(dev, src), _t(dev, coef), rows, out, ws, X=_t(dev, X), d=d, L=L,
                       step_stride=S * d, Xb=_t(dev, Xb), Lb=L, stepb_stride=S)
    np.testing.assert_allclose(float(out.item()) - 3.0, exp, rtol=2e-5)
    # E part only, single step
    out.zero_()
    ops.merged_sq_norm(_t(dev, keys), _t(dev, src), _t(dev, coef), rows, out, ws, X=_t(dev, X[0]), d=d, L=1)
    M = np.zeros((rows, d))
    np.add.at(M, keys[~pads], coef[~pads, None].astype(np.float64) * X[0][src[~pads]])
    np.testing.assert_allclose(float(out.item()), (M ** 2).sum(), rtol=2e-5)


@pytest.mark.parametrize("d,n_ent,Vf,max_len,ns,phases", [
    (128, 3000, 400, 12, (4096, 1024), (3,)),        # radix both stages, Zipf duplicates, shared tokens
    (32, 50, 30, 5, (300, 17), (1, 2)),              # rank-sorted entities, radix tokens, two halves
    (64, 200, 5000, 3, (64,), (3,)),                 # everything below the rank-sort limit
    (128, 40000, 100002, 64, (20000, 1024), (1, 2)), # C3-like shape
    (16, 10, 8, 64, (9000, 0, 5), (3,)),             # few entities: runs of thousands, an empty site
    (32, 5000, 900, 6, (30000, 100), (1, 2)),        # > 24 k lookups: the entity stage takes the radix sort
    (64, 30000, 2000, 8, (16384, 1024), (1, 2)),     # C3-sized entity stage: the one-launch LDS sort
])
def test_sparse_adagrad_bags(dev, d, n_ent, Vf, max_len, ns, phases):
    """arx_sparse_adagrad_bags (merge per entity, then per token) == the plain contribution-level
    sum: g[tok] = sum over lookups r, tokens k of bag(id_r): coef_s / len * G[row_r]."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + n_ent + max_len)
    vals, starts, lens = _csr(rng, n_ent, Vf, max_len, zipf=True)
    vals[rng.integers(0, len(vals), size=3)] = Vf + 5          # out-of-range tokens are dropped
    E = rng.standard_normal((Vf, d)).astype(np.float32)
    acc = (0.1 + rng.random((Vf, d))).astype(np.float32)
    bias = rng.standard_normal((Vf,)).astype(np.float32)
    bacc = np.full((Vf,), 0.1, dtype=np.float32)
    m = sum(ns) + 7
    G = rng.standard_normal((m, d)).astype(np.float32)
    Gb = rng.standard_normal((m,)).astype(np.float32)
    p = 1.0 / np.arange(1, n_ent + 1) ** 1.05
    p /= p.sum()
    sites, keys, src, coef = [], [], [], []
    row0 = 3
    for s_i, n in enumerate(ns):
        ids = rng.choice(n_ent, size=n, p=p).astype(np.int32)
        if n > 4:
            ids[:2] = [n_ent + 3, -1]                          # invalid entity ids are dropped
        c = 0.5 / (s_i + 1)
        sites.append((ids, row0, c))
        for j, e in enumerate(ids):
            if 0 <= e < n_ent:
                for t in vals[starts[e]:starts[e] + lens[e]]:
                    if 0 <= t < Vf:
                        keys.append(t); src.append(row0 + j); coef.append(c / lens[e])
        row0 += n
    lr, gs = 0.3, 0.7
    rE, racc, rb, rbacc = _ref_sparse_adagrad(E, acc, bias, bacc, np.array(keys, dtype=np.int64),
                                              np.array(src, dtype=np.int64), np.array(coef), G, Gb, lr, gs)
    lr_dev = torch.tensor([lr], dtype=torch.float32, device=dev)
    gs_dev = torch.tensor([gs], dtype=torch.float32, device=dev)
    tv, tst, tl = _t(dev, vals), _t(dev, starts), _t(dev, lens)
    tG, tGb = _t(dev, G), _t(dev, Gb)
    outs = []
    for rep in range(2):
        tE, tacc, tb, tbacc = _t(dev, E), _t(dev, acc), _t(dev, bias), _t(dev, bacc)
        args = ops.BagSiteArgs([(_t(dev, ids), r0, c) for ids, r0, c in sites], max_len)
        ws = ops.Workspace(dev)
        cnt = torch.zeros(Vf, dtype=torch.int32, device=dev)
        for ph in phases:
            ops.sparse_adagrad_bags(tE, tacc, tb, tbacc, tv, tst, tl, args, tG, tGb, lr_dev, ws,
                                    gscale_dev=gs_dev, phase=ph, aux_cnt=cnt)
        torch.cuda.synchronize()
        outs.append((tE, tacc, tb, tbacc))
        assert int(cnt.abs().sum().item()) == 0
    tE, tacc, tb, tbacc = outs[0]
    np.testing.assert_allclose(tacc.cpu().numpy(), racc, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tE.cpu().numpy(), rE, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(tbacc.cpu().numpy(), rbacc, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tb.cpu().numpy(), rb, rtol=1e-4, atol=2e-5)
    for a, b in zip(outs[0], outs[1]):                 # bit-reproducible
        assert torch.equal(a, b)
    # no bias: the bias side is optional
    tE, tacc = _t(dev, E), _t(dev, acc)
    args = ops.BagSiteArgs([(_t(dev, ids), r0, c) for ids, r0, c in sites], max_len)
    ops.sparse_adagrad_bags(tE, tacc, None, None, tv, tst, tl, args, tG, None, lr_dev, ops.Workspace(dev),
                            gscale_dev=gs_dev)
    np.testing.assert_allclose(tE.cpu().numpy(), rE, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("d,n_ent,n_user,Vf,max_len,ns,phases", [
    (128, 40000, 50000, 100002, 48, (16384, 1024, 16384), (1, 2)),   # C3 HET shape: radix sort, window apply
    (32, 300, 100, 70, 6, (500, 40, 300), (3,)),                      # below the rank-sort limit
    (64, 5000, 7000, 900, 9, (6000, 0, 5000), (1, 2)),                # an empty site; 11 k keys
    (16, 12, 9, 8, 30, (9000, 100, 50), (3,)),                        # few entities: runs of thousands (finish kernel side output)
    (64, 12, 9, 8, 30, (9000, 100, 50), (1, 2)),                      # the same past d = 32: long runs of several work items (group.hip)
    (128, 2000, 3000, 50, 40, (30000, 500, 20000), (1, 2)),           # 50 token rows: long token runs, duplicate tokens inside the bags
    (256, 700, 50, 3000, 70, (9000, 0, 300), (3,)),                   # d = 256, bags longer than a wave
])
@pytest.mark.parametrize("virtual,sgd,maps", [(False, False, True), (True, False, True), (False, True, True),
                                              (True, True, True), (False, False, False), (True, False, False)])
def test_sparse_adagrad_cat_multi_bags(dev, d, n_ent, n_user, Vf, max_len, ns, phases, virtual, sgd, maps):
    """arx_sparse_adagrad_cat_multi_bags: the one-hot pass over (item id table, user table) with the
    item's multi-hot table riding on it == the three tables updated separately from the plain
    contribution lists (id rows: coef * G[row]; token rows: coef / len * G[row] per bag token).
    virtual: no item id table (MIX layout) -- table 0 of the pass is just the entity ids.
    maps: every table brings its zeroed per-row map -> past 8192 contributions the pass takes the
    grouped path (group.hip: no sort, run-centric apply); without them the radix / window path."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + n_ent + max_len)
    vals, starts, lens = _csr(rng, n_ent, Vf, max_len, zipf=True)
    vals[rng.integers(0, len(vals), size=3)] = Vf + 5          # out-of-range tokens are dropped
    m = sum(ns) + 5
    G = rng.standard_normal((m, d)).astype(np.float32)
    Gb = rng.standard_normal((m,)).astype(np.float32)
    p_it = 1.0 / np.arange(1, n_ent + 1) ** 1.05
    p_it /= p_it.sum()
    n_t, n_p, n_u = ns
    it_t = rng.choice(n_ent, size=n_t, p=p_it).astype(np.int32)
    it_p = rng.choice(n_ent, size=n_p, replace=n_p > n_ent).astype(np.int32)
    us = rng.integers(0, n_user, size=n_u).astype(np.int32)
    if n_t > 4:
        it_t[:2] = [n_ent + 3, n_ent + 5]                      # ids the map sends to -1 are dropped everywhere
    sites = [(0, it_t, 2, 0.5), (0, it_p, 2 + n_t, 0.25), (1, us, 2 + n_t + n_p, 1.0)]
    # entity -> id-table row: one-to-one, not the identity (vocabulary rows 0, 1 are reserved)
    ent2row = (rng.permutation(n_ent) + 2).astype(np.int32)
    n_rows0 = n_ent + 2

    def table(V):
        return (rng.standard_normal((V, d)).astype(np.float32), (0.1 + rng.random((V, d))).astype(np.float32),
                rng.standard_normal((V,)).astype(np.float32), np.full((V,), 0.1, dtype=np.float32))
    T_it, T_us, T_bag = table(n_rows0), table(n_user), table(Vf)
    lr, gs = 0.3, 0.7

    def ref(tab, keys, src, coef):
        E, acc, bias, bacc = [x.astype(np.float64).copy() for x in tab]
        g = np.zeros_like(E)
        gb = np.zeros_like(bias)
        np.add.at(g, keys, coef[:, None] * G[src].astype(np.float64))
        np.add.at(gb, keys, coef * Gb[src].astype(np.float64))
        g *= gs
        gb *= gs
        t = np.zeros(E.shape[0], bool)
        t[keys] = True
        if sgd:                                    # acc == NULL: plain gradient descent, slots untouched
            E[t] -= lr * g[t]
            bias[t] -= lr * gb[t]
            return E, acc, bias, bacc
        acc[t] += g[t] ** 2
        E[t] -= lr * g[t] / np.sqrt(acc[t])
        bacc[t] += gb[t] ** 2
        bias[t] -= lr * gb[t] / np.sqrt(bacc[t])
        return E, acc, bias, bacc

    def contribs(tb):
        ks, ss, cs = [], [], []
        for t, ids, r0, c in sites:
            if t != tb:
                continue
            ok = (ids >= 0) & (ids < (n_ent if tb == 0 else n_user))
            ks.append(ids[ok].astype(np.int64)); ss.append(r0 + np.nonzero(ok)[0]); cs.append(np.full(ok.sum(), c))
        return np.concatenate(ks), np.concatenate(ss), np.concatenate(cs)
    k0, s0, c0 = contribs(0)
    R_it = ref(T_it, ent2row[k0].astype(np.int64), s0, c0)
    R_us = ref(T_us, *contribs(1))
    bk, bs, bc = [], [], []
    for e, s_, c_ in zip(k0, s0, c0):
        toks = vals[starts[e]:starts[e] + lens[e]]
        ok = (toks >= 0) & (toks < Vf)
        bk.append(toks[ok].astype(np.int64)); bs.append(np.full(ok.sum(), s_)); bc.append(np.full(ok.sum(), c_ / lens[e]))
    R_bag = ref(T_bag, np.concatenate(bk), np.concatenate(bs), np.concatenate(bc))

    lr_dev = torch.tensor([lr], dtype=torch.float32, device=dev)
    gs_dev = torch.tensor([gs], dtype=torch.float32, device=dev)
    starts_r = np.zeros(n_rows0, dtype=np.int32)
    lens_r = np.zeros(n_rows0, dtype=np.int32)
    starts_r[ent2row] = starts
    lens_r[ent2row] = lens
    if virtual:
        starts_r, lens_r = starts, lens
    tv, tst, tl = _t(dev, vals), _t(dev, starts_r), _t(dev, lens_r)
    # the map is indexed by the lookup id: out-of-range ids must not reach it (the reference's ids are
    # always entities); the test's two invalid ids are mapped through a padded copy
    map_pad = np.full(n_ent + 8, -1, dtype=np.int32)
    map_pad[:n_ent] = ent2row
    tmap = _t(dev, map_pad)
    tG, tGb = _t(dev, G), _t(dev, Gb)
    outs = []
    # rep 2 (round 6): the bag table's STATIC token order (ops.BagCSC, csrc/csc.hip) instead of expansion + sort --
    # same lists, same bits; its flag bytes are zero again after every pass
    cs = ops.BagCSC(tv, tst, tl, max_len, Vf)
    assert cs.ok
    csc = (cs,) + cs.scratch()
    for rep in range(3):
        D_it, D_us, D_bag = [[_t(dev, x) for x in tab] for tab in (T_it, T_us, T_bag)]
        cnts = [torch.zeros(n_rows0, dtype=torch.int32, device=dev), torch.zeros(n_user, dtype=torch.int32, device=dev)]
        if sgd:
            for D in (D_it, D_us, D_bag):
                D[1] = D[3] = None
        vcnt = torch.zeros(n_ent, dtype=torch.int32, device=dev)
        if not maps:
            cnts_ = [None, None]
        else:
            cnts_ = cnts
        t0 = ((None, None, None, None, vcnt if maps else None, n_ent) if virtual
              else (D_it[0], D_it[1], D_it[2], D_it[3], cnts_[0]))
        args = ops.MultiCatArgs([t0, (D_us[0], D_us[1], D_us[2], D_us[3], cnts_[1])],
                                [(t, tmap if (t == 0 and not virtual) else None, _t(dev, ids), r0, c)
                                 for t, ids, r0, c in sites])
        n = args.total
        kb_ = torch.empty(n, dtype=torch.int32, device=dev)
        sb_ = torch.empty(n, dtype=torch.int32, device=dev)
        cb_ = torch.empty(n, dtype=torch.float32, device=dev)
        ws, bws = ops.Workspace(dev), ops.Workspace(dev)
        bcnt = torch.zeros(Vf, dtype=torch.int32, device=dev)
        for ph in phases:
            ops.sparse_adagrad_cat_multi_bags(args, tG, tGb, lr_dev, kb_, sb_, cb_, ws, D_bag[0], D_bag[1],
                                              D_bag[2], D_bag[3], tv, tst, tl, max_len, bws,
                                              gscale_dev=gs_dev, phase=ph, bag_aux_cnt=bcnt if maps else None,
                                              csc=csc if rep == 2 else None)
        torch.cuda.synchronize()
        assert int(csc[1].sum().item()) == 0
        assert int(bcnt.abs().sum().item()) == 0 and all(int(c.abs().sum().item()) == 0 for c in cnts)
        assert int(vcnt.abs().sum().item()) == 0
        outs.append(D_it + D_us + D_bag)
    want_it = list(T_it) if virtual else list(R_it)          # virtual: the id table is not part of the pass
    for got, want in zip(outs[0], want_it + list(R_us) + list(R_bag)):
        if got is not None:
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=2e-5)
    for a, b, c in zip(*outs):                         # bit-reproducible; the static token order: same bits
        assert a is None or (torch.equal(a, b) and torch.equal(a, c))


@pytest.mark.parametrize("d,sizes", [(128, (16384, 16384, 1024)), (32, (5, 0, 300)), (64, (7, 1, 2048))])
def test_lookup_multi(dev, d, sizes):
    """arx_lookup_multi: one-hot, multi-hot and (id + bag) lookups in one launch == the three
    single-site entry points, bit for bit (ragged sizes, an empty site, with and without bias)."""
    from arx import ops
    import torch
    rng = np.random.default_rng(d + sum(sizes))
    n_ent, Vf, Vu = 3000, 900, 700
    vals, starts, lens = _csr(rng, n_ent, Vf, 24, zipf=True)
    E_id = _t(dev, rng.standard_normal((n_ent + 2, d)).astype(np.float32))
    b_id = _t(dev, rng.standard_normal((n_ent + 2,)).astype(np.float32))
    cmap = _t(dev, (rng.permutation(n_ent) + 2).astype(np.int32))
    E_tok = _t(dev, rng.standard_normal((Vf, d)).astype(np.float32))
    b_tok = _t(dev, rng.standard_normal((Vf,)).astype(np.float32))
    E_u = _t(dev, rng.standard_normal((Vu, d)).astype(np.float32))
    b_u = _t(dev, rng.standard_normal((Vu,)).astype(np.float32))
    tv, tst, tl = _t(dev, vals), _t(dev, starts), _t(dev, lens)
    n_u, n_b, n_ib = sizes
    ids_u = _t(dev, rng.integers(0, Vu, size=n_u).astype(np.int32))
    ids_b = _t(dev, rng.integers(0, n_ent, size=n_b).astype(np.int32))
    ids_ib = _t(dev, rng.integers(0, n_ent, size=n_ib).astype(np.int32))
    for wb in (True, False):
        outs = [torch.full((n, d), 7.0, dtype=torch.float32, device=dev) for n in sizes]
        bouts = [torch.full((n,), 7.0, dtype=torch.float32, device=dev) if wb else None for n in sizes]
        ls = ops.LookupSet([
            (E_u, b_u if wb else None, None, None, None, None, None, None, ids_u, outs[0], 1.0, bouts[0]),
            (None, None, None, E_tok, b_tok if wb else None, tv, tst, tl, ids_b, outs[1], 0.7, bouts[1]),
            (E_id, b_id if wb else None, cmap, E_tok, b_tok if wb else None, tv, tst, tl, ids_ib, outs[2], 0.5,
             bouts[2])])
        ops.lookup_multi(ls)
        refs = [torch.empty_like(o) for o in outs]
        rb = [torch.empty_like(b) if wb else None for b in bouts]
        if n_u:
            ops.gather_onehot(E_u, b_u if wb else None, None, ids_u, refs[0], scale=1.0, bias_out=rb[0])
        if n_b:
            ops.gather_mulhot_mean(E_tok, b_tok if wb else None, tv, tst, tl, ids_b, refs[1], scale=0.7, bias_out=rb[1])
        if n_ib:
            ops.gather_id_plus_bag(E_id, b_id if wb else None, cmap, E_tok, b_tok if wb else None, tv, tst, tl,
                                   ids_ib, refs[2], scale=0.5, bias_out=rb[2])
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(outs[k], refs[k]), k
            if wb:
                assert torch.equal(bouts[k], rb[k]), k


def test_pool_bitmap_in_front_of_slot_map(dev):
    """arx_slot_map_attach_bitmap: the 1-bit "in the pool?" table follows arx_slot_map_set (set, clear,
    redraw with overlap, duplicates) and the `_pos` losses give bit-identical results with it."""
    from arx import ops
    import torch
    rng = np.random.default_rng(11)
    V, S, B, NU, NP = 5000, 256, 96, 300, 30
    maps = [torch.full((V + 1,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
    bits = torch.zeros((V + 1 + 32) // 32, dtype=torch.int32, device=dev)
    ops.slot_map_attach_bitmap(maps[0], bits)                       # maps[1]: probed directly
    logits = _t(dev, rng.standard_normal((B, S)).astype(np.float32))
    ts = _t(dev, rng.standard_normal((B,)).astype(np.float32))
    users = _t(dev, rng.integers(0, NU, size=B).astype(np.int32))
    ptr = _t(dev, (np.arange(NU + 1) * NP).astype(np.int32))
    old = None
    for rnd in range(3):
        pool = rng.choice(V, size=S, replace=False).astype(np.int32)
        if old is not None:
            pool[:S // 2] = old[:S // 2]                            # half of the previous pool stays
        pool[3] = pool[7]                                           # a duplicate id: the later slot wins
        items = rng.integers(0, V, size=NU * NP).astype(np.int32)
        items[::7] = pool[rng.integers(0, S, size=len(items[::7]))]      # some positives ARE in the pool
        t_items, t_pool = _t(dev, items), _t(dev, pool)
        outs = []
        for m in maps:
            if old is not None:
                ops.slot_map_set(m, _t(dev, old), clear=True)
            ops.slot_map_set(m, t_pool, clear=False)
            bl = torch.empty(B, device=dev); dl = torch.empty(B, S, device=dev); dt = torch.empty(B, device=dev)
            ops.loss_mw_pos(logits, ts, users, ptr, t_items, m, bl, dl, dt, 1.0 / B)
            outs.append((bl, dl, dt))
        torch.cuda.synchronize()
        assert torch.equal(maps[0], maps[1])
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        want = np.zeros(len(bits) * 32, dtype=bool)
        want[:V + 1] = maps[0].cpu().numpy() >= 0
        got = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder='little').astype(bool)
        assert np.array_equal(got, want), rnd
        assert float(outs[0][1].abs().sum().item()) > 0
        old = pool


def _unpack_bits(words, n):
    w = words.astype(np.uint32)
    return ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(w.shape[0], -1)[:, :n].astype(bool)


@pytest.mark.parametrize("B,S,d,mask_rows,maxpos", [(16384, 1024, 128, 0, 30), (320, 256, 64, 0, 30),
                                                     (1024, 2048, 128, 0, 30), (77, 128, 64, 0, 30),
                                                     (1, 128, 128, 0, 30), (4096, 512, 64, 1024, 30),
                                                     (200, 384, 128, 0, 30), (515, 256, 128, 0, 100),
                                                     (515, 256, 64, 0, 100)])
def test_mw_scorer(dev, B, S, d, mask_rows, maxpos):
    """The fused 'mw' scorer (csrc/scorer.hip: arx_mw_scorer_fwd / _bwd_du / _bwd_di): target score, loss, g, dt, the
    rank-one terms and the act BITS (both orientations) against the oracle's logits -> compute_loss('mw') ->
    compute_loss_bwd chain (embed_attribute.py:148-206, 208-220, 641-649), positives of the row's user masked
    (user of row r = users[r % mask_rows]: the sequence model's time-major rows); then the two backward products
    and the bias gradient out of the bits, incl. the per-time-step products (step_rows).  maxpos = 100: positives
    lists longer than the 32 entries a half wave of k_sc_prep lists per round (round 5), most of them past the
    kScHits = 8 pool slots a hit list holds (those rows walk their lists in k_sc_rows); B = 515: a ragged last
    8-row hit-list block and 128-row tile."""
    from arx import ops
    import torch
    rng = np.random.default_rng(B + S)
    U = (rng.standard_normal((B, d)) * 0.3).astype(np.float32)
    P = (rng.standard_normal((S, d)) * 0.3).astype(np.float32)
    pb = (rng.standard_normal(S) * 0.1).astype(np.float32)
    T = (rng.standard_normal((B, d)) * 0.3).astype(np.float32)
    tb = (rng.standard_normal(B) * 0.1).astype(np.float32)
    n_items, n_users = 5 * S, 50
    pool = rng.permutation(n_items)[:S].astype(np.int32)
    i2s = np.full(n_items + 1, -1, dtype=np.int32)
    i2s[pool] = np.arange(S, dtype=np.int32)
    npos = rng.integers(0, maxpos, size=n_users)
    ptr = np.concatenate([[0], np.cumsum(npos)]).astype(np.int32)
    pitems = rng.integers(0, n_items, size=int(ptr[-1])).astype(np.int32)     # duplicates happen
    mrows = mask_rows or B
    users = rng.integers(0, n_users, size=mrows).astype(np.int32)
    rw = rng.random(B).astype(np.float32)
    gscale = 1.0 / B
    # oracle
    logits = U.astype(np.float64) @ P.astype(np.float64).T + pb
    t = (U.astype(np.float64) * T).sum(1) + tb
    mask = np.ones((B, S), dtype=bool)
    for r in range(B):
        u = users[r % mrows]
        sl = i2s[pitems[ptr[u]:ptr[u + 1]]]
        mask[r, sl[sl >= 0]] = False
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.dt = np.dtype(np.float64)
    bl, cache = e.compute_loss(logits, t, 'mw', mask)
    dl, dt = e.compute_loss_bwd(cache, rw.astype(np.float64) * gscale)
    g = rw.astype(np.float64) * gscale / (1.0 + cache['s'])
    # device
    f32 = torch.float32
    sc = ops.MwScorer(B, S, d, dev)
    out_bl, out_t = (torch.empty(B, dtype=f32, device=dev) for _ in range(2))
    dU, dT = (torch.empty((B, d), dtype=f32, device=dev) for _ in range(2))
    dts = torch.empty(B, dtype=f32, device=dev)
    tU, tP = _t(dev, U), _t(dev, P)
    for rep in range(2):               # twice: the state is reusable (the second call overwrites every bit)
        sc.fwd(tU, tP, _t(dev, pb), _t(dev, T), _t(dev, tb), _t(dev, users), _t(dev, ptr), _t(dev, pitems),
               _t(dev, i2s), out_bl, out_t, dts, dU, dT, gscale, row_w=_t(dev, rw), mask_rows=mask_rows)
    torch.cuda.synchronize()
    words = sc.act_bits[:, :B].cpu().numpy()                              # [S / 32, B]
    act = _unpack_bits(np.ascontiguousarray(words.T).view(np.uint32), S)
    diff = act != cache['act']
    v = logits - t[:, None] + 1
    assert np.all(np.abs(v[diff]) < 1e-5) and diff.sum() <= max(3, B * S // 100000)   # fp32 borderline only
    # the transposed copy: bit (r & 31) of word [r >> 5][s]
    wt = sc.act_bits_t.cpu().numpy().view(np.uint32)                      # [Bp / 32, S]
    act_t = ((wt[:, None, :] >> np.arange(32, dtype=np.uint32)[None, :, None]) & 1).reshape(-1, S).astype(bool)
    assert np.array_equal(act_t[:B], act) and not act_t[B:].any()
    same = ~diff.any(axis=1)           # a row with a borderline hinge has another count: dt = -g * cnt moves by g
    np.testing.assert_allclose(out_t.cpu().numpy(), t, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out_bl.cpu().numpy(), bl, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(sc.g[:B].cpu().numpy(), g, rtol=RTOL, atol=1e-10)
    assert not sc.g[B:].cpu().numpy().any()
    np.testing.assert_allclose(dts.cpu().numpy()[same], dt[same], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(dU.cpu().numpy()[same], (dt[:, None] * T)[same], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(dT.cpu().numpy()[same], (dt[:, None] * U)[same], rtol=RTOL, atol=1e-9)
    assert not np.any(act & ~mask)                                  # masked positives carry no bit
    # ---- the backward products read the bits ----
    # Expected values from the DEVICE's own bits: a borderline hinge (|x - t + 1| < 1e-5, fp32 vs f64) flips a bit
    # and the products legitimately follow the bit -- with the oracle's act matrix the check had to be skipped
    # whenever ONE of the B * S bits differed, which at the C3 shape (16.8 M logits) is the expected case.  g moves by
    # < 1e-5 relative with such a flip (the hinge sum by |v| < 1e-5), inside the tolerance.
    dl_dev = g[:, None] * act                                       # d loss / d logits as the device sees it
    dt_dev = -dl_dev.sum(1)
    dUx = dU.clone()
    sc.bwd_dU(dUx, beta=1.0)                                                     # dU += g * (act . P)
    np.testing.assert_allclose(dUx.cpu().numpy(), dt_dev[:, None] * T + (dl_dev @ P.astype(np.float64)), rtol=RTOL,
                               atol=3e-8)
    # ... and against the oracle's own backward on every row without a flipped bit: almost all of them
    assert same.mean() >= 0.99, same.mean()
    np.testing.assert_allclose(dUx.cpu().numpy()[same], (dt[:, None] * T + (dl @ P.astype(np.float64)))[same],
                               rtol=RTOL, atol=2e-8)
    dI0 = rng.standard_normal((S, d)).astype(np.float32)
    dI = _t(dev, dI0)
    db = torch.empty(S, dtype=f32, device=dev)
    lsum = torch.full((1,), 7.0, dtype=f32, device=dev)
    sc.bwd_dI(dI, db=db, beta=0.5, loss=(out_bl, gscale, _t(dev, rw), lsum))     # dI = 0.5 dI + act^T . (g U)
    # the step's scalar out of the same reduce launch: gscale * sum_r row_w_r * loss_r (f32 sums of <= B terms)
    ref_sum = float((rw.astype(np.float64) * gscale * bl).sum())
    bound = 4e-7 * float((rw.astype(np.float64) * gscale * np.abs(bl)).sum()) * (6 + np.log2(max(B, 2))) + 1e-12
    assert abs(float(lsum.item()) - ref_sum) <= bound, (float(lsum.item()), ref_sum, bound)
    np.testing.assert_allclose(dI.cpu().numpy(), 0.5 * dI0 + dl_dev.T @ U.astype(np.float64), rtol=RTOL, atol=3e-8)
    np.testing.assert_allclose(db.cpu().numpy(), dl_dev.sum(0), rtol=RTOL, atol=3e-8)
    same_c = ~diff.any(axis=0)         # pool columns without a flipped bit: the oracle's own dI rows / db cells
    assert same_c.mean() >= 0.99, same_c.mean()
    np.testing.assert_allclose(dI.cpu().numpy()[same_c], (0.5 * dI0 + dl.T @ U.astype(np.float64))[same_c],
                               rtol=RTOL, atol=2e-8)
    np.testing.assert_allclose(db.cpu().numpy()[same_c], dl.sum(0)[same_c], rtol=RTOL, atol=2e-8)
    if mask_rows:                      # per-time-step products (TF-1.0's clip norm squares them one by one)
        L = B // mask_rows
        dIs = torch.empty((L, S, d), dtype=f32, device=dev)
        dbs = torch.empty((L, S), dtype=f32, device=dev)
        dI2 = torch.empty((S, d), dtype=f32, device=dev)
        sc.bwd_dI(dI2, db=db, step_rows=mask_rows, dI_steps=dIs, db_steps=dbs)
        for k in range(L):
            rows = slice(k * mask_rows, (k + 1) * mask_rows)
            np.testing.assert_allclose(dIs[k].cpu().numpy(), dl_dev[rows].T @ U[rows].astype(np.float64), rtol=RTOL,
                                       atol=3e-8)
            np.testing.assert_allclose(dbs[k].cpu().numpy(), dl_dev[rows].sum(0), rtol=RTOL, atol=3e-8)
        np.testing.assert_allclose(dI2.cpu().numpy(), dl_dev.T @ U.astype(np.float64), rtol=RTOL, atol=3e-8)
    if mask_rows:
        # the sequence model's example weights formed by the first launch (arx_mw_scorer_fwd_seqw): the arithmetic
        # of arx_seq_weights bit for bit, and g / the scalar loss follow from them
        L = B // mask_rows
        w_raw = _t(dev, (rng.random(B) * (rng.random(B) > 0.2)).astype(np.float32))
        wn_ref = torch.empty(B, dtype=f32, device=dev)
        ops.seq_weights(w_raw, L, mask_rows, wn_ref)
        wn = torch.full((B,), 7.0, dtype=f32, device=dev)
        sc.fwd(tU, tP, _t(dev, pb), _t(dev, T), _t(dev, tb), _t(dev, users), _t(dev, ptr), _t(dev, pitems),
               _t(dev, i2s), out_bl, out_t, dts, dU, dT, 1.0, row_w=wn, mask_rows=mask_rows, seq_w=w_raw,
               seq_rows=mask_rows)
        assert torch.equal(wn, wn_ref)
        wn64 = wn.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(sc.g[:B].cpu().numpy(), wn64 / (1.0 + cache['s']), rtol=RTOL, atol=1e-10)
        sc.bwd_dI(dI, db=db, loss=(out_bl, 1.0, wn, lsum))
        ref_sum = float((wn64 * bl).sum())
        assert abs(float(lsum.item()) - ref_sum) <= 4e-7 * ref_sum * (6 + np.log2(B)) + 1e-12


@pytest.mark.parametrize("B,S,mask_rows,maxpos,bias", [(320, 256, 0, 30, True), (77, 128, 0, 30, True),
                                                        (1, 128, 0, 30, False), (4096, 512, 1024, 30, True),
                                                        (515, 2048, 0, 100, True), (2048, 1024, 0, 400, True),
                                                        (5120, 1024, 1024, 30, False)])
@pytest.mark.parametrize("d", [64, 128])
def test_mce_scorer(dev, B, S, mask_rows, maxpos, bias, d):
    """The build-defined sampled softmax 'mce' on the fused family (csrc/scorer.hip k_mc_flow / k_mc_rows:
    arx_mce_scorer_fwd / _bwd_di_loss, d = 64 and -- round 6 -- d = 128) against the oracle's logits -> compute_loss('mce') ->
    compute_loss_bwd chain in f64: loss, target score, dt, dT = dt U, the COMPLETE latent gradient dU = dl . P + dt T
    out of the forward, then dI = beta dI + dl^T . U, db, the step's scalar loss and the per-time-step products.  No
    [B, S] array exists on the device; the positives of the row's user are masked pairs (maxpos = 400: rows with
    dozens of them in the pool); B = 515 / 77 / 1: ragged stationary and stream tiles; S = 2048: four column splits."""
    from arx import ops
    import torch
    assert ops.mce_scorer_supported(B, S, d)
    rng = np.random.default_rng(B + S + 1)
    U = (rng.standard_normal((B, d)) * 0.4).astype(np.float32)
    P = (rng.standard_normal((S, d)) * 0.4).astype(np.float32)
    pb = (rng.standard_normal(S) * 0.2).astype(np.float32) if bias else None
    T = (rng.standard_normal((B, d)) * 0.4).astype(np.float32)
    tb = (rng.standard_normal(B) * 0.1).astype(np.float32)
    n_items, n_users = 5 * S, 50
    pool = rng.permutation(n_items)[:S].astype(np.int32)
    i2s = np.full(n_items + 1, -1, dtype=np.int32)
    i2s[pool] = np.arange(S, dtype=np.int32)
    npos = rng.integers(0, maxpos, size=n_users)
    ptr = np.concatenate([[0], np.cumsum(npos)]).astype(np.int32)
    pitems = rng.integers(0, n_items, size=int(ptr[-1])).astype(np.int32)
    mrows = mask_rows or B
    users = rng.integers(0, n_users, size=mrows).astype(np.int32)
    rw = rng.random(B).astype(np.float32)
    gscale = 1.0 / B
    logits = U.astype(np.float64) @ P.astype(np.float64).T + (pb if bias else 0.0)
    t = (U.astype(np.float64) * T).sum(1) + tb
    mask = np.ones((B, S), dtype=bool)
    for r in range(B):
        u = users[r % mrows]
        sl = i2s[pitems[ptr[u]:ptr[u + 1]]]
        mask[r, sl[sl >= 0]] = False
    assert (~mask).sum() > 0 or B == 1
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.dt = np.dtype(np.float64)
    bl, cache = e.compute_loss(logits, t, 'mce', mask)
    dl, dt = e.compute_loss_bwd(cache, rw.astype(np.float64) * gscale)
    U64, P64 = U.astype(np.float64), P.astype(np.float64)
    f32 = torch.float32
    sc = ops.MceScorer(B, S, d, dev)
    out_bl, out_t = (torch.empty(B, dtype=f32, device=dev) for _ in range(2))
    dU, dT = (torch.full((B, d), 9.0, dtype=f32, device=dev) for _ in range(2))
    dts = torch.empty(B, dtype=f32, device=dev)
    tU, tP, tpb = _t(dev, U), _t(dev, P), (_t(dev, pb) if bias else None)
    args = (_t(dev, T), _t(dev, tb), _t(dev, users), _t(dev, ptr), _t(dev, pitems), _t(dev, i2s))
    for rep in range(2):               # twice: the state (mask tables included) is reusable
        sc.fwd(tU, tP, tpb, *args, out_bl, out_t, dts, dU, dT, gscale, row_w=_t(dev, rw), mask_rows=mask_rows)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_t.cpu().numpy(), t, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out_bl.cpu().numpy(), bl, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(dts.cpu().numpy(), dt, rtol=RTOL, atol=1e-10)
    np.testing.assert_allclose(dT.cpu().numpy(), dt[:, None] * U64, rtol=RTOL, atol=1e-9)
    scale_u = np.abs(dl) @ np.abs(P64) + np.abs(dt[:, None] * T)           # term scale of a gradient entry
    errU = np.abs(dU.cpu().numpy() - (dl @ P64 + dt[:, None] * T))
    assert np.all(errU <= 2e-6 * scale_u + 1e-12), float((errU / (scale_u + 1e-30)).max())
    sc.bwd_dU(dU, beta=1.0)            # (no-op: the forward wrote the whole latent gradient)
    dI0 = rng.standard_normal((S, d)).astype(np.float32)
    dI = _t(dev, dI0)
    db = torch.empty(S, dtype=f32, device=dev)
    lsum = torch.full((1,), 7.0, dtype=f32, device=dev)
    sc.bwd_dI(dI, db=db, beta=0.5, loss=(out_bl, gscale, _t(dev, rw), lsum))
    ref_sum = float((rw.astype(np.float64) * gscale * bl).sum())
    bound = 4e-7 * float((rw.astype(np.float64) * gscale * np.abs(bl)).sum()) * (6 + np.log2(max(B, 2))) + 1e-12
    assert abs(float(lsum.item()) - ref_sum) <= bound, (float(lsum.item()), ref_sum, bound)
    scale_i = np.abs(dl).T @ np.abs(U64)
    errI = np.abs(dI.cpu().numpy() - (0.5 * dI0 + dl.T @ U64))
    assert np.all(errI <= 2e-6 * scale_i + 1e-7 * np.abs(dI0) + 1e-12), float((errI / (scale_i + 1e-30)).max())
    np.testing.assert_allclose(db.cpu().numpy(), dl.sum(0), rtol=2e-5, atol=1e-12)
    if mask_rows:
        L = B // mask_rows
        dIs = torch.empty((L, S, d), dtype=f32, device=dev)
        dbs = torch.empty((L, S), dtype=f32, device=dev)
        dI2 = torch.empty((S, d), dtype=f32, device=dev)
        sc.bwd_dI(dI2, db=db, step_rows=mask_rows, dI_steps=dIs, db_steps=dbs)
        for k in range(L):
            rows = slice(k * mask_rows, (k + 1) * mask_rows)
            ref = dl[rows].T @ U64[rows]
            err = np.abs(dIs[k].cpu().numpy() - ref)
            assert np.all(err <= 2e-6 * (np.abs(dl[rows]).T @ np.abs(U64[rows])) + 1e-12)
            np.testing.assert_allclose(dbs[k].cpu().numpy(), dl[rows].sum(0), rtol=2e-5, atol=1e-12)
        err = np.abs(dI2.cpu().numpy() - dl.T @ U64)
        assert np.all(err <= 2e-6 * scale_i + 1e-12)
        # the sequence model's example weights formed by the first launch, as in arx_mw_scorer_fwd_seqw
        w_raw = _t(dev, (rng.random(B) * (rng.random(B) > 0.2)).astype(np.float32))
        wn_ref = torch.empty(B, dtype=f32, device=dev)
        ops.seq_weights(w_raw, L, mask_rows, wn_ref)
        wn = torch.full((B,), 7.0, dtype=f32, device=dev)
        sc.fwd(tU, tP, tpb, *args, out_bl, out_t, dts, dU, dT, 1.0, row_w=wn, mask_rows=mask_rows, seq_w=w_raw,
               seq_rows=mask_rows)
        assert torch.equal(wn, wn_ref)
        wn64 = wn.cpu().numpy().astype(np.float64)
        _, dt2 = e.compute_loss_bwd(cache, wn64)
        np.testing.assert_allclose(dts.cpu().numpy(), dt2, rtol=RTOL, atol=1e-10)


@pytest.mark.parametrize("B,S,scale", [(320, 256, 2.0), (515, 1024, 3.0)])
def test_mce_scorer_large_logits_stay_finite(dev, B, S, scale):
    """Round-5 advisor (medium, scorer.hip k_mc_flow): the fused 'mce' family anchors exp() at the target score; a
    pair with x_rs - t_r > 88.7 made s = inf, coef = 0, dt = NaN and poisoned dU / dT / the tables for good.  The
    build-defined loss now saturates its exponent at 64 (one definition: csrc/common.h kMceSat, oracle MCE_SAT):
    operands scaled until hundreds of pairs pass 88.7 -- every output finite, and equal to the oracle's (saturating)
    f64 chain: loss, dt, dT, dU, dI, db.  Gradient tolerance 3e-4 of the term scale: |x| ~ 100 carries an f32 product
    chain's ~2e-5 absolute error into the exponent."""
    from arx import ops
    import torch
    d = 64
    rng = np.random.default_rng(B + S)
    U = (rng.standard_normal((B, d)) * scale).astype(np.float32)
    U[:64] *= 0.05                                            # plain rows: every logit within a few units of t
    P = (rng.standard_normal((S, d)) * scale).astype(np.float32)
    pb = (rng.standard_normal(S) * 0.2).astype(np.float32)
    T = (rng.standard_normal((B, d)) * 0.4).astype(np.float32)
    tb = (rng.standard_normal(B) * 0.1).astype(np.float32)
    n_items, n_users = 5 * S, 50
    pool = rng.permutation(n_items)[:S].astype(np.int32)
    i2s = np.full(n_items + 1, -1, dtype=np.int32)
    i2s[pool] = np.arange(S, dtype=np.int32)
    npos = rng.integers(0, 30, size=n_users)
    ptr = np.concatenate([[0], np.cumsum(npos)]).astype(np.int32)
    pitems = rng.integers(0, n_items, size=int(ptr[-1])).astype(np.int32)
    users = rng.integers(0, n_users, size=B).astype(np.int32)
    rw = rng.random(B).astype(np.float32)
    gscale = 1.0 / B
    U64, P64 = U.astype(np.float64), P.astype(np.float64)
    logits = U64 @ P64.T + pb
    t = (U64 * T).sum(1) + tb
    mask = np.ones((B, S), dtype=bool)
    for r in range(B):
        sl = i2s[pitems[ptr[users[r]]:ptr[users[r] + 1]]]
        mask[r, sl[sl >= 0]] = False
    lead = np.where(mask, logits - t[:, None], -np.inf)
    assert (lead > 88.7).sum() > 100 and (lead.max(1) < 64).sum() > 10      # overflowing rows AND plain ones
    e = rg.RefEmbeddingAttribute.__new__(rg.RefEmbeddingAttribute)
    e.dt = np.dtype(np.float64)
    bl, cache = e.compute_loss(logits, t, 'mce', mask)
    dl, dt = e.compute_loss_bwd(cache, rw.astype(np.float64) * gscale)
    f32 = torch.float32
    sc = ops.MceScorer(B, S, d, dev)
    out_bl, out_t, dts = (torch.empty(B, dtype=f32, device=dev) for _ in range(3))
    dU, dT = (torch.full((B, d), 9.0, dtype=f32, device=dev) for _ in range(2))
    sc.fwd(_t(dev, U), _t(dev, P), _t(dev, pb), _t(dev, T), _t(dev, tb), _t(dev, users), _t(dev, ptr), _t(dev, pitems),
           _t(dev, i2s), out_bl, out_t, dts, dU, dT, gscale, row_w=_t(dev, rw))
    dI = torch.zeros((S, d), dtype=f32, device=dev)
    db = torch.empty(S, dtype=f32, device=dev)
    sc.bwd_dI(dI, db=db, beta=0.0)
    torch.cuda.synchronize()
    for x in (out_bl, out_t, dts, dU, dT, dI, db):
        assert bool(torch.isfinite(x).all())
    np.testing.assert_allclose(out_bl.cpu().numpy(), bl, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(dts.cpu().numpy(), dt, rtol=3e-4, atol=1e-10)
    np.testing.assert_allclose(dT.cpu().numpy(), dt[:, None] * U64, rtol=3e-4, atol=1e-9)
    scale_u = np.abs(dl) @ np.abs(P64) + np.abs(dt[:, None] * T)
    errU = np.abs(dU.cpu().numpy() - (dl @ P64 + dt[:, None] * T))
    assert np.all(errU <= 3e-4 * scale_u + 1e-12), float((errU / (scale_u + 1e-30)).max())
    scale_i = np.abs(dl).T @ np.abs(U64)
    errI = np.abs(dI.cpu().numpy() - dl.T @ U64)
    assert np.all(errI <= 3e-4 * scale_i + 1e-12), float((errI / (scale_i + 1e-30)).max())
    np.testing.assert_allclose(db.cpu().numpy(), dl.sum(0), rtol=3e-4, atol=1e-12)


@pytest.mark.parametrize("M,N,K,beta", [(51200, 64, 256, 0.0), (777, 64, 256, 1.0), (130, 128, 128, 0.5),
                                          (1, 32, 64, 0.0), (4096, 96, 64, 0.0)])
def test_gemm_bt_bx6(dev, M, N, K, beta):
    """arx_gemm_bt_bx6 (the LSTM's dx = dz . W_x^T on the six-term bf16 tiles, W_x read as it lies, the rows of dz
    streamed through LDS and split on the fly): against an f64 product at 1e-6 of the term scale -- an f32 FMA chain's
    own error -- incl. ragged row blocks and a strided second operand (rows of a wider matrix)."""
    from arx import ops
    import torch
    rng = np.random.default_rng(M + N + K)
    A = (rng.standard_normal((M, K)) * (2.0 ** rng.integers(-8, 9, size=(M, 1)))).astype(np.float32)
    W = rng.standard_normal((N + 5, K + 8)).astype(np.float32)          # Bt = rows [0, N), columns [0, K) of it
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    tA, tW, tC = _t(dev, A), _t(dev, W), _t(dev, C0)
    from arx import _lib
    assert _lib.lib.arx_gemm_bt_bx6_supported(M, N, K)      # (ops.gemm_bt_bx6_supported also reads ARX_SCORER_F32)
    ops.gemm_bt_bx6(tA, tW[:N, :K], tC, beta=beta)
    Bt = W[:N, :K].astype(np.float64)
    ref = beta * C0 + A.astype(np.float64) @ Bt.T
    scale = np.abs(A).astype(np.float64) @ np.abs(Bt).T + np.abs(beta * C0)
    err = np.abs(tC.cpu().numpy() - ref)
    assert np.all(err <= 1e-6 * scale + 1e-30), float((err / (scale + 1e-30)).max())


def test_mw_scorer_products_f32_exact(dev):
    """The piece arithmetic of the scorer at its edges (the round-3 ruling's condition ii).  Stated input domain:
    finite f32 operands whose products and sums stay inside the f32 normal range -- what an f32 FMA chain needs as
    well; non-finite inputs are covered by the next test.  Here: operand magnitudes over 2^-20 .. 2^20 per row /
    column (products over 2^-40 .. 2^40), a row of 2^60 against a row of 2^-60 (pieces 2 and 3 of tiny values are
    bf16 subnormals or zero: they must not be needed), a value whose bf16 rounding carries into the exponent
    (0x3F7FFFFF), negative zeros.  The hinge decision equals an f64 evaluation except where the f64 sum itself is a
    rounding away from zero; the row sums and the backward product match f64 at 1e-5 of the term scale."""
    from arx import ops
    import torch
    B, S, d = 256, 128, 128
    rng = np.random.default_rng(7)
    U = rng.standard_normal((B, d)).astype(np.float32)
    P = rng.standard_normal((S, d)).astype(np.float32)
    U *= (2.0 ** rng.integers(-20, 21, size=(B, 1))).astype(np.float32)
    P *= (2.0 ** rng.integers(-20, 21, size=(S, 1))).astype(np.float32)
    U[0, :] = np.float32(1.0)
    P[0, :] = np.frombuffer(np.uint32(0x3F7FFFFF).tobytes(), dtype=np.float32)[0]     # rounds UP to 1.0 in bf16
    U[1, ::2] = -0.0
    P[1, :] = 2.0 ** -60
    U[2, :] = 2.0 ** 60
    pb = np.zeros(S, np.float32)
    T = np.zeros((B, d), np.float32)
    tb = np.ones(B, np.float32)                                   # t = 1: v = x
    users = np.zeros(B, np.int32)
    ptr = np.zeros(3, np.int32)
    i2s = np.full(S + 1, -1, np.int32)
    sc = ops.MwScorer(B, S, d, dev)
    f32 = torch.float32
    bl, tt, dts = (torch.empty(B, dtype=f32, device=dev) for _ in range(3))
    dU, dT = (torch.zeros((B, d), dtype=f32, device=dev) for _ in range(2))
    sc.fwd(_t(dev, U), _t(dev, P), _t(dev, pb), _t(dev, T), _t(dev, tb), _t(dev, users), _t(dev, ptr),
           _t(dev, np.zeros(1, np.int32)), _t(dev, i2s), bl, tt, dts, dU, dT, 1.0)
    x = U.astype(np.float64) @ P.astype(np.float64).T
    scale = np.abs(U.astype(np.float64)) @ np.abs(P.astype(np.float64)).T
    act = _unpack_bits(np.ascontiguousarray(sc.act_bits[:, :B].cpu().numpy().T).view(np.uint32), S)
    wrong = act != (x > 0)
    assert np.all(np.abs(x[wrong]) <= 1e-6 * scale[wrong])
    s_ref = np.where(x > 0, x, 0.0).sum(1)
    got = np.expm1(bl.cpu().numpy().astype(np.float64))          # loss = log(1 + s)
    ok = (s_ref < 1e30) & (s_ref > 1e-3)                         # (log1p / expm1 of tiny or huge sums: f32 log range)
    assert np.isfinite(bl.cpu().numpy()).all()
    np.testing.assert_allclose(got[ok], s_ref[ok], rtol=2e-5)
    g = sc.g[:B].cpu().numpy().astype(np.float64)
    sc.bwd_dU(dU, beta=0.0)
    ref = g[:, None] * (act.astype(np.float64) @ P.astype(np.float64))
    bound = g[:, None] * (act.astype(np.float64) @ np.abs(P.astype(np.float64)))
    assert np.all(np.abs(dU.cpu().numpy() - ref) <= 1e-5 * bound + 1e-37)


def test_mw_scorer_nonfinite_inputs_propagate(dev):
    """NaN / Inf in a latent row or a pool row (the ruling's condition ii): the split of a non-finite value keeps it
    non-finite (piece 1 = the value, pieces 2 / 3 = NaN), so the affected rows' losses come out non-finite -- as
    they do with an f32 FMA chain -- and every OTHER row is untouched; an Inf logit that loses the hinge
    (x = -inf) contributes nothing."""
    from arx import ops
    import torch
    B, S, d = 128, 128, 64
    rng = np.random.default_rng(3)
    U = (rng.standard_normal((B, d)) * 0.3).astype(np.float32)
    P = (rng.standard_normal((S, d)) * 0.3).astype(np.float32)
    clean_U, clean_P = U.copy(), P.copy()
    U[5, 3] = np.nan
    U[9, 0] = np.inf
    pb = np.zeros(S, np.float32)
    T = (rng.standard_normal((B, d)) * 0.3).astype(np.float32)
    tb = np.zeros(B, np.float32)
    users = np.zeros(B, np.int32)
    ptr = np.zeros(3, np.int32)
    i2s = np.full(S + 1, -1, np.int32)
    f32 = torch.float32
    res = []
    for UU, PP in ((U, P), (clean_U, clean_P)):
        sc = ops.MwScorer(B, S, d, dev)
        bl, tt, dts = (torch.empty(B, dtype=f32, device=dev) for _ in range(3))
        dU, dT = (torch.zeros((B, d), dtype=f32, device=dev) for _ in range(2))
        sc.fwd(_t(dev, UU), _t(dev, PP), _t(dev, pb), _t(dev, T), _t(dev, tb), _t(dev, users), _t(dev, ptr),
               _t(dev, np.zeros(1, np.int32)), _t(dev, i2s), bl, tt, dts, dU, dT, 1.0)
        res.append(bl.cpu().numpy())
    dirty, clean = res
    assert not np.isfinite(dirty[5]) and not np.isfinite(dirty[9])
    keep = np.ones(B, bool)
    keep[[5, 9]] = False
    assert np.array_equal(dirty[keep], clean[keep])


@pytest.mark.parametrize("M,N,K", [(16384, 1024, 128), (4097, 256, 64), (70, 128, 128)])
def test_gemm_nt_bx6(dev, M, N, K):
    """arx_gemm_nt_bx6 (experimental logits GEMM on the bf16 pipe, three exact bf16 pieces per f32 operand, six
    MFMAs per product term): against an f64 product its error is no larger than the f32-MFMA kernel's on the
    same inputs -- rows of very different magnitude, a bias, ragged M, strided operands -- and within the
    f32 bound K * 2^-24 * sum |a||b|; logits equal to rtol 1e-6 of that scale."""
    from arx import ops
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(M + K)
    Ubuf = torch.randn(M, K + 8, device=dev, generator=g) * torch.exp(3 * torch.randn(M, 1, device=dev, generator=g))
    U = Ubuf[:, :K]                                                   # lda = K + 8
    I = torch.randn(N, K, device=dev, generator=g)
    I[:, 0] = torch.arange(N, device=dev) * 0.01                      # asymmetric: a swapped C layout cannot pass
    b = torch.randn(N, device=dev, generator=g)
    Lbuf = torch.full((M, N + 4), 7.0, device=dev)
    L1 = Lbuf[:, :N]                                                  # ldc = N + 4
    L0 = torch.empty(M, N, device=dev)
    ops.gemm(U, I, L0, ops.Workspace(dev), transB=True, col_bias=b)   # f32 MFMA
    ops.gemm_nt_bx6(U, I, L1, b)
    ref = U.double() @ I.double().T + b.double()
    scale = U.double().abs() @ I.double().abs().T + b.double().abs()
    e0 = ((L0.double() - ref).abs() / scale)
    e1 = ((L1.double() - ref).abs() / scale)
    assert float(e1.max()) <= K * 2.0 ** -24
    assert float(e1.mean()) <= float(e0.mean()) * 1.05 and float(e1.max()) <= float(e0.max()) * 1.05
    assert bool((Lbuf[:, N:] == 7.0).all())                           # nothing written past the row


def test_captured_graph_feed_nodes(dev):
    """Placeholder feeds captured as graph nodes (arx_capture_end_feeds / arx_graph_set_feed): a replay copies
    whatever sources were set last, nine feeds = two copy nodes told apart by their first destination, None = the
    nodes copy nothing, and a feed list over other destinations is refused by feeds_match."""
    from arx import ops
    import torch
    n_feeds = 9
    dst = [torch.zeros(100 + 7 * k, dtype=torch.int32, device=dev) for k in range(n_feeds)]
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    mk = lambda base: [torch.full((100 + 7 * k,), base + k, dtype=torch.int32, device=dev) for k in range(n_feeds)]
    src0 = mk(10)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = ops.CapturedGraph()
    feeds0 = list(zip(src0, dst))
    with torch.cuda.stream(side):
        g.begin()
        ops.copy_words(feeds0)
        ops.sum_scaled(dst[8].view(torch.float32), 1.0, out)       # a consumer behind the feed nodes
        g.end(feeds=feeds0)
    torch.cuda.current_stream().wait_stream(side)
    g.launch()
    torch.cuda.synchronize()
    assert all(int(d[0]) == 10 + k and int(d[-1]) == 10 + k for k, d in enumerate(dst))
    src1 = mk(50)
    feeds1 = list(zip(src1, dst))
    assert g.feeds_match(feeds1) and not g.feeds_match(feeds1[:8]) and not g.feeds_match(list(zip(src1, src0)))
    g.set_feeds(feeds1)
    g.launch()
    torch.cuda.synchronize()
    assert all(bool((d == 50 + k).all()) for k, d in enumerate(dst))
    for d in dst:
        d.zero_()
    g.set_feeds(None)                     # nothing fed: the nodes must not re-copy the last sources
    g.launch()
    g.launch()
    torch.cuda.synchronize()
    assert all(not bool(d.any()) for d in dst)
    g.set_feeds(feeds0)
    g.launch()
    torch.cuda.synchronize()
    assert all(bool((d == 10 + k).all()) for k, d in enumerate(dst))


def test_captured_graph_feed_nodes_in_flight(dev):
    """Feed nodes re-pointed while EARLIER launches of the same executable are still in flight (advisor, round 4):
    hipGraphExecKernelNodeSetParams must only affect launches issued after it.  The graph is made long (a few passes
    over 64 MB behind the consumer) so that the host runs ahead: 48 set_feeds + launch pairs with distinct sources
    and no synchronisation in between; each launch's consumer result is copied out by an eager copy ordered behind
    it.  Validated on ROCm 7.2 / gfx950 (the step's default path relies on it: Runtime.feeds_in_graph)."""
    from arx import ops
    import torch
    n, K = 4096, 48
    f32, i32 = torch.float32, torch.int32
    dst = torch.zeros(n, dtype=i32, device=dev)
    out = torch.zeros(1, dtype=f32, device=dev)
    sink = torch.zeros(1, dtype=f32, device=dev)
    big = torch.ones(1 << 24, dtype=f32, device=dev)
    srcs = [torch.full((n,), float(k + 1), dtype=f32, device=dev).view(i32) for k in range(K)]
    res = torch.zeros(K, dtype=f32, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = ops.CapturedGraph()
    feeds0 = [(srcs[0], dst)]
    with torch.cuda.stream(side):
        g.begin()
        ops.copy_words(feeds0)
        ops.sum_scaled(dst.view(f32), 1.0, out)                   # the consumer behind the feed node
        for _ in range(6):
            ops.sum_scaled(big, 1.0, sink)                        # ... and ~100 us of other work: launches queue up
        g.end(feeds=feeds0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for k in range(K):
        g.set_feeds([(srcs[k], dst)])
        g.launch()
        res[k:k + 1].copy_(out, non_blocking=True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    got = res.cpu().numpy()
    assert np.array_equal(got, np.arange(1, K + 1, dtype=np.float32) * n), got
    # (informational: the host was ahead of the device for most of the run when t_host << t_all)
    print("feed nodes in flight: host %.2f ms, device %.2f ms" % (1e3 * t_host, 1e3 * t_all))


@pytest.mark.parametrize("rows,d,frac", [(100000, 128, 0.3), (777, 64, 0.5), (5, 32, 1.0), (4096, 128, 0.0)])
def test_adagrad_rows_nonzero_equals_dense_step(dev, rows, d, frac):
    """arx_adagrad_rows_nonzero (the replicated token table of the sharded HET step): the rows whose gradient row is
    not all zero move exactly as under arx_adagrad_dense, the others not at all (bit for bit: the dense step on a zero
    gradient is the identity), the bias vector alike, and the gradient table comes back all zero."""
    from arx import ops
    import torch
    rng = np.random.default_rng(rows + d)
    W = rng.standard_normal((rows, d)).astype(np.float32)
    A = (0.1 + rng.random((rows, d))).astype(np.float32)
    b = rng.standard_normal(rows).astype(np.float32)
    Ab = (0.1 + rng.random(rows)).astype(np.float32)
    touched = rng.random(rows) < frac
    G = np.where(touched[:, None], rng.standard_normal((rows, d)), 0.0).astype(np.float32)
    G[touched, ::3] = 0.0                                  # zeros inside a live row are fine
    Gb = np.where(touched, rng.standard_normal(rows), 0.0).astype(np.float32)
    if rows > 10:
        Gb[1] = 0.5                                        # a row whose only gradient is its bias cell
        G[1] = 0.0
    lr = torch.tensor([0.37], dtype=torch.float32, device=dev)
    w1, a1, b1, ab1 = _t(dev, W), _t(dev, A), _t(dev, b), _t(dev, Ab)
    ops.adagrad_dense(w1, a1, _t(dev, G), lr)
    ops.adagrad_dense(b1, ab1, _t(dev, Gb), lr)
    w2, a2, b2, ab2, g2, gb2 = _t(dev, W), _t(dev, A), _t(dev, b), _t(dev, Ab), _t(dev, G), _t(dev, Gb)
    ops.adagrad_rows_nonzero(w2, a2, b2, ab2, g2, gb2, lr)
    torch.cuda.synchronize()
    assert torch.equal(w1, w2) and torch.equal(a1, a2) and torch.equal(b1, b2) and torch.equal(ab1, ab2)
    assert not bool(g2.any()) and not bool(gb2.any())
