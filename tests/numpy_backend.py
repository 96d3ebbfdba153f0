"""TEST DOUBLE: numpy implementation of the compute-stage interface of
arx.dist.ShardedHMF (the product backend is HipBackend / libarx.so).  It lets the
world_size-2 gloo tests check the sharding + exchange logic on CPU against the
single-process oracle.  Lives in tests/ on purpose: the package has no CPU path."""
import numpy as np
import torch

KEY_NONE = 0x7FFFFFFF


def _n(t):
    return t.numpy()


class NumpyBackend(object):
    def gather_rows(self, E, bias, rows, out, bias_out, scale=1.0):
        r = _n(rows).astype(np.int64)
        _n(out)[...] = scale * _n(E)[r]
        if bias_out is not None:
            _n(bias_out)[...] = scale * _n(bias)[r]

    def gather_bags(self, E, bias, vals, starts, lens, ids, out, bias_out, scale=1.0, accumulate=False):
        e, b, v, st, ln = _n(E), _n(bias), _n(vals), _n(starts), _n(lens)
        o, ob = _n(out), _n(bias_out)
        for r, i in enumerate(_n(ids).astype(np.int64)):
            tok = v[st[i]:st[i] + ln[i]].astype(np.int64)
            row = scale * e[tok].astype(np.float64).sum(0) / float(ln[i])
            bb = scale * b[tok].astype(np.float64).sum() / float(ln[i])
            o[r] = (o[r] + row) if accumulate else row
            ob[r] = (ob[r] + bb) if accumulate else bb

    def bags_adagrad(self, E, acc, bias, bias_acc, vals, starts, lens, sites, G, Gb, lr):
        """sites: [(entity ids, row_base, coef)]; tokens >= E.shape[0] are dropped."""
        e, a, b, ba = _n(E), _n(acc), _n(bias), _n(bias_acc)
        v, st, ln = _n(vals), _n(starts), _n(lens)
        g_all, gb_all = _n(G).astype(np.float64), _n(Gb).astype(np.float64)
        g = np.zeros(e.shape, dtype=np.float64)
        gb = np.zeros(e.shape[0], dtype=np.float64)
        touched = np.zeros(e.shape[0], dtype=bool)
        for ids, base, coef in sites:
            for j, i in enumerate(_n(ids).astype(np.int64)):
                for t in v[st[i]:st[i] + ln[i]].astype(np.int64):
                    if 0 <= t < e.shape[0]:
                        g[t] += coef / float(ln[i]) * g_all[base + j][:e.shape[1]]
                        gb[t] += coef / float(ln[i]) * gb_all[base + j]
                        touched[t] = True
        lrv = float(_n(lr)[0])
        rows = np.nonzero(touched)[0]
        a[rows] = a[rows] + g[rows] ** 2
        e[rows] = e[rows] - lrv * g[rows] / np.sqrt(a[rows])
        ba[rows] = ba[rows] + gb[rows] ** 2
        b[rows] = b[rows] - lrv * gb[rows] / np.sqrt(ba[rows])

    def bags_grad_dense(self, D, Db, vals, starts, lens, sites, G, Gb):
        e, b = _n(D), _n(Db)
        v, st, ln = _n(vals), _n(starts), _n(lens)
        g_all, gb_all = _n(G).astype(np.float64), _n(Gb).astype(np.float64)
        g = np.zeros(e.shape, dtype=np.float64)
        gb = np.zeros(e.shape[0], dtype=np.float64)
        for ids, base, coef in sites:
            for j, i in enumerate(_n(ids).astype(np.int64)):
                for t in v[st[i]:st[i] + ln[i]].astype(np.int64):
                    if 0 <= t < e.shape[0]:
                        g[t] += coef / float(ln[i]) * g_all[base + j][:e.shape[1]]
                        gb[t] += coef / float(ln[i]) * gb_all[base + j]
        e[...] = (e + g).astype(np.float32)
        b[...] = (b + gb).astype(np.float32)

    def adagrad_dense(self, w, acc, g, lr):
        ww, aa, gg = _n(w), _n(acc), _n(g).astype(np.float64)
        a2 = aa.astype(np.float64) + gg * gg
        ww[...] = (ww - float(_n(lr)[0]) * gg / np.sqrt(a2)).astype(np.float32)
        aa[...] = a2.astype(np.float32)

    def adagrad_rows_nonzero(self, W, acc, bias, bias_acc, G, Gb, lr):
        g, gb = _n(G), _n(Gb)
        nz = (g != 0).any(axis=1) | (gb != 0)
        rows = np.nonzero(nz)[0]
        lrv = float(_n(lr)[0])
        w, a, b, ba = _n(W), _n(acc), _n(bias), _n(bias_acc)
        g64, gb64 = g[rows].astype(np.float64), gb[rows].astype(np.float64)
        a2 = a[rows].astype(np.float64) + g64 * g64
        w[rows] = (w[rows] - lrv * g64 / np.sqrt(a2)).astype(np.float32)
        a[rows] = a2.astype(np.float32)
        b2 = ba[rows].astype(np.float64) + gb64 * gb64
        b[rows] = (b[rows] - lrv * gb64 / np.sqrt(b2)).astype(np.float32)
        ba[rows] = b2.astype(np.float32)
        g[rows] = 0
        gb[rows] = 0

    def fill_zero(self, t):
        _n(t)[...] = 0

    def take_i32(self, table, idx, out, fill):
        i = _n(idx).astype(np.int64)
        _n(out)[...] = np.where(i >= 0, _n(table)[np.maximum(i, 0)], fill).astype(np.int32)

    def gather_rows_packed(self, E, bias, rows, out):
        r = _n(rows).astype(np.int64)
        d = _n(E).shape[1]
        _n(out)[:, :d] = _n(E)[r]
        _n(out)[:, d] = _n(bias)[r]

    def gemm(self, A, B, C, transA=False, transB=False, beta=0.0, col_bias=None, a_rowsum=None):
        a = _n(A).astype(np.float64)
        b = _n(B).astype(np.float64)
        a = a.T if transA else a
        b = b.T if transB else b
        c = a @ b
        if beta != 0.0:
            c = c + beta * _n(C)
        if col_bias is not None:
            c = c + _n(col_bias)[None, :]
        _n(C)[...] = c.astype(np.float32)
        if a_rowsum is not None:
            _n(a_rowsum)[...] = a.sum(1).astype(np.float32)

    def dot_score(self, U, T, tb, out):
        _n(out)[...] = ((_n(U).astype(np.float64) * _n(T)).sum(1) + _n(tb)).astype(np.float32)

    def dot_score_bwd(self, U, T, ds, dU, acc, dT):
        g = _n(ds)[:, None]
        if acc:
            _n(dU)[...] += g * _n(T)
        else:
            _n(dU)[...] = g * _n(T)
        if dT is not None:
            _n(dT)[...] = g * _n(U)

    def copy_strided(self, src, dst):
        _n(dst)[...] = _n(src)

    def copy_2d(self, src, dst):
        dst.copy_(src)

    def transpose(self, src, dst):
        _n(dst)[...] = _n(src).T

    def gather_rows_wide(self, src, rows, dst):
        r = _n(rows).astype(np.int64)
        ok = (r >= 0) & (r < _n(src).shape[0])
        out = np.zeros(_n(dst).shape, dtype=np.float32)
        out[ok] = _n(src)[r[ok]]
        _n(dst)[...] = out

    def add_2d(self, src, dst):
        _n(dst)[...] += _n(src)

    def shard_route(self, ids, world, rank, zero_row, rows_out, keys_out):
        i = _n(ids).astype(np.int64)
        own = (i % world) == rank
        if rows_out is not None:
            _n(rows_out)[...] = np.where(own, i // world, zero_row).astype(np.int32)
        if keys_out is not None:
            _n(keys_out)[...] = np.where(own, i // world, KEY_NONE).astype(np.int32)

    def pool_blocks(self, ids, world, rank, zero_row, cap, counts, gidx=None, my_slots=None, pool_rows=None):
        # counts: world + 1 cells, [world] = negative ids (no owner: gidx -1, no row of any block) -- include/arx.h
        i = _n(ids).astype(np.int64)
        S = i.shape[0]
        live = i >= 0
        owner = np.where(live, i % world, -1)
        c = _n(counts)
        c[:world] = np.bincount(owner[live], minlength=world).astype(np.int32)
        c[world] = int((~live).sum())
        if cap == 0:
            return
        pos = np.zeros(S, np.int64)
        for g in range(world):
            sel = np.nonzero(owner == g)[0]
            pos[sel] = np.arange(len(sel))
        _n(gidx)[...] = np.where(live, owner * cap + pos, -1).astype(np.int32)
        mine = np.nonzero(owner == rank)[0]
        ms, pr = _n(my_slots), _n(pool_rows)
        ms[...] = S
        pr[...] = zero_row
        ms[:len(mine)] = mine.astype(np.int32)
        pr[:len(mine)] = (i[mine] // world).astype(np.int32)

    def loss_mw_pos(self, logits, t, urows, ptr, items, i2s, bl, dl, dt, gscale):
        x = _n(logits).astype(np.float64)
        tt = _n(t).astype(np.float64)
        B, S = x.shape
        mask = np.ones((B, S), dtype=bool)
        p, it, m = _n(ptr), _n(items), _n(i2s)
        for r, u in enumerate(_n(urows)):
            for v in it[p[u]:p[u + 1]]:
                j = m[v]
                if j >= 0:
                    mask[r, j] = False
        v = x - tt[:, None] + 1
        act = mask & (v > 0)
        s = np.where(act, v, 0).sum(1)
        _n(bl)[...] = np.log(1 + s)
        g = gscale / (1 + s)
        d = act * g[:, None]
        _n(dl)[...] = d
        _n(dt)[...] = -d.sum(1)

    def loss_mw_fused_pos(self, logits, U, T, tb, urows, ptr, items, i2s, bl, dl, t_out, dt, dU, dT, gscale):
        """arx_loss_mw_fused_pos: target score + loss + dT = dt*U, dU = dt*T (written)."""
        self.dot_score(U, T, tb, t_out)
        tmp = torch.zeros(int(t_out.shape[0]), dtype=torch.float32)
        self.loss_mw_pos(logits, t_out, urows, ptr, items, i2s, bl, dl, tmp, gscale)
        dt.copy_(tmp)
        _n(dU)[...] = 0.0
        self.dot_score_bwd(U, T, tmp, dU, True, dT)

    def sum_scaled(self, x, scale, out):
        _n(out)[...] = _n(x).astype(np.float64).sum() * scale

    def sparse_adagrad(self, E, acc, bias, bias_acc, keys, G, Gb, lr):
        k = _n(keys).astype(np.int64)
        ok = k != KEY_NONE
        e, a = _n(E), _n(acc)
        g = np.zeros(e.shape, dtype=np.float64)
        np.add.at(g, k[ok], _n(G)[ok].astype(np.float64))
        rows = np.unique(k[ok])
        lrv = float(_n(lr)[0])
        a[rows] = a[rows] + g[rows] ** 2
        e[rows] = e[rows] - lrv * g[rows] / np.sqrt(a[rows])
        if bias is not None:
            b, ba = _n(bias), _n(bias_acc)
            gb = np.zeros(b.shape, dtype=np.float64)
            np.add.at(gb, k[ok], _n(Gb)[ok].astype(np.float64))
            ba[rows] = ba[rows] + gb[rows] ** 2
            b[rows] = b[rows] - lrv * gb[rows] / np.sqrt(ba[rows])

    def sparse_adagrad_multi(self, tables, sites, G, Gb, lr):
        """tables: [(E, acc, bias|None, bias_acc|None)]; sites: [(table, local_rows, row_base[, coef])]."""
        g_all, gb_all = _n(G).astype(np.float64), _n(Gb).astype(np.float64)
        lrv = float(_n(lr)[0])
        for t, (E, acc, bias, bacc) in enumerate(tables):
            e, a = _n(E), _n(acc)
            g = np.zeros(e.shape, dtype=np.float64)
            gb = np.zeros(e.shape[0], dtype=np.float64)
            touched = []
            for site in sites:
                tt, rows, base = site[0], site[1], site[2]
                coef = site[3] if len(site) > 3 else 1.0
                if tt != t:
                    continue
                k = _n(rows).astype(np.int64)
                src = base + np.arange(len(k))
                ok = (k >= 0) & (k < e.shape[0])                 # KEY_NONE: rows of other owners
                k, src = k[ok], src[ok]
                np.add.at(g, k, coef * g_all[src][:, :e.shape[1]])
                np.add.at(gb, k, coef * gb_all[src])
                touched.append(k)
            if not touched:
                continue
            rows = np.unique(np.concatenate(touched))
            a[rows] = a[rows] + g[rows] ** 2
            e[rows] = e[rows] - lrv * g[rows] / np.sqrt(a[rows])
            if bias is not None:
                b, ba = _n(bias), _n(bacc)
                ba[rows] = ba[rows] + gb[rows] ** 2
                b[rows] = b[rows] - lrv * gb[rows] / np.sqrt(ba[rows])

    def slot_map_set(self, m, ids, clear):
        i = _n(ids).astype(np.int64)
        if clear:
            _n(m)[i] = -1
        else:
            for s, v in enumerate(i):
                _n(m)[v] = s

    def copy_i32(self, src, dst):
        dst.copy_(src)
