"""Micro-benchmark of the fused 'mce' family (csrc/scorer.hip k_mc_flow): every launch group of the forward alone and
the pool-side backward, against the materialising path (logits GEMM + loss kernel + two GEMMs).
usage: python tools/mcebench.py [B S [mask_rows [d]]]   (d = 64 or 128)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import torch
from arx import ops


def t_us(fn, iters=30):
    for _ in range(iters):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    a = [int(x) for x in sys.argv[1:]]
    B, S = (a + [51200, 1024])[:2] if len(a) < 2 else a[:2]
    mrows = a[2] if len(a) > 2 else 0
    d = a[3] if len(a) > 3 else 64
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    U = torch.randn(B, d, device=dev, generator=g) * 0.3
    P = torch.randn(S, d, device=dev, generator=g) * 0.3
    T = torch.randn(B, d, device=dev, generator=g) * 0.3
    pb = torch.randn(S, device=dev, generator=g) * 0.1
    tb = torch.randn(B, device=dev, generator=g) * 0.1
    n_items, n_users = 1000000, 100000
    pool = torch.randperm(n_items, device=dev, generator=g)[:S].to(torch.int32)
    i2s = torch.full((n_items + 1,), -1, dtype=torch.int32, device=dev)
    i2s[pool.long()] = torch.arange(S, dtype=torch.int32, device=dev)
    ptr = (torch.arange(n_users + 2, device=dev) * 20).clamp(max=n_users * 20).to(torch.int32)
    items = torch.randint(0, n_items, (n_users * 20,), device=dev, generator=g).to(torch.int32)
    users = torch.randint(0, n_users, (mrows or B,), device=dev, generator=g).to(torch.int32)
    bl, ts, dts = (torch.empty(B, device=dev) for _ in range(3))
    dU, dT = torch.empty(B, d, device=dev), torch.empty(B, d, device=dev)
    dI, db = torch.empty(S, d, device=dev), torch.empty(S, device=dev)
    fl = 2.0 * B * S * d
    sc = ops.MceScorer(B, S, d, dev)

    def fwd(ph):
        sc.fwd(U, P, pb, T, tb, users, ptr, items, i2s, bl, ts, dts, dU, dT, 1.0 / B, mask_rows=mrows, phases=ph)
    fwd(7)
    print("B=%d S=%d d=%d  (2BSd = %.2f GFLOP f32-equivalent; a flow launch = 12 bf16 terms of it)" % (B, S, d, fl / 1e9))
    for name, ph in (("prep (+ masks)", 1), ("flow dU", 2), ("rows", 4), ("fwd (all)", 7)):
        t = t_us(lambda: fwd(ph))
        extra = "  %.0f TF bf16 = %.2f of 2500" % (12 * fl / t / 1e6, 12 * fl / t / 1e6 / 2500) if ph == 2 else ""
        print("  %-18s %7.1f us%s" % (name, t, extra))
    t = t_us(lambda: sc.bwd_dI(dI, db=db))
    print("  %-18s %7.1f us  (flow dI + reduce) %.0f TF bf16 = %.2f of 2500" % ("bwd dI", t, 12 * fl / t / 1e6, 12 * fl / t / 1e6 / 2500))
    if mrows:
        L = B // mrows
        dIs, dbs = torch.empty(L, S, d, device=dev), torch.empty(L, S, device=dev)
        t = t_us(lambda: sc.bwd_dI(dI, db=db, step_rows=mrows, dI_steps=dIs, db_steps=dbs))
        print("  %-18s %7.1f us" % ("bwd dI per step", t))
    ws = ops.Workspace(dev)
    logits, dl = torch.empty(B, S, device=dev), torch.empty(B, S, device=dev)
    t1 = t_us(lambda: ops.gemm(U, P, logits, ws, transB=True, col_bias=pb))
    t2 = t_us(lambda: ops.loss_mw_fused_pos(logits, U, T, tb, users, ptr, items, i2s, bl, dl, ts, dts, dU, dT, 1.0 / B,
                                            None, mrows, kind='mce'))
    t3 = t_us(lambda: ops.gemm(dl, P, dU, ws, beta=1.0))
    t4 = t_us(lambda: ops.gemm(dl, U, dI, ws, transA=True, a_rowsum=db))
    print("  materialising: logits GEMM %.1f + loss %.1f + dU %.1f + dI %.1f = %.1f us" % (t1, t2, t3, t4, t1 + t2 + t3 + t4))


if __name__ == "__main__":
    main()
