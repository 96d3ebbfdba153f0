"""Full-vocabulary top-k of the recommend path (hmf/hmf_model.py StreamTopK), fused against chunked, on random rows.
usage: python tools/topk_bench.py [B V d k]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import torch
from arx import graph as G
from arx.hmf.hmf_model import StreamTopK

a = [int(x) for x in sys.argv[1:]]
B, V, d, k = (a + [4096, 1000000, 128, 100])[:4] if len(a) < 4 else a[:4]
dev = torch.device('cuda', 0)
rt = G.Runtime(dev)
g = torch.Generator(device=dev)
g.manual_seed(0)


class Leaf(G.Node):
    def __init__(self, shape, scale):
        super().__init__(rt, shape)
        self.value = torch.randn(shape, device=dev, generator=g) * scale
        self.bias_value = None


lat, pool = Leaf((B, d), 0.3), Leaf((V, d), 0.3)
pool.bias_value = torch.randn(V, device=dev, generator=g) * 0.1
res = {}
for mode in ('fused', 'chunked'):
    tk = StreamTopK(rt, lat, pool, k)
    tk.fused = mode == 'fused'
    for _ in range(2):
        tk.forward(False)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 3
    for _ in range(n):
        tk.forward(False)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    res[mode] = (ms, tk.indices.clone(), tk.value.clone(), int(tk.overflow.item()))
    tf = 2.0 * B * V * d / ms / 1e9
    print("%-8s %8.2f ms  %6.1f M rows x items/s  %5.1f TF f32  overflow %d" % (mode, ms, B * V / ms / 1e3, tf, res[mode][3]))
    del tk
print("identical indices:", bool(torch.equal(res['fused'][1], res['chunked'][1])),
      " identical values:", bool(torch.equal(res['fused'][2], res['chunked'][2])))
if os.environ.get("TOPK_CHECK"):
    nr = min(B, 256)
    ref = (lat.value[:nr].double() @ pool.value.double().t() + pool.bias_value.double()[None, :]).float()
    tv, ti = torch.topk(ref, k, dim=1)
    for mode in ('fused', 'chunked'):
        idx = res[mode][1][:nr].long()
        same = (idx == ti).all(1)
        # where they differ: is it a near-tie (f32 vs f64 rounding) or a real miss?
        gap = (torch.gather(ref, 1, idx) - tv).abs().max().item()
        print(mode, "rows equal to torch.topk(f64 scores):", int(same.sum()), "/", nr, " max |value gap|", gap)
    d_ = (res['fused'][1] != res['chunked'][1]).any(1)
    print("rows where fused != chunked:", int(d_.sum()), "of", B, " first:", d_.nonzero()[:5].flatten().tolist())
    r = int(d_.nonzero()[0]) if d_.any() else 0
    print("row", r, "fused  ", res['fused'][1][r, :8].tolist(), res['fused'][2][r, :4].tolist())
    print("row", r, "chunked", res['chunked'][1][r, :8].tolist(), res['chunked'][2][r, :4].tolist())
if os.environ.get("EVAL_BENCH"):
    # the full-vocabulary evaluation loss (graph.StreamEvalLoss): fused GEMM epilogue against chunks
    class Ids(G.Node):
        def __init__(self, n):
            super().__init__(rt, (n,))
            self.value = torch.randint(0, V, (n,), device=dev, dtype=torch.int32, generator=g)
    tgt = Ids(B)
    for kind in ('ce', 'warp'):
        out = {}
        for mode in ('fused', 'chunked'):
            ev = G.StreamEvalLoss(rt, kind, lat, pool, tgt)
            ev.fused = mode == 'fused'
            for _ in range(2):
                ev.forward(False)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                ev.forward(False)
            torch.cuda.synchronize()
            ms = (time.time() - t0) / 3 * 1e3
            out[mode] = (ms, ev.value.clone())
            del ev
        rel = ((out['fused'][1] - out['chunked'][1]).abs() / out['chunked'][1].abs().clamp_min(1e-6)).max().item()
        print("eval %-4s fused %8.2f ms  chunked %8.2f ms  (%.1fx)  max rel diff %.2e" % (
            kind, out['fused'][0], out['chunked'][0], out['chunked'][0] / out['fused'][0], rel))
if os.environ.get("TOPK_MATERIALISED"):
    # the materialising form (hmf_model.TopK over Prediction): one [B, V] GEMM + arx_topk
    from arx import ops
    lg = torch.empty((B, V), dtype=torch.float32, device=dev)
    tv, ti = torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B, k), dtype=torch.int32, device=dev)
    def mat():
        ops.gemm(lat.value, pool.value, lg, rt.ws, transB=True, col_bias=pool.bias_value)
        ops.topk(lg, k, tv, ti)
    for _ in range(2):
        mat()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        mat()
    torch.cuda.synchronize()
    print("materialised %8.2f ms   same indices as fused: %s" % ((time.time() - t0) / 3 * 1e3, bool(torch.equal(ti, res['fused'][1]))))
