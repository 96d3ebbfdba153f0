"""sq_norm_accum micro-benchmark: python tools/normbench.py (env ARX_NORM_BLOCKS)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import torch
from arx import ops
dev = torch.device('cuda', 0)
n, d = 51200, 64
x = torch.randn(n, d, device=dev)
rs = torch.rand(n, device=dev)
out = torch.zeros(1, device=dev)
for name, r in (('plain', None), ('row_scale', rs)):
    for _ in range(5):
        ops.sq_norm_accum(x, out, d=d, row_scale=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.sq_norm_accum(x, out, d=d, row_scale=r)
    e1.record()
    torch.cuda.synchronize()
    out.zero_()
    ops.sq_norm_accum(x, out, d=d, row_scale=r)
    ref = float((x.double() ** 2 * (1.0 if r is None else r.double()[:, None])).sum())
    print(name, 'blocks', os.environ.get('ARX_NORM_BLOCKS'), '%.2f us' % (e0.elapsed_time(e1) * 1000 / 200),
          'rel err %.2e' % (abs(float(out.item()) - ref) / ref))
