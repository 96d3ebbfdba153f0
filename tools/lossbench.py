"""K6 stand-alone: the fused 'mw' loss at the C2/C3 shape with (a) the real positives chain
(20 positives per user, ~none of them in the pool), (b) empty positive lists -- the difference is what
the item2slot probes cost.  usage: python tools/lossbench.py [B]"""
import sys, os
sys.path.insert(0, '/root/repo/a-recsys_amd'); sys.path.insert(0, '/root/repo')
import torch
from arx import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
S, d, V, NU, NP = 1024, 128, 1000000, 1000000, 20
g = torch.Generator(device=dev); g.manual_seed(0)
logits = torch.randn(B, S, device=dev, generator=g)
U = torch.randn(B, d, device=dev, generator=g); T = torch.randn(B, d, device=dev, generator=g)
tb = torch.randn(B, device=dev, generator=g)
users = torch.randint(0, NU, (B,), device=dev, generator=g, dtype=torch.int32)
ptr = (torch.arange(NU + 1, device=dev, dtype=torch.int64) * NP).to(torch.int32)
items = torch.randint(0, V, (NU * NP,), device=dev, generator=g, dtype=torch.int32)
i2s = torch.full((V,), -1, dtype=torch.int32, device=dev)
pool = torch.randperm(V, device=dev, generator=g)[:S].to(torch.int32)
ops.slot_map_set(i2s, pool, clear=False)
ptr0 = torch.zeros(NU + 1, dtype=torch.int32, device=dev)
bl = torch.empty(B, device=dev); dl = torch.empty(B, S, device=dev); ts = torch.empty(B, device=dev)
dt = torch.empty(B, device=dev); dU = torch.empty(B, d, device=dev); dT = torch.empty(B, d, device=dev)
def t(fn, it=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, p in (('positives', ptr), ('no positives', ptr0)):
    us = t(lambda: ops.loss_mw_fused_pos(logits, U, T, tb, users, p, items, i2s, bl, dl, ts, dt, dU, dT, 1.0 / B))
    print('B=%d %-14s %.1f us  (%.2f TB/s of the 2 x [B,S] fp32 passes)' % (B, name, us, 2 * B * S * 4 / us / 1e6))
