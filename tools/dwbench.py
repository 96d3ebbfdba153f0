"""The LSTM cell's paired weight-gradient product alone (arx_gemm_f32_tn_pair: product + split-K reduce) at the C4
shape: HIP events over 200 launches (round 6: the yardstick of profiles/r06_dw_bx6.txt)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import torch
from arx import ops

L, B, din, h = 50, 1024, 64, 64
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
dz = torch.randn((L * B, 4 * h), device=dev, generator=g)
x = torch.randn((L * B, din), device=dev, generator=g)
hs = torch.randn((L * B, h), device=dev, generator=g)
Wg = torch.empty((din + h, 4 * h), device=dev)
db = torch.empty((4 * h,), device=dev)
ws = ops.Workspace(dev)
for _ in range(5):
    ops.gemm_tn_pair(dz, x, hs, B, Wg, ws, a_rowsum=db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    ops.gemm_tn_pair(dz, x, hs, B, Wg, ws, a_rowsum=db)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
fl = 2.0 * (4 * h) * (din + h) * L * B
print("dW pair (product + split-K reduce): %.1f us  %.1f TF f32-equivalent" % (us, fl / us / 1e6))
