"""dx = dz . W_x^T of the C4 LSTM backward: arx_gemm_bt_bx6 (six-term bf16 tiles, W_x as it lies) against the f32-MFMA
LDS-DMA GEMM on W_x^T.  usage: python tools/dxbench.py [M N K]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import torch
from arx import ops


def t_us(fn, iters=50):
    for _ in range(iters):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


a = [int(x) for x in sys.argv[1:]]
M, N, K = a if len(a) == 3 else (51200, 64, 256)
dev = torch.device('cuda', 0)
dz = torch.randn(M, K, device=dev)
W = torch.randn(N + 64, K, device=dev)
Wt = W[:N].t().contiguous()
dx = torch.empty(M, N, device=dev)
ws = ops.Workspace(dev)
fl = 2.0 * M * N * K
t = t_us(lambda: ops.gemm_bt_bx6(dz, W[:N], dx))
print("bt_bx6   %7.1f us  %.0f TF f32-eq  (%.2f TB/s of dz)" % (t, fl / t / 1e6, M * K * 4 / t / 1e6))
t = t_us(lambda: ops.gemm(dz, Wt, dx, ws))
print("f32 mfma %7.1f us  %.0f TF" % (t, fl / t / 1e6))
