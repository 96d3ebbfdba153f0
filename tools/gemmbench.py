import sys, os
sys.path.insert(0, '/root/repo/a-recsys_amd'); sys.path.insert(0, '/root/repo')
import torch
from arx import ops
dev = torch.device('cuda:0')
ws = ops.Workspace(dev)
def t(fn, it=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
B, S, d = int(os.environ.get('GB_B', 4096)), 1024, 128
U = torch.randn(B, d, device=dev); I = torch.randn(S, d, device=dev); L = torch.empty(B, S, device=dev)
dL = torch.randn(B, S, device=dev); dU = torch.empty(B, d, device=dev); dI = torch.empty(S, d, device=dev)
tag = 'B=%d bm=%s splits=%s' % (B, os.environ.get('ARX_DMA_BM'), os.environ.get('ARX_DMA_SPLITS'))
print(tag, 'logits %.1f us' % t(lambda: ops.gemm(U, I, L, ws, transB=True)),
      'dU %.1f us' % t(lambda: ops.gemm(dL, I, dU, ws)),
      'dI %.1f us' % t(lambda: ops.gemm(dL, U, dI, ws, transA=True)))
# dU and dI issued on two streams (do two co-resident workgroups per CU fill each other's stalls?)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ws2 = ops.Workspace(dev)
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ops.gemm(dL, I, dU, ws)
    with torch.cuda.stream(s2):
        ops.gemm(dL, U, dI, ws2, transA=True)
    cur.wait_stream(s1); cur.wait_stream(s2)
print(tag, 'dU || dI on two streams %.1f us' % t(both))
