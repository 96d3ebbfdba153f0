"""Cycle stamps of one workgroup of the bf16-pipe bit-operand product (variant library built with
tools/build_variant.sh bx6_trace -DBX6_TRACE, VARIANT_FILES=gemm_bx6; ARX_LIB=<that .so> ARX_GEMM_BX6=1)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np, torch
from arx import ops, _lib
dev = torch.device('cuda:0')
M, N, K = 16384, 128, 1024
tn = bool(os.environ.get('TN'))
rng = np.random.default_rng(0)
A = rng.random((M, K)) < 0.45
words = np.ascontiguousarray(np.packbits(A.reshape(M, K // 32, 32), axis=2, bitorder='little').view(np.uint32).reshape(M, K // 32).T)
tb = torch.from_numpy(words.view(np.int32)).to(dev)
ws = ops.Workspace(dev)
if not tn:
    Bm = torch.randn(K, N, device=dev); Cm = torch.zeros(M, N, device=dev); rs = torch.rand(M, device=dev)
    run = lambda: ops.gemm_bits(tb, Bm, Cm, ws, beta=1.0, row_scale=rs)
else:
    B2 = torch.randn(M, N, device=dev); C2 = torch.empty(K, N, device=dev); g = torch.rand(M, device=dev); r2 = torch.empty(K, device=dev)
    run = lambda: ops.gemm_bits(tb, B2, C2, ws, transA=True, gvec=g, a_rowsum=r2)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = np.zeros(12 * 256, dtype=np.uint64)
_lib.lib.arx_bx6_trace_read.argtypes = [C.c_void_p]
assert _lib.lib.arx_bx6_trace_read(buf.ctypes.data) == 0
t0 = min(int(v) & ((1 << 56) - 1) for v in buf if v)
names = {1: 'compute: prologue done', 2: 'compute: past first barrier', 3: 'compute: stage issued', 4: 'compute: past barrier',
         5: 'loader: loop top', 6: 'loader: past barrier (loads issued before)', 7: 'loader: tile stored'}
for wv in (0, 4, 8):
    ev = [(int(v) & ((1 << 56) - 1), int(v) >> 56) for v in buf[wv * 256:(wv + 1) * 256] if v]
    print("wave", wv)
    prev = None
    for t, k in ev[:26]:
        print("   %8d  (+%6d)  %s" % (t - t0, (t - prev) if prev else 0, names[k]))
        prev = t
