"""bf16x6 logits GEMM (arx_gemm_nt_bx6) vs the f32-MFMA kernel: error against f64, time."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np, torch
from arx import ops, _lib
lib = _lib.lib
lib.arx_gemm_nt_bx6_workspace_bytes.restype = C.c_size_t
lib.arx_gemm_nt_bx6_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
lib.arx_gemm_nt_bx6.restype = C.c_int
lib.arx_gemm_nt_bx6.argtypes = [C.c_int64] * 3 + [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                  C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
dev = torch.device('cuda:0')
ws, ws2 = ops.Workspace(dev), ops.Workspace(dev)

def bx6(U, I, L, bias=None):
    M, K = U.shape; N = I.shape[0]
    p, n = ws2.get(lib.arx_gemm_nt_bx6_workspace_bytes(N, K))
    rc = lib.arx_gemm_nt_bx6(M, N, K, U.data_ptr(), U.stride(0), I.data_ptr(), I.stride(0),
                             bias.data_ptr() if bias is not None else None, L.data_ptr(), L.stride(0), p, n,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.arx_last_error()

def t(fn, it=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

for (M, N, K) in [(200, 256, 128), (16384, 1024, 128), (51200, 1024, 64)]:
    g = torch.Generator(device=dev); g.manual_seed(M)
    U = torch.randn(M, K, device=dev, generator=g) * torch.exp(2 * torch.randn(M, 1, device=dev, generator=g))
    I = torch.randn(N, K, device=dev, generator=g)
    I[:, 0] = torch.arange(N, device=dev) * 0.01          # asymmetric
    b = torch.randn(N, device=dev, generator=g)
    L0, L1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(U, I, L0, ws, transB=True, col_bias=b)
    bx6(U, I, L1, b)
    ref = U.double() @ I.double().T + b.double()
    scale = (U.double().abs() @ I.double().abs().T) + 1e-30
    e0 = ((L0.double() - ref).abs() / scale).max().item()
    e1 = ((L1.double() - ref).abs() / scale).max().item()
    r0 = ((L0.double() - ref).abs() / scale).mean().item()
    r1 = ((L1.double() - ref).abs() / scale).mean().item()
    print("M=%d N=%d K=%d  err/(|a||b|): f32-mfma max %.2e mean %.2e | bx6 max %.2e mean %.2e" % (M, N, K, e0, r0, e1, r1))
    if M >= 16384:
        print("   f32-mfma %.1f us   bx6 (split + gemm) %.1f us" % (t(lambda: ops.gemm(U, I, L0, ws, transB=True, col_bias=b)),
                                                                 t(lambda: bx6(U, I, L1, b))))
