"""A/B timing of the two backward scorer products: fp32 dlogits operand (arx_gemm_f32) vs the bit
operand (arx_gemm_bits_f32) at the C2 shape.  usage: python tools/bitsbench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np
import torch

from arx import ops


def t_us(fn, it=50):
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    dev = torch.device('cuda:0')
    B, S, d = 16384, 1024, 128
    A = (torch.rand(B, S, device=dev) < 0.3)
    dl = A.float() * 0.01
    P = torch.randn(S, d, device=dev)
    U = torch.randn(B, d, device=dev)
    g = torch.rand(B, device=dev)
    words = torch.from_numpy(np.ascontiguousarray(
        np.packbits(A.cpu().numpy().reshape(B, S // 32, 32), axis=2, bitorder='little').view(np.uint32)
        .reshape(B, S // 32).T).view(np.int32)).to(dev)
    ws = ops.Workspace(dev)
    dU = torch.zeros(B, d, device=dev)
    dI = torch.zeros(S, d, device=dev)
    db = torch.zeros(S, device=dev)
    print('dU fp32  %.1f us' % t_us(lambda: ops.gemm(dl, P, dU, ws, beta=1.0)))
    print('dU bits  %.1f us' % t_us(lambda: ops.gemm_bits(words, P, dU, ws, beta=1.0, row_scale=g)))
    print('dI fp32  %.1f us' % t_us(lambda: ops.gemm(dl, U, dI, ws, transA=True, a_rowsum=db)))
    print('dI bits  %.1f us' % t_us(lambda: ops.gemm_bits(words, U, dI, ws, transA=True, gvec=g, a_rowsum=db)))


if __name__ == '__main__':
    main()
