"""A/B timing of the two backward scorer products: fp32 dlogits operand (arx_gemm_f32) vs the bit
operand (arx_gemm_bits_f32), and of the forward pair (logits GEMM + fused mw loss) vs the hinge-epilogue
GEMM (arx_mw_gemm_fused_fwd).  usage: python tools/bitsbench.py [B S d]   (default: the C2 shape;
51200 1024 64 = the C4 scorer over [L*B, S])"""
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
    B, S, d = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 1024, 128)
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
    # forward: logits GEMM + fused loss (target score inside) vs the hinge-epilogue GEMM
    NU, NP, V = 100000, 20, 1000000
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    T = torch.randn(B, d, device=dev); tb = torch.randn(B, device=dev); pb = torch.randn(S, device=dev)
    users = torch.randint(0, NU, (B,), device=dev, generator=gen, dtype=torch.int32)
    ptr = (torch.arange(NU + 1, device=dev, dtype=torch.int64) * NP).to(torch.int32)
    items = torch.randint(0, V, (NU * NP,), device=dev, generator=gen, dtype=torch.int32)
    i2s = torch.full((V,), -1, dtype=torch.int32, device=dev)
    ops.slot_map_set(i2s, torch.randperm(V, device=dev, generator=gen)[:S].to(torch.int32), clear=False)
    logits = torch.empty(B, S, device=dev); dlg = torch.empty(B, S, device=dev)
    bl = torch.empty(B, device=dev); ts = torch.empty(B, device=dev); dt = torch.empty(B, device=dev)
    dT = torch.empty(B, d, device=dev); dU2 = torch.empty(B, d, device=dev)
    t_g = t_us(lambda: ops.gemm(U, P, logits, ws, transB=True, col_bias=pb))
    t_l = t_us(lambda: ops.loss_mw_fused_pos(logits, U, T, tb, users, ptr, items, i2s, bl, dlg, ts, dt, dU2, dT, 1.0 / B))
    print('fwd fp32: logits GEMM %.1f us + fused loss %.1f us = %.1f us' % (t_g, t_l, t_g + t_l))
    bits = torch.empty(S // 32, B, dtype=torch.int32, device=dev)
    gv = torch.empty(B, device=dev); Ug = torch.empty(B, d, device=dev)
    t_f = t_us(lambda: ops.mw_gemm_fused_fwd(U, P, pb, T, tb, users, ptr, items, i2s, bl, ts, bits, gv, Ug, dt, dU2, dT,
                                             1.0 / B, ws))
    print('fwd hinge-epilogue GEMM (bits out) %.1f us' % t_f)


if __name__ == '__main__':
    main()
