"""Cycle stamps of one compute wave of k_sc_hinge (build: VARIANT_FILES=scorer tools/build_variant.sh sc_trace
-fno-slp-vectorize -DSC_TRACE; run: ARX_LIB=a-recsys_amd/arx/lib/exp/sc_trace.so python tools/sc_trace.py)."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import numpy as np
import torch
from arx import ops, _lib

B, S, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 16384), 1024, (int(sys.argv[2]) if len(sys.argv) > 2 else 128)
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
U, T = torch.randn(B, d, device=dev, generator=g) * 0.3, torch.randn(B, d, device=dev, generator=g) * 0.3
P = torch.randn(S, d, device=dev, generator=g) * 0.3
pb, tb = torch.zeros(S, device=dev), torch.zeros(B, device=dev)
users = torch.zeros(B, dtype=torch.int32, device=dev)
ptr = torch.zeros(3, dtype=torch.int32, device=dev)
items = torch.zeros(1, dtype=torch.int32, device=dev)
i2s = torch.full((S + 1,), -1, dtype=torch.int32, device=dev)
bl, ts, dts = (torch.empty(B, device=dev) for _ in range(3))
dU, dT = torch.empty(B, d, device=dev), torch.empty(B, d, device=dev)
sc = ops.MwScorer(B, S, d, dev)
for _ in range(20):
    sc.fwd(U, P, pb, T, tb, users, ptr, items, i2s, bl, ts, dts, dU, dT, 1.0 / B)
torch.cuda.synchronize()
sc.fwd(U, P, pb, T, tb, users, ptr, items, i2s, bl, ts, dts, dU, dT, 1.0 / B, phases=int(os.environ.get("SC_PHASE", "2")))
torch.cuda.synchronize()
buf = np.zeros(4096, dtype=np.uint64)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.arx_sc_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
ev = (buf >> np.uint64(56)).astype(np.int64)
t = (buf & np.uint64((1 << 56) - 1)).astype(np.int64)
n = int((buf != 0).sum())
print("events", n)
real = {int(ev[k]): int(t[k]) for k in range(n) if ev[k] >= 120}
keep = [k for k in range(n) if ev[k] < 120]
if len(real) == 2:
    dt_us = (real[121] - real[120]) / 100.0
    cyc = int(t[keep[-1]] - t[keep[0]])
    print("wave lifetime %.2f us by the 100 MHz counter, %d cycle stamps -> %.2f GHz" % (dt_us, cyc, cyc / dt_us / 1e3))
ev, t, n = ev[keep], t[keep], len(keep)
prev = t[0]
line = []
for k in range(n):
    line.append("%d:%d" % (ev[k], t[k] - prev))
    prev = t[k]
    if ev[k] == 21 or ev[k] == 101:
        print(" ".join(line))
        line = []
print(" ".join(line))
print("total cycles", t[n - 1] - t[0])
