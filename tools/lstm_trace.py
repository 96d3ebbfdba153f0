"""Cycle stamps of one wave of k_lstm_fwd_r4 (a -DLSTM_TRACE style variant library: see DESIGN 6 "C4")."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))
import numpy as np, torch
from arx import ops, _lib
L, B, din, h = 50, 1024, 64, 64
dev = torch.device('cuda', 0)
x = torch.randn(L * B, din, device=dev) * 0.3
W = torch.randn(din + h, 4 * h, device=dev) * 0.1
b = torch.zeros(4 * h, device=dev)
hs, cs = torch.empty(L * B, h, device=dev), torch.empty(L * B, h, device=dev)
gates = torch.empty(L * B, 4 * h, device=dev)
for _ in range(3):
    ops.lstm_fwd(x, W, b, L, B, din, h, 1.0, hs, cs, gates)
if os.environ.get('LT_BWD'):
    dhs = torch.randn(L * B, h, device=dev) * 0.01
    dz = torch.empty(L * B, 4 * h, device=dev)
    for _ in range(3):
        ops.lstm_bwd(W, hs, cs, gates, dhs, L, B, din, h, dz)
torch.cuda.synchronize()
buf = np.zeros(4096, dtype=np.uint64)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.arx_lstm_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
ev = (buf >> np.uint64(56)).astype(np.int64); t = (buf & np.uint64((1 << 56) - 1)).astype(np.int64)
n = int((buf != 0).sum())
d = {}
for k in range(1, n):
    d.setdefault((ev[k - 1], ev[k]), []).append(t[k] - t[k - 1])
for key in sorted(d):
    v = np.array(d[key][2:])
    print("stamp %d -> %d: median %d cycles (n=%d)" % (key[0], key[1], np.median(v), len(v)))
