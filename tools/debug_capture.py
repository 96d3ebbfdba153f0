"""Debug aid: sparse Adagrad inside a hipGraph capture at several n (run on the GPU box)."""
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np
import torch

from arx import ops


def run(n, rows=5000, d=32, capture=True, ticket=False):
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(n)
    E = torch.randn(rows, d, device=dev)
    acc = torch.full((rows, d), 0.1, device=dev)
    keys = torch.tensor(rng.integers(0, rows, n), dtype=torch.int32, device=dev)
    src = torch.tensor(rng.integers(0, 64, n), dtype=torch.int32, device=dev)
    coef = torch.ones(n, device=dev)
    G = torch.randn(64, d, device=dev)
    lr = torch.tensor([0.1], device=dev)
    ws = ops.Workspace(dev)
    aux = torch.zeros(rows, dtype=torch.int32, device=dev) if ticket else None
    ops.sparse_adagrad(E, acc, None, None, keys, src, coef, G, None, lr, ws, n=n, aux_cnt=aux)
    torch.cuda.synchronize()
    if capture:
        g = ops.CapturedGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g.begin()
            try:
                ops.sparse_adagrad(E, acc, None, None, keys, src, coef, G, None, lr, ws, n=n,
                                   aux_cnt=aux)
            finally:
                g.end()
        torch.cuda.current_stream().wait_stream(side)
        g.launch()
        torch.cuda.synchronize()
    print('ok n=%d capture=%s ticket=%s' % (n, capture, ticket), flush=True)


if __name__ == '__main__':
    for n in (5000, 13000, 300000):
        for t in (False, True):
            run(n, capture=True, ticket=t)
