"""K7 microbenchmark: sparse Adagrad (sort + apply) on synthetic key streams.

usage: python tools/scatterbench.py [config ...]     (run on the GPU box)
configs: uniq5k  zipf5k  uniq66k  zipf66k  mulhot100k  mulhot1m
Prints the event-timed duration of the whole entry point (sort + apply) per config; run it
under `rocprofv3 --kernel-trace --stats` with ONE config for the per-kernel split.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np
import torch

from arx import ops

CONFIGS = {
    # name: (n contributions, table rows, source rows, distribution)
    'uniq5k': (5120, 1000000, 5120, 'uniform'),
    'zipf5k': (5120, 1000000, 5120, 'zipf_items'),
    'uniq66k': (66560, 1000000, 66560, 'uniform'),
    'zipf66k': (66560, 1000000, 66560, 'zipf_items'),
    'mulhot100k': (102400, 100002, 5120, 'zipf_tokens'),
    'mulhot1m': (1331200, 100002, 66560, 'zipf_tokens'),
}


def make(name, d=128, seed=0):
    n, V, m, dist = CONFIGS[name]
    rng = np.random.default_rng(seed)
    if dist == 'uniform':
        keys = rng.integers(0, V, n)
    elif dist == 'zipf_items':
        p = 1.0 / np.arange(1, V + 1) ** 1.05
        keys = rng.permutation(V)[rng.choice(V, size=n, p=p / p.sum())]
    else:
        p = 1.0 / np.arange(1, V + 1)
        keys = rng.choice(V, size=n, p=p / p.sum())
    src = np.sort(rng.integers(0, m, n))       # bag-major order, like csr_expand emits
    return keys.astype(np.int32), src.astype(np.int32), V, m


def main():
    names = sys.argv[1:] or list(CONFIGS)
    dev = torch.device('cuda:0')
    d = 128
    for name in names:
        keys, src, V, m = make(name, d)
        n = len(keys)
        E = torch.randn(V, d, device=dev)
        acc = torch.full((V, d), 0.1, device=dev)
        G = torch.randn(m, d, device=dev)
        tk = torch.tensor(keys, device=dev)
        ts = torch.tensor(src, device=dev)
        tc = torch.rand(n, device=dev)
        lr = torch.tensor([0.1], device=dev)
        cnt = torch.zeros(V, dtype=torch.int32, device=dev)
        ws = ops.Workspace(dev)
        uniq = len(np.unique(keys))
        for mode, aux in (('ticket', cnt), ('list', None)):
            def call():
                ops.sparse_adagrad(E, acc, None, None, tk, ts, tc, G, None, lr, ws, n=n, aux_cnt=aux)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            g = ops.CapturedGraph()          # time graph replays: no host launch gaps
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g.begin()
                try:
                    call()
                finally:
                    g.end()
            torch.cuda.current_stream().wait_stream(side)
            g.launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20
            e0.record()
            for _ in range(iters):
                g.launch()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            alg = uniq * (16 * d + 4) + n * 4 + m * 4 * d
            print('%-11s %-6s n=%8d uniq=%7d  %8.1f us  alg %6.1f MB -> %6.0f GB/s' %
                  (name, mode, n, uniq, us, alg / 1e6, alg / us / 1e3), flush=True)
        assert int(cnt.abs().sum().item()) == 0


if __name__ == '__main__':
    main()
