"""K1 microbenchmark: multi-hot gather + segment-mean over tables of growing size (in L2 ->
in the 256 MB Infinity Cache -> past it), uniform and Zipf token draws.

usage: python tools/gatherbench.py [bags] [tokens_per_bag]       (run on the GPU box)
Prints event-timed us per launch and algorithmic GB/s (516 B/token + 520 B/bag at d=128,
SURVEY 8(d)); run under rocprofv3 --pmc FETCH_SIZE for the counter view.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np
import torch

from arx import ops


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device('cuda:0')
    d = 128
    rng = np.random.default_rng(0)
    Vs = [int(v) for v in os.environ['GB_V'].split(',')] if os.environ.get('GB_V') else \
        (8192, 32768, 65536, 100002, 1000002, 4000000)
    for V in Vs:
        E = torch.randn(V, d, device=dev)
        for dist in ('uniform', 'zipf1.0'):
            n_items = nb
            lens = np.full(n_items + 1, L, dtype=np.int32)
            starts = np.zeros(n_items + 2, dtype=np.int32)
            starts[1:] = np.cumsum(lens)
            tot = int(starts[-1])
            if dist == 'uniform':
                vals = rng.integers(0, V, tot).astype(np.int32)
            else:
                p = 1.0 / np.arange(1, V + 1)
                vals = rng.permutation(V)[rng.choice(V, size=tot, p=p / p.sum())].astype(np.int32)
            tv = torch.from_numpy(vals).to(dev)
            tst = torch.from_numpy(starts).to(dev)
            tl = torch.from_numpy(lens).to(dev)
            ids = torch.arange(nb, dtype=torch.int32, device=dev)
            out = torch.empty(nb, d, device=dev)

            def call():
                ops.gather_mulhot_mean(E, None, tv, tst, tl, ids, out)
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 30
            e0.record()
            for _ in range(it):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / it * 1e3
            by = nb * L * (4 * d + 4) + nb * (4 * d + 8)
            print('K1 V=%8d (%6.1f MB) %-8s bags=%d x %d: %7.1f us  %7.0f GB/s algorithmic' %
                  (V, V * d * 4 / 1e6, dist, nb, L, us, by / us / 1e3), flush=True)
        del E


if __name__ == '__main__':
    main()
