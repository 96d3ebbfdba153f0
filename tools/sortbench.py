import sys, os
sys.path.insert(0, '/root/repo/a-recsys_amd'); sys.path.insert(0, '/root/repo')
import torch, numpy as np
from arx import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
d, Vf = 128, 1000002
E = torch.randn(Vf, d, device=dev); acc = torch.full_like(E, 0.1)
for n in (4096, 5120):
    keys = torch.from_numpy(rng.integers(0, Vf, size=n).astype(np.int32)).to(dev)
    G = torch.randn(n, d, device=dev)
    lr = torch.tensor([0.1], device=dev)
    ws = ops.Workspace(dev)
    for _ in range(5): ops.sparse_adagrad(E, acc, None, None, keys, None, None, G, None, lr, ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): ops.sparse_adagrad(E, acc, None, None, keys, None, None, G, None, lr, ws)
    e1.record(); torch.cuda.synchronize()
    print('TPE', os.environ.get('ARX_RANK_TPE'), 'block' if os.environ.get('ARX_BLOCK_SORT') else 'rank', 'n', n, 'us/call', round(e0.elapsed_time(e1) / 200 * 1e3, 2))
