"""Pool redraw cost: DeviceSampler.sample(S) over a 1 M-item population with p ~ count^0.5.
usage: python tools/samplerbench.py [n_items]"""
import sys, os
sys.path.insert(0, '/root/repo/a-recsys_amd'); sys.path.insert(0, '/root/repo')
import torch
from arx.utils.prepare_train import DeviceSampler
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = torch.Generator(device=dev); g.manual_seed(0)
w = (torch.rand(n, device=dev, generator=g) ** 4 + 1e-6)
w = (w / w.sum()).float()
s = DeviceSampler(torch.arange(n, dtype=torch.int32, device=dev), w, device=dev, seed=1)
out = torch.empty(1024, dtype=torch.int32, device=dev)
for _ in range(3): s.sample(1024, out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): s.sample(1024, out)
e1.record(); torch.cuda.synchronize()
print('n=%d  sample(1024): %.1f us  (cap in use: %s)' % (n, e0.elapsed_time(e1) / 20 * 1e3, s._cap_for(1024) > 0))
