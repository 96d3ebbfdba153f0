"""Times the pieces of the sharded model's pool redraw (world 1, 100 M items): the device race, the
draw_global_pool wrapper, set_pool, and a step right after a redraw against a steady-state step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'a-recsys_amd'))
import numpy as np, torch
import torch.distributed as dist
from arx import dist as D, ops
from arx.utils.prepare_train import DeviceSampler

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
dist.init_process_group('gloo', rank=0, world_size=1)
dev = torch.device('cuda', 0)
n_items = int(os.environ.get('N_ITEMS', 100000000)); S = 1024; B = 16384
model = D.ShardedHMF(1000000, n_items, 128, B, S, 0.1, 0, 1, dev, seed=0)
gen = torch.Generator(device=dev); gen.manual_seed(7)
nu, n_pos = model.nu_loc, 20
ptr = (torch.arange(nu + 2, device=dev, dtype=torch.int64) * n_pos).clamp(max=nu * n_pos).to(torch.int32)
pos_items = D._zipf_items(nu * n_pos, n_items, gen, dev)
model.set_positives(ptr, pos_items)
batches = []
for _ in range(8):
    lu = torch.randint(0, nu, (B,), device=dev, generator=gen)
    k = torch.randint(0, n_pos, (B,), device=dev, generator=gen)
    batches.append(model.prepare_route(lu.to(torch.int32), pos_items[lu * n_pos + k]))
cnt = torch.zeros(n_items, dtype=torch.int32, device=dev)
ops.item_frequency(pos_items, n_items, cnt)
wts = (cnt.to(torch.float64) / cnt.sum(dtype=torch.int64).to(torch.float64)).pow(0.5).to(torch.float32)
sampler = DeviceSampler(torch.arange(n_items, device=dev, dtype=torch.int32), wts, device=dev, seed=1)

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3

print('sampler.sample            %.3f ms' % t(lambda: sampler.sample(S)))
print('sampler.sample_with_keys  %.3f ms' % t(lambda: sampler.sample_with_keys(S)))
print('draw_global_pool          %.3f ms' % t(lambda: D.draw_global_pool(sampler, S)))
ids = D.draw_global_pool(sampler, S)
print('set_pool                  %.3f ms' % t(lambda: model.set_pool(ids)))
model.set_pool(ids)
def steps(n, redraw_every=0):
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(n):
        if redraw_every and k % redraw_every == 0:
            model.set_pool(D.draw_global_pool(sampler, S))
        model.step(batches[k % 8])
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
steps(20)
print('step, no redraw           %.4f ms' % steps(100))
print('step, redraw every 50     %.4f ms' % steps(100, 50))
print('step, redraw every 10     %.4f ms' % steps(100, 10))
