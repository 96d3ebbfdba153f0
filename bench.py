#!/usr/bin/env python
"""bench.py -- A-RecSys hot path on MI355X: training interactions/sec.

Workload (BASELINE.json configs[1], "C2"): synthetic 1M-item / 1M-user HMF,
dim 128, id-only attributes, WMRB sampled loss ('mw'), 1024 negatives shared
per step, pool redrawn every 50 steps (run_hmf.py:63), Adagrad.  One "step" is
one full pass of the hot path over one batch: user gather -> pool gather ->
scorer GEMM -> target score -> WMRB loss fwd+bwd -> backward GEMMs -> sparse
scatter + Adagrad on both tables.  Inputs are device-resident before timing.

  python bench.py --gpus N --steps K --warmup W      (N>1 via torch.distributed.run)

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel,
HIP-event timed here), "roofline_hbm" (the gather / scatter+Adagrad kernels
against HBM peak), "cpu_baseline" (the oracle's restatement of the reference's
TF1 CPU algorithm, bounded sample, host cores stated).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "a-recsys_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3    # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16384,
                    help="interactions per step per GPU (SURVEY 8(d) C2 throughput batches: 4096, 16384)")
    ap.add_argument("--n-items", type=int, default=None,
                    help="item table rows (default: 1M on one GPU = configs[1]; 100M row-sharded for "
                         "--gpus N > 1 = configs[4])")
    ap.add_argument("--n-users", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--n-sampled", type=int, default=1024)
    ap.add_argument("--n-resample", type=int, default=50)
    ap.add_argument("--mulhot", action="store_true", help="C3: add a multi-hot item attribute")
    ap.add_argument("--zipf-items", type=float, default=1.05,
                    help="popularity exponent of the synthetic item draw (0 = uniform; experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def _evt_time_ms(fn, iters):
    """Average duration of fn() in ms, HIP events on the stream the kernels use."""
    # warm up as long as the timed loop: after the host-side pause before this call the shader
    # clock needs a few ms of load to be back at its sustained 2.4 GHz (tools/clockwatch.py) --
    # in the step's graph the kernels always run on a loaded chip
    for _ in range(max(3, iters)):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_rooflines(model, args):
    """Per-kernel timings of the step's main kernels with the step's own buffers."""
    from arx import ops, graph as G
    rt = model.rt
    plan = model._plan('train')
    nodes = {type(n).__name__ + str(i): n for i, n in enumerate(plan.order)}
    pred = [n for n in plan.order if isinstance(n, G.Prediction)][0]
    latent, pool = pred.inputs
    B, S, d = latent.shape[0], pool.shape[0], latent.shape[1]
    ws = rt.ws
    res = {}
    # --- GEMMs (MFMA bound) ---
    flops = 2.0 * B * S * d
    t = _evt_time_ms(lambda: ops.gemm(latent.value, pool.value, pred.value, ws, transB=True,
                                      col_bias=pool.bias_value), 50)
    res['gemm_logits_nt'] = dict(ms=t, flops=flops, tflops=flops / t / 1e9)
    dl = pred.grad
    gU = latent.alloc_grad()
    t = _evt_time_ms(lambda: ops.gemm(dl, pool.value, gU, ws), 50)
    res['gemm_dU_nn'] = dict(ms=t, flops=flops, tflops=flops / t / 1e9)
    gP = pool.alloc_grad()
    t = _evt_time_ms(lambda: ops.gemm(dl, latent.value, gP, ws, transA=True), 50)
    res['gemm_dI_tn'] = dict(ms=t, flops=flops, tflops=flops / t / 1e9)
    # --- gathers (HBM bound): algorithmic bytes per SURVEY 8(d) ---
    for n in plan.order:
        if isinstance(n, G.EntityEmbed):
            rows = n.shape[0]
            by = 0.0
            toks = 0.0
            for f in n.feats:
                if f.kind == 'cat':
                    by += rows * (4 * d + 4 + 4 + (4 if n.with_bias else 0))
                else:
                    lens = f.maps[2][n.inputs[0].value.long()].sum().item()
                    toks += lens
                    by += lens * (4 * d + 4) + rows * 8
            by += rows * 4 * d   # output write (once; accumulate re-reads stay in L2)
            t = _evt_time_ms(lambda n=n: n.forward(False), 50)
            res['gather_%s_%d' % (n.inputs[0].name, rows)] = dict(ms=t, bytes=by, gbs=by / t / 1e6,
                                                                  tokens=toks)
    # --- scatter + sparse Adagrad (HBM bound): 16d+4 per unique row + 4d per source row ---
    for table, sites, bufs, total in plan.tables:
        # this table's contributions of the LAST step, rebuilt the way the un-fused K7 pass builds
        # them (the step itself may have sorted several tables in one shared pass)
        for s_ in sites:
            ks = bufs['keys'][s_.key_off:s_.key_off + s_.cap]
            ss = bufs['src'][s_.key_off:s_.key_off + s_.cap]
            cs = bufs['coef'][s_.key_off:s_.key_off + s_.cap]
            if s_.kind == 'cat':
                ops.sparse_site_onehot(s_.maps[0], s_.ids_node.value, s_.node.row0, s_.coef, ks, ss, cs)
            else:
                ops.bag_expand_padded(s_.maps[0], s_.maps[1], s_.maps[2], s_.ids_node.value, s_.max_len,
                                      s_.node.row0, s_.coef, ks, ss, cs)
        keys = bufs['keys']
        valid = keys[keys != ops.KEY_NONE]
        uniq = int(torch.unique(valid).numel())
        nsrc = int(valid.numel())
        by = uniq * (16 * d + 4) + nsrc * (4 + 4 + 4) + sum(s.n for s in sites) * 4 * d
        node0 = sites[0].node
        snap = (table.E.clone(), table.acc.clone())
        use_bias = table.bias is not None
        t = _evt_time_ms(lambda: ops.sparse_adagrad(
            table.E, table.acc, table.bias if use_bias else None,
            table.bias_acc if use_bias else None, bufs['keys'], bufs['src'], bufs['coef'],
            node0.arena, node0.arena_b if use_bias else None, rt.lr, rt.ws, n=total,
            aux_cnt=getattr(table, 'aux_cnt', None)), 50)
        table.E.copy_(snap[0])
        table.acc.copy_(snap[1])
        res['scatter_adagrad_%s' % table.name] = dict(ms=t, bytes=by, gbs=by / t / 1e6, unique_rows=uniq,
                                                      contributions=nsrc)
    return res


def cpu_baseline(args, syn):
    """The reference's algorithm on the host CPU (numpy fp32 restatement of the TF1
    graph: full-table scorer GEMM, dense gradient, dense Adagrad) on a bounded
    sample: the reference's default batch (64) for ~args.cpu_seconds."""
    from oracle import ref_graph as rg
    B = 64
    d = args.dim
    rng = np.random.default_rng(123)
    params = syn.glorot_params(d, seed=5)
    ref = rg.RefLatentProductModel(d, B, 0.1, syn.u_attr, syn.i_attr, None,
                                   syn.logit_ind2item_ind, loss_function='mw',
                                   n_sampled=args.n_sampled, params=params, dtype=np.float32)
    ref.att_emb.item_ind2logit_ind = {}
    ref.att_emb.target_mapping = lambda item_target: [[0] * len(x) for x in item_target]

    class _Pos(dict):      # {user: [items]} view over the positives CSR (built lazily)
        def __contains__(self, u):
            return 0 <= u < syn.n_users

        def __getitem__(self, u):
            return syn.pos_items[syn.pos_ptr[u]:syn.pos_ptr[u + 1]].tolist()

    ref.prepare_warp(_Pos(), _Pos())
    pool = syn.sample_pool(args.n_sampled, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    users, items = syn.sample_batch(B, rng)
    ref.step(list(users), list(items), pool, id2idx, loss='mw')       # warm-up (BLAS threads)
    t0 = time.time()
    n = 0
    while True:
        users, items = syn.sample_batch(B, rng)
        ref.step(list(users), list(items), None, id2idx, loss='mw')
        n += 1
        if time.time() - t0 > args.cpu_seconds and n >= 2:
            break
    dt = time.time() - t0
    try:
        import threadpoolctl
        th = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        th = os.cpu_count() or 1
    return {"value": B * n / dt, "unit": "interactions/s", "cores": int(th), "kind": "port",
            "sample": "restatement of the TF1 CPU path (TensorFlow unavailable): %d steps at the "
                      "reference's default batch 64, %d-item table, S=%d, numpy fp32 (full-table "
                      "scorer GEMM + dense Adagrad), %.1f s on %d host threads (os.cpu_count=%s)"
                      % (n, args.n_items, args.n_sampled, dt, th, os.cpu_count()),
            "ms_per_step": 1e3 * dt / n}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        if args.n_items is None:
            args.n_items = 100000000      # configs[4]: 100 M-item dim-128 table, row-sharded
        from arx import dist as arx_dist
        return arx_dist.bench_main(args, world, rank, local_rank)
    if args.n_items is None:
        args.n_items = 1000000            # configs[1]

    torch.cuda.set_device(0)
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.synthetic import SyntheticHMF
    from arx import ops

    B, S, d = args.batch, args.n_sampled, args.dim
    t_setup = time.time()
    syn = SyntheticHMF(n_users=args.n_users, n_items=args.n_items, item_mulhot=args.mulhot,
                       permute_logits=False, seed=0, zipf_items=args.zipf_items)
    model = LatentProductModel(args.n_users, args.n_items, d, 1, B, 0.1, 1.0, syn.u_attr, syn.i_attr,
                               syn.item2logit[:args.n_items], syn.logit_ind2item_ind,
                               loss_function='mw', n_sampled=S, use_graph=not args.no_graph)
    model.prepare_warp(syn.positives_csr(), syn.positives_csr())
    dev = model.rt.device
    total = args.steps + args.warmup
    rng = np.random.default_rng(1)
    # the shared negative pool is redrawn every n_resample steps ON DEVICE, inside the timed
    # region (prepare_train.py:7-17 sample_items with p ~ count^0.5 -> arx_sample_wor)
    from arx.utils.prepare_train import DeviceSampler
    sampler = DeviceSampler(syn.item_population, syn.p_sample, device=dev, seed=1)
    nb = min(total, 64)      # device-resident ring of distinct batches
    batches = []
    for _ in range(nb):
        u, i = syn.sample_batch(B, rng)
        batches.append((torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev)))
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    def run(k0, k1):
        for k in range(k0, k1):
            pool = sampler.sample(S) if k % args.n_resample == 0 else None
            u, i = batches[k % nb]
            model.step_async(None, u, i, None, pool, None, loss='mw')

    run(0, args.warmup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    run(args.warmup, total)
    e1.record()
    torch.cuda.synchronize()
    wall = time.time() - t0
    ev_ms = e0.elapsed_time(e1)
    final_loss = float(model.loss.read().item())
    ms_per_step = 1e3 * wall / args.steps

    kr = kernel_rooflines(model, args)
    step_ms = {k: v['ms'] for k, v in kr.items()}
    dom = max((k for k in kr if k.startswith('gemm')), key=lambda k: kr[k]['ms'])
    hb = max((k for k in kr if not k.startswith('gemm')), key=lambda k: kr[k]['ms'])
    roofline = {"kernel": dom, "bound": "mfma", "achieved": kr[dom]['tflops'],
                "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": kr[dom]['tflops'] / FP32_MFMA_PEAK_TF, "traffic": None,
                "flops_per_launch": kr[dom]['flops'], "ms_per_launch": kr[dom]['ms']}
    roofline_hbm = {"kernel": hb, "bound": "hbm", "achieved": kr[hb]['gbs'], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kr[hb]['gbs'] / HBM_PEAK_GBS, "traffic": None,
                    "bytes_per_launch": kr[hb]['bytes'], "ms_per_launch": kr[hb]['ms']}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            # PMC passes (FETCH_SIZE / WRITE_SIZE, tools/profile.sh) of this workload: HBM-side
            # bytes per launch of the kernels behind the two roofline entries
            tr = json.load(open(pmc))
            want = "c%d_b%d" % (3 if args.mulhot else 2, B)
            tags = sorted(k for k in tr if want in k)
            if tags:
                ent = tr[tags[-1]]
                roofline["traffic"] = (ent.get(dom) or {}).get("traffic_bytes")
                roofline["traffic_source"] = "profiles/pmc_traffic.json:%s" % tags[-1]
                hb_key = {"gather": "gather_mulhot" if args.mulhot else "gather_onehot",
                          "scatter": "sparse_apply_window"}.get(hb.split("_")[0])
                roofline_hbm["traffic"] = (ent.get(hb_key) or {}).get("traffic_bytes")
                if hb_key == "sparse_apply_window":
                    # the PMC pass sees the step's FUSED apply (user + item tables in one launch);
                    # bytes_per_launch above is this one table's stand-alone sort + apply
                    roofline_hbm["traffic_note"] = "PMC figure = fused user+item window apply of the step"
        except Exception:
            pass

    out = {
        "metric": "training interactions/sec + sampled-negatives/sec, dim-128, 1/2/4/8 MI355X",
        "value": B * args.steps / wall, "unit": "interactions/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "C2: synthetic %d-item/%d-user HMF, dim %d, id-only%s, WMRB 'mw' loss, "
                               "%d shared negatives/step (pool redrawn on device every %d steps), Adagrad, "
                               "B=%d interactions/step" % (args.n_items, args.n_users, d,
                                                           " + multi-hot item attribute (C3)" if args.mulhot else "",
                                                           S, args.n_resample, B),
                   "batch": B, "n_sampled": S, "dim": d, "n_items": args.n_items,
                   "n_users": args.n_users, "hipgraph": not args.no_graph,
                   "sampled_negative_logits_per_s": B * S * args.steps / wall,
                   "pool_rows_per_s": S * args.steps / wall,
                   "hip_event_ms_per_step": ev_ms / args.steps, "final_loss": final_loss,
                   "setup_s": setup_s},
        "roofline": roofline, "roofline_hbm": roofline_hbm, "kernels_ms": step_ms,
        "kernels": kr,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, syn)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
