#!/usr/bin/env python
"""bench.py -- A-RecSys hot path on MI355X: training interactions/sec.

ONE process, ONE JSON line (rank 0).  The headline workload is BASELINE.json configs[2] ("C3"):
synthetic 1M-item / 1M-user HMF, dim 128, items = id feature + a multi-hot category attribute
(~20 tokens per item over a 100 k-row table; HET layout: two tables, logits = mean of the two
feature scores, embed_attribute.py:205), WMRB sampled loss ('mw'), 1024 negatives shared per
step, pool redrawn ON DEVICE every 50 steps -- the first timed step always redraws --
Adagrad.  One "step" is one full pass of the hot path over one batch: user gather -> pool /
target gathers (one-hot + multi-hot segment-mean) -> scorer GEMM -> target score -> WMRB loss
fwd+bwd -> backward GEMMs -> sparse scatter + Adagrad on every table.  Inputs are
device-resident before timing.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without WORLD_SIZE in the environment: bench.py launches its own N ranks (re-exec through
      torch.distributed.run on 127.0.0.1); under torch.distributed.run (WORLD_SIZE set) it is one rank.
      N > 1 runs BASELINE configs[4] (C5: 100 M-item table row-sharded, arx.dist.ShardedHMF) and carries
      "scaling_anchor": the SAME code path and table on one rank (also: --gpus 1 --workload c5).

`value` times exactly K steps of the headline workload.  The same line carries
  "sub": C2 (configs[1], id-only), C3-MIX (one bag = id token + categories over ONE 1.1 M-row
         table = 563 MB, past the 256 MB Infinity Cache), C4 (configs[3], LSTM, 'mw' and the
         build-defined sampled softmax 'mce'), each with its own rooflines;
  "roofline"      dominant kernel (scorer GEMM, fp32 MFMA), HIP-event timed here;
  "roofline_hbm"  K7 scatter + sparse Adagrad of the step's FUSED passes (all tables: sorts +
                  merge/apply + finish), replayed in situ on the step's own buffers;
  "roofline_gather"  K1 multi-hot gather of the step + a past-LLC K1 measurement;
  "cpu_baseline"  the oracle's restatement of the reference's TF1 CPU algorithm, bounded sample.
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

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3    # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TF = 2500.0   # dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16; MI355X_MICROARCH.md)
# f32-equivalent ceilings of the bf16-pipe scorer: six bf16 MFMA terms per f32 product term (three where one operand
# is the 0/1 activity matrix) -- `achieved` counts 2MNK f32 flops, so the peak is the bf16 peak over 6 (3)
BX6_PEAK_TF = BF16_MFMA_PEAK_TF / 6.0
BX3_PEAK_TF = BF16_MFMA_PEAK_TF / 3.0
METRIC = "training interactions/sec + sampled-negatives/sec, dim-128, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16384,
                    help="interactions per step per GPU (SURVEY 8(d) throughput batches: 4096, 16384)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c3mix", "c5", "c4h"],
                    help="headline workload (default c3 = BASELINE configs[2], HET layout; c5 = configs[4] on ONE "
                         "rank: the N = 1 anchor of the --gpus N curve)")
    ap.add_argument("--no-anchor", action="store_true",
                    help="N > 1: do not run the world-1 anchor of the same workload after the N-rank run")
    ap.add_argument("--mulhot", action="store_true", help="(compat) same as --workload c3")
    ap.add_argument("--subs", default="c2,c3mix,c3mce,c3host,c4,c4mce,c4host,k1,c5w1,c3repw1,topk,c3_f32mfma,c2_f32mfma",
                    help="comma list of sub-results besides the headline ('' = none)")
    ap.add_argument("--sub-steps", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each; the line reports the median one")
    ap.add_argument("--n-items", type=int, default=None,
                    help="item table rows (default: 1M on one GPU; 100M row-sharded for --gpus N > 1 = configs[4])")
    ap.add_argument("--n-users", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--n-sampled", type=int, default=1024)
    ap.add_argument("--n-resample", type=int, default=50)
    ap.add_argument("--zipf-items", type=float, default=1.05,
                    help="popularity exponent of the synthetic item draw (0 = uniform; experiments only)")
    ap.add_argument("--lstm-batch", type=int, default=1024)
    ap.add_argument("--sharded-bags", action="store_true",
                    help="N > 1 (or WORLD_SIZE set): HET items on the sharded step -- id table striped by item, "
                         "a 100 k-token multi-hot table striped by TOKEN (arx.dist.ShardedHMFBags); 1 M items")
    ap.add_argument("--sharded-rep-tokens", action="store_true",
                    help="N > 1 (or WORLD_SIZE set): HET items on the sharded step with the 100 k-token table REPLICATED "
                         "and its merged gradient all-reduced (arx.dist.ShardedHMFRepTokens, round 5); 1 M items")
    ap.add_argument("--exchange", choices=["rows", "logits"], default="rows",
                    help="the sharded id-only step's exchange (--workload c5 / --gpus N): 'rows' = gather the pool rows "
                         "(default), 'logits' = all-to-all of the negative-sample logits (SURVEY 8e steps 1-5; eager launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ring", action="store_true",
                    help="announce the next batch (prepare_next): step t sorts step t + 1's lookups.  Off by default: "
                         "measured slower (DESIGN.md section 4: two graphs launched in turn cost ~15 us per step)")
    ap.add_argument("--no-rooflines", action="store_true", help="skip the per-kernel timings (profiling runs)")
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


def _graph_time_ms(fn, iters=30):
    """fn() captured once into a hipGraph, replayed back to back: no host launch gaps (how the
    kernels run inside the step's graph)."""
    from arx import ops
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = ops.CapturedGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g.begin()
        try:
            fn()
        finally:
            g.end()
    torch.cuda.current_stream().wait_stream(side)
    return _evt_time_ms(g.launch, iters)


def _load_pmc(tag_part):
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return None, None
    try:
        tr = json.load(open(pmc))
    except Exception:
        return None, None
    tags = sorted(k for k in tr if tag_part in k)
    return (tr[tags[-1]], "profiles/pmc_traffic.json:%s" % tags[-1]) if tags else (None, None)


def k7_in_situ(model, d):
    """K7 (lookup-gradient scatter + sparse Adagrad) of the step, ALL tables, exactly the passes
    the step runs (fused one-hot pass, two-stage multi-hot pass), on the step's own ids and
    gradient arena: (sorts + merge/apply + finish) and the sorts alone, as hipGraph replays.
    Algorithmic bytes per SURVEY 8(d): (16d+4) per unique updated row + 4d per gradient source
    row + 12 per contribution."""
    from arx import ops
    plan = model._plan('train')
    rt = model.rt
    uniq = contrib = src_rows = 0
    per_table = {}
    for table, sites, bufs, total in plan.tables:
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
        u, c, m = int(torch.unique(valid).numel()), int(valid.numel()), sum(s.n for s in sites)
        per_table[table.name] = dict(unique_rows=u, contributions=c, source_rows=m)
        uniq, contrib, src_rows = uniq + u, contrib + c, src_rows + m
    by = uniq * (16 * d + 4) + src_rows * 4 * d + contrib * 12
    snaps = [(t.E.clone(), t.acc.clone(), None if t.bias is None else t.bias.clone(),
              None if t.bias is None else t.bias_acc.clone()) for t, _, _, _ in plan.tables]

    def full():
        plan._k7_early = None
        plan._apply_sparse()
    t_full = _graph_time_ms(full)
    jobs = list(plan._jobs)

    def sorts():
        for kind, what, key in (j[:3] for j in jobs):
            if kind == 'multi':
                plan._apply_multi(what[0], phase=1, key=key, bag=what[1])
            elif kind == 'bags':
                plan._bag_pass(what, key, phase=1)
            else:
                plan._cat_pass(what, key, phase=1)
    t_sort = _graph_time_ms(sorts) if jobs and len(jobs) == plan._n_passes else None
    for (t, _, _, _), (E, acc, b, ba) in zip(plan.tables, snaps):
        t.E.copy_(E); t.acc.copy_(acc)
        if b is not None:
            t.bias.copy_(b); t.bias_acc.copy_(ba)
    res = dict(ms=t_full, bytes=by, gbs=by / t_full / 1e6, unique_rows=uniq, contributions=contrib,
               source_rows=src_rows, tables=per_table, passes=[j[0] for j in jobs])
    if t_sort is not None:
        res['ms_sorts'] = t_sort
        res['ms_apply'] = t_full - t_sort
        res['gbs_apply'] = by / max(t_full - t_sort, 1e-6) / 1e6
    return res


def run_topk(args, B=4096, k=100):
    """SURVEY 8(f) #3, the other side of training: tf.nn.top_k over the FULL vocabulary for B users (hmf_model.py:154)
    without the [B, V] logits -- arx.hmf.hmf_model.StreamTopK on random rows of the C2 shape, fused (threshold-filter
    epilogue of the scorer GEMM) against chunked (GEMM + radix select + merge per 65 536 columns)."""
    from arx import graph as G
    from arx.hmf.hmf_model import StreamTopK
    dev = torch.device('cuda', 0)
    rt = G.Runtime(dev)
    V, d = int(args.n_items), int(args.dim)
    g = torch.Generator(device=dev)
    g.manual_seed(0)

    class Leaf(G.Node):
        def __init__(self, shape):
            super().__init__(rt, shape)
            self.value = torch.randn(shape, device=dev, generator=g) * 0.3
            self.bias_value = None

    lat, pool = Leaf((B, d)), Leaf((V, d))
    pool.bias_value = torch.randn(V, device=dev, generator=g) * 0.1
    res = {"config": {"workload": "full-vocabulary top-%d, %d users x %d items, d=%d, f32" % (k, B, V, d)}}
    for mode in ("fused", "chunked"):
        tk = StreamTopK(rt, lat, pool, k)
        tk.fused = mode == "fused"
        for _ in range(2):
            tk.forward(False)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 3
        for _ in range(n):
            tk.forward(False)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / n * 1e3
        res[mode] = {"ms": ms, "user_item_scores_per_s": B * V / ms * 1e3, "tflops_f32": 2.0 * B * V * d / ms / 1e9,
                     "overflow": int(tk.overflow.item())}
        del tk
    res["value"] = res["fused"]["user_item_scores_per_s"]
    res["unit"] = "user x item scores ranked /s"
    res["speedup_vs_chunked"] = res["chunked"]["ms"] / res["fused"]["ms"]
    return res


def kernel_rooflines(model, d):
    """Per-kernel timings of the step's main kernels with the step's own buffers."""
    from arx import ops, graph as G
    rt = model.rt
    plan = model._plan('train')
    pred = [n for n in plan.order if isinstance(n, G.Prediction)][0]
    latent, pool = pred.inputs
    B, S = latent.shape[0], pool.shape[0]
    ws = rt.ws
    res = {}
    # --- GEMMs (MFMA bound) ---
    flops = 2.0 * B * S * d
    if pred.fused_into_loss:
        # 'mw' train plans (default): csrc/scorer.hip -- no logits / dlogits; each launch of the forward timed alone
        # (arx_mw_scorer_fwd_phases), the hinge GEMM and the two bit-operand products priced against the bf16 pipe
        bl = [n for n in plan.order if isinstance(n, G.BatchLoss) and n.gemm_fused][0]
        ms = bl.mask
        ptr, items = ms.pos_getter()
        uid, i2s = ms.user_ids.value, ms.slot_map_getter()
        tgt = bl.inputs[1]
        lat_, te = tgt.inputs
        sc = pred.scorer

        def fwd(ph):
            sc.fwd(lat_.value, pool.value, pool.bias_value, te.value, te.bias_value, uid, ptr, items, i2s, bl.value,
                   tgt.value, tgt.grad, lat_.grad, te.grad, bl.gscale, mask_rows=bl.mask_rows, phases=ph)
        t = _evt_time_ms(lambda: fwd(1), 50)
        res['scorer_prep'] = dict(ms=t, note="pool rows -> bf16 planes (both layouts) + positives' hit lists")
        t = _evt_time_ms(lambda: fwd(2), 50)
        res['gemm_logits_hinge'] = dict(ms=t, flops=flops, tflops=flops / t / 1e9, peak=BX6_PEAK_TF,
                                        note="k_sc_hinge: target score + scorer GEMM + hinge epilogue (act bits)")
        t = _evt_time_ms(lambda: fwd(4), 50)
        res['scorer_rows'] = dict(ms=t, note="loss, g, rank-one terms, g U planes, transposed bits")
        gU, gP = latent.alloc_grad(), pool.alloc_grad()
        t = _evt_time_ms(lambda: sc.bwd_dU(gU, beta=1.0), 50)
        res['gemm_dU_bits'] = dict(ms=t, flops=flops, tflops=flops / t / 1e9, peak=BX3_PEAK_TF)
        t = _evt_time_ms(lambda: sc.bwd_dI(gP, db=pool.bias_grad), 50)
        res['gemm_dI_bits'] = dict(ms=t, flops=flops, tflops=flops / t / 1e9, peak=BX3_PEAK_TF,
                                   note="k_sc_bits over K slices + k_sc_tn_reduce")
    else:
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
            tab_bytes = 0
            for f in n.feats:
                tab_bytes += f.table.E.numel() * 4
                if f.kind == 'cat':
                    by += rows * (4 * d + 4 + 4 + (4 if n.with_bias else 0))
                else:
                    lens = f.maps[2][n.inputs[0].value.long()].sum().item()
                    toks += lens
                    by += lens * (4 * d + 4) + rows * 8
            by += rows * 4 * d   # output write (once; accumulate re-reads stay in L2)
            t = _evt_time_ms(lambda n=n: n.forward(False), 50)
            res['gather_%s_%d' % (n.inputs[0].name, rows)] = dict(
                ms=t, bytes=by, gbs=by / t / 1e6, tokens=toks, table_bytes=tab_bytes,
                kind='mulhot' if toks else 'onehot')
    # the step issues all its lookups as ONE launch (arx_lookup_multi): that launch, on the step's ids
    per_node = {id(n): res['gather_%s_%d' % (n.inputs[0].name, n.shape[0])] for n in plan.order
                if isinstance(n, G.EntityEmbed) and 'gather_%s_%d' % (n.inputs[0].name, n.shape[0]) in res}
    for gs, grp in (plan._pregather or []):
        if isinstance(gs, ops.LookupSet) and all(id(n) in per_node for n in grp):
            t = _evt_time_ms(lambda gs=gs: ops.lookup_multi(gs), 50)
            by = sum(per_node[id(n)]['bytes'] for n in grp)
            tabs = {}
            for n in grp:
                for f in n.feats:
                    tabs[id(f.table)] = f.table.E.numel() * 4
            res['gather_step_lookups_%d' % sum(n.shape[0] for n in grp)] = dict(
                ms=t, bytes=by, gbs=by / t / 1e6, tokens=sum(per_node[id(n)]['tokens'] for n in grp),
                table_bytes=sum(tabs.values()), kind='mulhot', pmc_key='lookup_multi', sites=len(grp))
    res['k7_step_fused'] = k7_in_situ(model, d)
    return res


def k1_past_llc(dev, d=128, bags=65536, L=20, V=4194304):
    """K1 (multi-hot gather + segment-mean) as an HBM measurement: a 2 GB table (V rows x 512 B, 8x the
    256 MB Infinity Cache); the tokens of one launch are a slice of a PERMUTATION of the rows (no row
    is read twice inside a launch, so L2 / LLC cannot serve in-launch repeats) and consecutive
    launches rotate over three disjoint slices (a slice is 671 MB: by the time one is read again
    1.3 GB of other rows went through the 256 MB cache).  Every byte counted is read from HBM:
    algorithmic bytes == distinct bytes here.  516 B/token + 520 B/bag."""
    from arx import ops
    rng = np.random.default_rng(0)
    E = torch.randn(V, d, device=dev)
    n_tok = bags * L
    n_sets = min(3, V // n_tok)
    perm = rng.permutation(V).astype(np.int32)
    lens = np.full(bags + 1, L, dtype=np.int32)
    starts = np.zeros(bags + 2, dtype=np.int32)
    starts[1:] = np.cumsum(lens)
    tst, tl = torch.from_numpy(starts).to(dev), torch.from_numpy(lens).to(dev)
    sets = []
    for q in range(n_sets):
        v = np.zeros(int(starts[-1]), dtype=np.int32)
        v[:n_tok] = perm[q * n_tok:(q + 1) * n_tok]
        sets.append(torch.from_numpy(v).to(dev))
    ids = torch.arange(bags, dtype=torch.int32, device=dev)
    out = torch.empty(bags, d, device=dev)
    k = [0]

    def launch():
        ops.gather_mulhot_mean(E, None, sets[k[0] % n_sets], tst, tl, ids, out)
        k[0] += 1
    t = _evt_time_ms(launch, 30)
    by = bags * L * (4 * d + 4) + bags * (4 * d + 8)
    # HBM bytes by the PMC counters (FETCH_SIZE x2 + WRITE_SIZE) of the builder's own rocprofv3 --pmc passes over
    # tools/k1_physical.py (the same launch), per launch
    pmc, src = _load_pmc("k1_past_llc")
    tr = ((pmc or {}).get("gather_mulhot") or {}).get("traffic_bytes")
    return {"kernel": "k_gather_mulhot (K1), %d bags x %d tokens over a %d-row table (%.0f MB = %.0fx the 256 MB "
                      "LLC); tokens of a launch = a permutation slice (no repeated row), %d disjoint slices "
                      "rotated between launches" % (bags, L, V, V * d * 4 / 1e6, V * d * 4 / (256 << 20), n_sets),
            "bound": "hbm", "achieved": by / t / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": by / t / 1e6 / HBM_PEAK_GBS, "bytes_per_launch": by, "distinct_bytes_per_launch": by,
            "ms_per_launch": t, "traffic": tr, "traffic_source": src}


def cpu_baseline(args, syn, label):
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
            "sample": "restatement of the TF1 CPU path (TensorFlow unavailable) on the %s workload: %d steps "
                      "at the reference's default batch 64, S=%d, numpy fp32 (full-table scorer GEMMs + dense "
                      "Adagrad over every table), %.1f s on %d host threads (os.cpu_count=%s)"
                      % (label, n, args.n_sampled, dt, th, os.cpu_count()),
            "ms_per_step": 1e3 * dt / n}


WORKLOADS = {
    # name: (label, SyntheticHMF kwargs)
    "c2": ("C2 (BASELINE configs[1]): id-only", dict()),
    "c3": ("C3 (BASELINE configs[2]): id feature + multi-hot category attribute, HET layout (two tables, "
           "~20 tokens/item over 100 k rows)", dict(item_mulhot=True)),
    "c3mix": ("C3-MIX: ONE bag per item = id token + ~20 category tokens over ONE table of 1.1 M rows "
              "(563 MB, past the 256 MB LLC; comb_attribute.py:100-148)", dict(item_mix=True)),
}


def _quiesce():
    """Between two workloads of one process: the finished workload's models (plans, captured hipGraphs) sit in reference
    cycles, so they are destroyed whenever the cyclic collector next runs -- if that is in the middle of the NEXT
    workload's replay loop, hipGraphExecDestroy runs beside its hipGraphLaunch calls, and the launch segfaulted in the
    HIP runtime (round 6: 2 of 3 default runs, always in the last graph workload, never with fewer workloads in front of
    it).  Collect and synchronise at the quiet point instead."""
    import gc
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.synchronize()


def run_hmf(args, name, steps, warmup, with_cpu=False, loss='mw', host_feed=False):
    from arx.hmf.hmf_model import LatentProductModel
    from arx.utils.synthetic import SyntheticHMF
    from arx.utils.prepare_train import DeviceSampler
    label, kw = WORKLOADS[name]
    B, S, d = args.batch, args.n_sampled, args.dim
    t_setup = time.time()
    syn = SyntheticHMF(n_users=args.n_users, n_items=args.n_items, permute_logits=False, seed=0,
                       zipf_items=args.zipf_items, **kw)
    model = LatentProductModel(args.n_users, args.n_items, d, 1, B, 0.1, 1.0, syn.u_attr, syn.i_attr,
                               syn.item2logit[:args.n_items], syn.logit_ind2item_ind,
                               loss_function=loss, n_sampled=S, use_graph=not args.no_graph)
    model.prepare_warp(syn.positives_csr(), syn.positives_csr())
    dev = model.rt.device
    total = steps + warmup
    rng = np.random.default_rng(1)
    # the shared negative pool is redrawn every n_resample steps ON DEVICE, inside the timed
    # region (prepare_train.py:7-17 sample_items with p ~ count^0.5 -> arx_sample_wor); the
    # cadence is counted from the first TIMED step, so every timed region holds >= 1 redraw
    sampler = DeviceSampler(syn.item_population, syn.p_sample, device=dev, seed=1)
    nb = min(total, 64)      # device-resident ring of distinct batches
    batches = []
    for _ in range(nb):
        u, i = syn.sample_batch(B, rng)
        if host_feed:
            # the reference's own hand-over (hmf_model.py:162-175 step(): python lists into feed_dict): the ids of a
            # step arrive as pageable host arrays and cross PCIe inside the timed region -- reported as a sub-result
            # (`c3host`), never as `value`
            batches.append((np.ascontiguousarray(u, dtype=np.int32), np.ascontiguousarray(i, dtype=np.int32)))
        else:
            batches.append((torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev)))
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    redraws = [0]

    def run(k0, k1):
        # --ring: the loader hands the NEXT step's ids (and its pool, when that step redraws) over one step early
        # (LatentProductModel.prepare_next): the ids-only half of the sparse update then runs a step ahead
        drawn = {}

        def pool_of(k):
            if k == 0 or (k >= warmup and (k - warmup) % args.n_resample == 0):
                if k not in drawn:
                    drawn[k] = sampler.sample(S)
                    redraws[0] += k >= warmup
                return drawn[k]
            return None
        for k in range(k0, k1):
            pool = pool_of(k)
            u, i = batches[k % nb]
            if args.ring and k + 1 < k1:
                un, in_ = batches[(k + 1) % nb]
                model.prepare_next(un, in_, pool_of(k + 1))
            model.step_async(None, u, i, None, pool, None, loss=loss)
            drawn.pop(k, None)

    run(0, warmup)
    torch.cuda.synchronize()
    # The timed region is EXACTLY `steps` steps between two synchronisations -- and it is repeated:
    # at the driver's --steps 20 one region is 7 ms, a single host hiccup moves it by several
    # percent.  `value` is the MEDIAN region (min / max in config); wall clock and HIP events side
    # by side.  Every region starts on a pool redraw (the cadence restarts with the region).
    walls, evs, hosts = [], [], []
    for rep in range(max(1, args.repeats)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.time()
        e0.record()
        run(warmup, total)
        e1.record()
        hosts.append(time.time() - t0)          # host time to ENQUEUE the region (the GPU is still running)
        torch.cuda.synchronize()
        walls.append(time.time() - t0)
        evs.append(e0.elapsed_time(e1))
    order = sorted(range(len(walls)), key=lambda r: walls[r])
    med = order[len(order) // 2]
    wall, ev_ms = walls[med], evs[med]
    redraws[0] //= len(walls)
    final_loss = float(model.loss.read().item())
    out = {
        "value": B * steps / wall, "unit": "interactions/s", "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * wall / steps,
        "config": {"workload": "%s; synthetic %d-item/%d-user HMF, dim %d, %s, %d shared "
                               "negatives/step (pool redrawn on device every %d steps, %d redraw(s) inside the "
                               "timed region), Adagrad, B=%d interactions/step"
                               % (label, args.n_items, args.n_users, d,
                                  "WMRB 'mw' loss" if loss == 'mw' else
                                  "sampled softmax 'mce' (build-defined: the reference accepts the name and holds no "
                                  "arithmetic for it) on its fused family, no [B, S] array",
                                  S, args.n_resample, redraws[0], B),
                   "batch": B, "n_sampled": S, "dim": d, "n_items": args.n_items, "n_users": args.n_users,
                   "hipgraph": not args.no_graph, "pool_redraws_timed": redraws[0],
                   "next_batch_announced": bool(args.ring),
                   "ids_fed_from": ("host (pageable numpy int32, 2 x %d B per step over PCIe inside the timed region)"
                                    % (4 * B)) if host_feed else "HBM (device-resident ring of batches)",
                   "host_enqueue_ms_per_step": 1e3 * hosts[med] / steps,
                   "sampled_negative_logits_per_s": B * S * steps / wall,
                   "pool_rows_per_s": S * steps / wall, "hip_event_ms_per_step": ev_ms / steps,
                   "timed_regions": len(walls), "ms_per_step_min": 1e3 * min(walls) / steps,
                   "ms_per_step_max": 1e3 * max(walls) / steps,
                   "final_loss": final_loss, "setup_s": setup_s},
    }
    if not args.no_rooflines:
        kr = kernel_rooflines(model, d)
        pmc, pmc_src = _load_pmc("%s_b%d" % (name, B))
        dom = max((k for k in kr if k.startswith('gemm')), key=lambda k: kr[k]['ms'])
        peak = kr[dom].get('peak', FP32_MFMA_PEAK_TF)
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": kr[dom]['tflops'],
                           "peak": peak, "unit": "TFLOP/s", "frac": kr[dom]['tflops'] / peak,
                           "peak_note": ("f32-equivalent flops (2MNK) against the dense bf16 MFMA peak over the number "
                                         "of bf16 terms per f32 product term" if 'peak' in kr[dom] else
                                         "f32-input MFMA peak"),
                           "traffic": ((pmc or {}).get(dom) or {}).get("traffic_bytes"),
                           "flops_per_launch": kr[dom]['flops'], "ms_per_launch": kr[dom]['ms']}
        ns, src = _instep_ns("%s_b%d" % (name, B), "k_sc_hinge" if "hinge" in dom else "\0")
        if ns:
            # the duration the kernel has INSIDE the step (rocprofv3 --kernel-trace --stats of this very command,
            # builder's committed summary): what the step pays, quoted next to the stand-alone HIP-event figure
            out["roofline"]["ms_in_step_rocprof"] = ns / 1e6
            out["roofline"]["frac_in_step"] = kr[dom]['flops'] / (ns / 1e9) / 1e12 / peak
            out["roofline"]["in_step_source"] = src
        k7 = kr['k7_step_fused']
        out["roofline_hbm"] = {
            "kernel": "K7 scatter + sparse Adagrad, the step's fused passes over ALL tables (%s): sorts + run "
                      "records + apply (k_run_apply; the one-hot pass a bag table rides on: window apply + "
                      "finish), replayed in situ" % "+".join(k7['passes']),
            "bound": "hbm", "achieved": k7['gbs'], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": k7['gbs'] / HBM_PEAK_GBS, "bytes_per_launch": k7['bytes'], "ms_per_launch": k7['ms'],
            "ms_sorts": k7.get('ms_sorts'), "ms_apply_finish": k7.get('ms_apply'),
            "frac_apply_finish_only": (k7['gbs_apply'] / HBM_PEAK_GBS) if 'gbs_apply' in k7 else None,
            "unique_rows": k7['unique_rows'], "contributions": k7['contributions'],
            "traffic": sum(((pmc or {}).get(k) or {}).get("traffic_bytes") or 0
                           for k in ("sparse_apply_window", "sparse_finish", "sparse_apply_runs")) or None}
        gk = [k for k in kr if k.startswith('gather') and kr[k].get('kind') == 'mulhot']
        if gk:
            fused_lookup = [k for k in gk if 'pmc_key' in kr[k]]
            g = fused_lookup[0] if fused_lookup else max(gk, key=lambda k: kr[k]['ms'])
            tr = ((pmc or {}).get(kr[g].get('pmc_key', "gather_mulhot")) or {}).get("traffic_bytes")
            alg_gbs = kr[g]['gbs']
            ent = {"kernel": "K1 " + g, "bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS, "bytes_per_launch": kr[g]['bytes'],
                   "ms_per_launch": kr[g]['ms'], "traffic": tr, "table_bytes": kr[g]['table_bytes']}
            if alg_gbs > HBM_PEAK_GBS or kr[g]['table_bytes'] < (256 << 20):
                # the table sits in L2 / the 256 MB Infinity Cache (Zipf-hot rows): algorithmic bytes over time is a
                # cache figure, not an HBM rate, and is NOT reported as `achieved` (round-5 verdict, weak #10): the
                # HBM-side figure is the PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, profiles/) over the same time; the
                # gather's headline is the physical past-LLC K1 measurement (sub `k1`, promoted in main())
                ent["note"] = ("table is LLC-resident: algorithmic bytes/time would be a cache-served rate and is "
                               "not reported; achieved/frac are PMC traffic over time when a profile is present")
                ent["algorithmic_bytes_over_time_cache_served"] = alg_gbs
                ent["achieved"] = (tr / kr[g]['ms'] / 1e6) if tr else None
                ent["frac"] = (ent["achieved"] / HBM_PEAK_GBS) if tr else None
            out["roofline_gather"] = ent
        if pmc_src:
            # the `traffic` fields are NOT measured by this run: they are read from the committed PMC summary of
            # the builder's own rocprofv3 --pmc passes over the same command (tools/profile.sh; FETCH_SIZE x2)
            out["traffic_source"] = pmc_src
            for rk in ("roofline", "roofline_hbm", "roofline_gather"):
                if rk in out and out[rk].get("traffic") is not None:
                    out[rk]["builder_profile"] = True
        out["kernels_ms"] = {k: v['ms'] for k, v in kr.items()}
        out["kernels"] = kr
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(args, syn, name.upper())
    del model, syn, sampler, batches
    torch.cuda.empty_cache()
    return out


def run_lstm(args, loss, steps, warmup, host_feed=False):
    """C4 (BASELINE configs[3]): LSTM seqModel d = h = 64, L = 50, 1 M items, S = 1024 sampled
    negatives, use_concat=False, clip 5.0, Adagrad lr 0.5; targets/s = sum(weights)/wall
    (lstm/run.py:466-470).  loss 'mw' (reference arithmetic) or 'mce' (build-defined sampled
    softmax: the reference accepts the flag but has no arithmetic for it)."""
    from arx import ops
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.lstm.seqModel import SeqModel, LSTM
    from arx.utils.synthetic import SyntheticHMF
    B, L, S, size = args.lstm_batch, 50, args.n_sampled, 64
    t0 = time.time()
    syn = SyntheticHMF(n_users=args.n_users, n_items=args.n_items, permute_logits=False, seed=0)
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    START = args.n_items
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, L, False, None, syn.logit_ind2item_ind)
    model = SeqModel([L], size, 1, 5.0, B, 0.5, 0.99, emb, loss=loss, use_concat=False, START_ID=START)
    emb.rt.use_graph = not args.no_graph
    emb.prepare_warp(syn.positives_csr(), syn.positives_csr())
    dev = model.rt.device
    rng = np.random.default_rng(1)
    total = steps + warmup
    nb = min(total, 8)
    batches = []
    for _ in range(nb):
        users = rng.integers(0, args.n_users, size=B).astype(np.int32)
        tg = np.stack([syn.sample_batch(B, rng)[1] for _ in range(L)], 0).astype(np.int32)
        inp = np.concatenate([np.full((1, B), START, dtype=np.int32), tg[:-1]], 0)
        lens = rng.integers(10, L + 1, size=B)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float32)
        if host_feed:
            # the reference's hand-over (seqModel.py:289-404 get_batch -> step(): host arrays into feed_dict): ids and
            # weights cross PCIe inside the timed region -- the `c4host` sub-result, never `value`
            batches.append((users, np.ascontiguousarray(inp), np.ascontiguousarray(tg), w, float(w.sum())))
        else:
            batches.append((torch.from_numpy(users).to(dev), torch.from_numpy(inp).to(dev),
                            torch.from_numpy(tg).to(dev), torch.from_numpy(w).to(dev), float(w.sum())))
    pool = torch.from_numpy(syn.sample_pool(S, rng)).to(dev)
    setup_s = time.time() - t0

    def run(k0, k1):
        tot, node = 0.0, None
        for k in range(k0, k1):
            u, i, t, w, ws = batches[k % nb]
            node = model.step_async(None, u, i, t, w, 0, pool if (k == 0 or k == warmup) else None, None)
            tot += ws
        return tot, node

    run(0, warmup)
    torch.cuda.synchronize()
    t1 = time.time()
    tot_w, node = run(warmup, total)
    torch.cuda.synchronize()
    wall = time.time() - t1
    per_target = float(node.read().item()) / max(batches[(total - 1) % nb][4], 1.0)
    out = {"value": tot_w / wall, "unit": "targets/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * wall / steps,
           "config": {"workload": "C4 (BASELINE configs[3]): LSTM d=h=%d, L=%d, B=%d sequences, %d items, S=%d "
                                  "sampled negatives, loss '%s'%s, clip 5.0, Adagrad"
                                  % (size, L, B, args.n_items, S, loss,
                                     " (build-defined sampled softmax: no reference arithmetic, no parity claim)"
                                     if loss == 'mce' else ""),
                      "timestep_rows_per_s": L * B * steps / wall, "final_loss_per_target": per_target,
                      "ids_fed_from": ("host (pageable numpy: users, inputs, targets, weights = %d B per step over "
                                       "PCIe inside the timed region)" % (4 * B + 12 * L * B)) if host_feed
                      else "HBM (device-resident ring of batches)",
                      "setup_s": setup_s}}
    if not args.no_rooflines:
        plan = model._plan(0, 'train')
        ln = [n for n in plan.order if isinstance(n, LSTM)][0]
        t_f = _evt_time_ms(lambda: ln.forward(True), 20)
        t_b = _evt_time_ms(lambda: ops.lstm_bwd(ln.W.w, ln.value, ln.cs, ln.gates, ln.grad, L, B, ln.din, ln.h, ln.dz), 20)
        fl = 2.0 * L * B * (ln.din + ln.h) * 4 * ln.h
        fl_b = 2.0 * L * B * 4 * ln.h * ln.h          # the kernel's recurrence dh = dz . W_h^T (dW / dx are GEMMs outside it)
        out["roofline"] = {"kernel": "k_lstm_fwd (persistent, all L steps; gate GEMM on fp32 MFMA)", "bound": "mfma",
                           "achieved": fl / t_f / 1e9, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                           "frac": fl / t_f / 1e9 / FP32_MFMA_PEAK_TF, "flops_per_launch": fl,
                           "ms_per_launch": t_f, "traffic": None,
                           "bwd": {"ms_per_launch": t_b, "flops_per_launch": fl_b,
                                   "achieved": fl_b / t_b / 1e9, "frac": fl_b / t_b / 1e9 / FP32_MFMA_PEAK_TF}}
        from arx import graph as G
        pred = [n for n in plan.order if isinstance(n, G.Prediction)][0]
        if loss == 'mce' and pred.fused_into_loss and isinstance(pred.scorer, ops.MceScorer):
            # the 'mce' family (csrc/scorer.hip k_mc_flow): the pass that leaves s_r and O_r (loss + latent-side
            # product) and the pool-side pass, each 12 bf16 MFMA terms of 2 (L B) S d (six for the recomputed logit
            # tile, six for the product out of the accumulators); no [L B, S] array exists
            bl = [n for n in plan.order if isinstance(n, G.BatchLoss) and n.gemm_fused][0]
            ms = bl.mask
            ptr, items = ms.pos_getter()
            uid, i2s = ms.user_ids.value, ms.slot_map_getter()
            tgt = bl.inputs[1]
            lat_, te = tgt.inputs
            pool_ = pred.inputs[1]
            sc = pred.scorer
            rw = bl.row_w.value if bl.row_w is not None else None
            t_u = _evt_time_ms(lambda: sc.fwd(lat_.value, pool_.value, pool_.bias_value, te.value, te.bias_value, uid,
                                              ptr, items, i2s, bl.value, tgt.value, tgt.grad, lat_.grad, te.grad,
                                              bl.gscale, row_w=rw, mask_rows=bl.mask_rows, phases=2), 20)
            t_i = _evt_time_ms(lambda: sc.bwd_dI(pool_.grad, db=pool_.bias_grad, step_rows=B, dI_steps=pred.C_steps,
                                                 db_steps=pred.rs_steps), 20)
            fl12 = 12 * 2.0 * L * B * S * size
            out["roofline_scorer"] = {
                "kernel": "k_mc_flow<dU role> (x tile six terms + e . P six terms out of the accumulators)",
                "bound": "mfma", "achieved": fl12 / t_u / 1e9, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": fl12 / t_u / 1e9 / BF16_MFMA_PEAK_TF, "flops_per_launch": fl12, "ms_per_launch": t_u,
                "traffic": None,
                "dI": {"kernel": "k_mc_flow<dI role> + k_sc_tn_reduce", "ms_per_launch": t_i,
                       "achieved": fl12 / t_i / 1e9, "frac": fl12 / t_i / 1e9 / BF16_MFMA_PEAK_TF},
                "note": "no [L*B, S] logits / weights in HBM; the materialising path (ARX_MCE_FUSED=0) runs the "
                        "same step ~125 us slower"}
    del model, emb, syn, batches
    torch.cuda.empty_cache()
    return out


def run_sharded_world1(args, rep_tokens=False):
    """The N > 1 code path (arx.dist.ShardedHMF, BASELINE configs[4]: 100 M-item table) on ONE rank:
    every exchange is a local copy.  This -- not the hipGraph single-process headline above, which
    is another workload (C3, 1 M items) -- is the N = 1 point of the weak-scaling curve that
    `bench.py --gpus N` continues for N = 2, 4, 8."""
    import copy
    import torch.distributed as dist
    from arx import dist as arx_dist
    a = copy.copy(args)
    a.n_items = 100000000
    if rep_tokens:
        # the HET variant of the sharded step (arx.dist.ShardedHMFRepTokens, round 5): C3's items (1 M, id + 20-token
        # bag over a 100 k-token table) with the id table striped and the token table replicated
        a.n_items, a.sharded_rep_tokens = 1000000, True
    a.steps, a.warmup = args.sub_steps, min(args.warmup, 10)
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port())),
                 ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
        os.environ.setdefault(k, v)
    try:
        out = arx_dist.bench_run(a, 1, 0, 0)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    return {k: out[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "n_gpus", "config", "roofline")}


DTYPE_DETAIL = ("f32 operands as 3 exact bf16 pieces, 6 bf16 MFMA terms (3 where one operand is the 0/1 activity "
                "matrix), f32 accumulate; everything else f32 (gathers, loss, Adagrad)")


def run_f32mfma(args, workload):
    """The A/B reference beside the headline (VERDICT r3 ruling, item iv): the same workload with the scorer on the
    f32-input MFMA kernels (logits GEMM, wave-per-row loss kernel over [B, S] logits, two f32 backward GEMMs) --
    the default path of rounds 1-3.  The switch is read once per process: a child process runs the workload."""
    import subprocess
    env = dict(os.environ, ARX_SCORER_F32="1", ARX_BENCH_CHILD="1")
    sharded = workload == "c5w1"            # the world-1 sharded step: a sub-result of a (short) child run
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c2" if sharded else workload,
           "--subs", "c5w1" if sharded else "", "--no-rooflines",
           "--no-cpu-baseline", "--steps", str(10 if sharded else args.sub_steps), "--warmup", str(min(args.warmup, 10)),
           "--sub-steps", str(args.sub_steps), "--repeats", "1" if sharded else str(args.repeats),
           "--batch", str(args.batch), "--n-sampled", str(args.n_sampled), "--dim", str(args.dim)]
    if args.n_items:
        cmd += ["--n-items", str(args.n_items)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    j = _child_detail(r.stdout)
    if sharded:
        j = j["sub"]["c5w1"]
        j.setdefault("steps", args.sub_steps)
    return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
            "dtype": "f32", "dtype_detail": "f32-input MFMA (v_mfma_f32_32x32x2_f32), f32 accumulate",
            "switches": "ARX_SCORER_F32=1 (the default path of rounds 1-3)",
            "parity": "tests/test_f32mfma_gpu.py re-runs the kernel, whole-step, sharded and BASELINE-sized oracle tests "
                      "with the switch set, same 1e-4",
            "config": {k: j["config"][k] for k in ("workload", "final_loss", "ms_per_step_min", "ms_per_step_max")
                       if k in j["config"]}}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves -- a re-exec of this very
    command line through torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) -- and hand its exit
    code back.  Rank 0 of the child job prints the ONE JSON line to our stdout (lstm/run.py:87,221-229 is the
    reference's only notion of a device list; there the devices are TF towers of one process)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def scaling_anchor(args):
    """The N = 1 point of the weak-scaling curve for an N > 1 line: the SAME sharded code path, table and per-GPU
    batch on ONE rank, run by rank 0 in a child process once the N-rank job has released the GPUs."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE",
                        "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE",
                        "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE")}
    env["ARX_BENCH_CHILD"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", "c5", "--subs", "",
           "--no-rooflines", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--n-sampled", str(args.n_sampled), "--dim", str(args.dim),
           "--n-items", str(args.n_items), "--n-users", str(args.n_users), "--n-resample", str(args.n_resample)]
    if args.sharded_bags:
        cmd.append("--sharded-bags")
    if args.sharded_rep_tokens:
        cmd.append("--sharded-rep-tokens")
    if getattr(args, 'exchange', 'rows') != 'rows':
        cmd += ["--exchange", args.exchange]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    else:
        cmd += ["--cpu-seconds", str(args.cpu_seconds)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
    j = _child_detail(r.stdout)
    return {"value": j["value"], "unit": j["unit"], "n_gpus": 1, "ms_per_step": j["ms_per_step"],
            "steps": j["steps"], "warmup": j["warmup"], "cpu_baseline": j.get("cpu_baseline"),
            "what": "the same sharded step (arx.dist), table and per-GPU batch on ONE rank -- `python bench.py --gpus 1 "
                    "--workload c5` -- run by rank 0 after the N-rank job; efficiency = value / (N * anchor value)"}


LINE_LIMIT = 8000            # the driver keeps an 8 KB tail of stdout: the final line must fit in it whole
DETAIL_FILE = "bench_detail.json"


def _short(s, n=200):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _instep_ns(tag_part, needle):
    """Average in-step duration (ns) of the kernel whose name holds `needle`, from the newest committed
    rocprofv3 --kernel-trace --stats summary of this workload (profiles/rNN_<tag>_kernel_stats.csv)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.csv" % tag_part)))
    if not files:
        return None, None
    try:
        for row in csv.DictReader(open(files[-1])):
            if needle in row["Name"]:
                return float(row["AverageNs"]), "profiles/" + os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def _child_detail(stdout):
    """The full result of a child bench process: its BENCH_DETAIL line (the final '{' line is the compact one)."""
    lines = stdout.decode(errors="replace").splitlines()
    det = [l for l in lines if l.startswith("BENCH_DETAIL ")]
    if det:
        return json.loads(det[-1][len("BENCH_DETAIL "):])
    return json.loads([l for l in lines if l.startswith("{")][-1])


def compact_line(out):
    """The FINAL stdout line: every field of the bench contract + `roofline` of the step's DOMINANT pass (by time)
    + `cpu_baseline`, small enough for the driver's 8 KB tail (round-5 verdict: a 20 KB line left the driver's record
    unparsed).  Everything else (sub-results, per-kernel table, long workload prose) goes to bench_detail.json and
    to an EARLIER stdout line."""
    c = out.get("config", {})
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                       "scaling", "dtype", "data", "value_per_gpu", "scaling_efficiency_vs_anchor"))
    line["vs_baseline"] = out.get("vs_baseline")
    line["dtype_detail"] = _short(out.get("dtype_detail", ""), 220)
    line["config"] = _pick(c, ("batch", "n_sampled", "dim", "n_items", "n_users", "hipgraph", "pool_redraws_timed",
                               "hip_event_ms_per_step", "ms_per_step_min", "ms_per_step_max", "timed_regions",
                               "sampled_negative_logits_per_s", "final_loss", "parallelism", "world", "batch_per_gpu",
                               "global_batch", "exchange", "routing_in_timed_region", "ids_fed_from"))
    line["config"]["workload"] = _short(c.get("workload", ""), 260)
    # dominant pass of the step by time: K7 (HBM) / the hinge GEMM (MFMA) / the step's lookups (HBM)
    cands = [(k, out[k]) for k in ("roofline_hbm", "roofline", "roofline_gather")
             if isinstance(out.get(k), dict) and out[k].get("ms_per_launch")]
    if cands:
        dk, dom = max(cands, key=lambda kv: kv[1]["ms_per_launch"])
        rf = _pick(dom, ("bound", "achieved", "peak", "unit", "frac", "bytes_per_launch", "flops_per_launch",
                         "ms_per_launch", "ms_sorts", "ms_apply_finish", "frac_apply_finish_only", "unique_rows",
                         "contributions", "ms_in_step_rocprof", "frac_in_step", "in_step_source"))
        rf["traffic"] = dom.get("traffic")
        rf["kernel"] = _short(dom.get("kernel", ""), 160)
        if out.get("ms_per_step"):
            rf["share_of_step"] = dom["ms_per_launch"] / out["ms_per_step"]
        if out.get("traffic_source"):
            rf["traffic_source"] = out["traffic_source"]
        for k, short in (("roofline", "mfma"), ("roofline_hbm", "scatter"), ("roofline_gather", "gather")):
            if k != dk and isinstance(out.get(k), dict):
                e = _pick(out[k], ("bound", "achieved", "peak", "unit", "frac", "ms_per_launch", "bytes_per_launch",
                                   "flops_per_launch", "ms_in_step_rocprof", "frac_in_step", "in_step_source"))
                e["traffic"] = out[k].get("traffic")
                e["kernel"] = _short(out[k].get("kernel", ""), 120)
                rf[short] = e
        line["roofline"] = rf
    elif isinstance(out.get("roofline"), dict):
        line["roofline"] = out["roofline"]
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "ms_per_step", "timed_in"))
        line["cpu_baseline"]["sample"] = _short(cb.get("sample", ""), 320)
    sub = out.get("sub")
    if isinstance(sub, dict):
        line["sub_ms_per_step"] = {k: (round(v["ms_per_step"], 5) if isinstance(v, dict) and "ms_per_step" in v
                                       else None) for k, v in sub.items()}
    if isinstance(out.get("scaling_anchor"), dict):
        line["scaling_anchor"] = _pick(out["scaling_anchor"], ("value", "unit", "n_gpus", "ms_per_step", "error"))
    if isinstance(out.get("roofline_comm_predicted"), dict):
        line["roofline_comm_predicted"] = {k: (_short(v, 120) if isinstance(v, str) else v)
                                           for k, v in out["roofline_comm_predicted"].items()}
    line["detail"] = DETAIL_FILE
    txt = json.dumps(line)
    if len(txt) >= LINE_LIMIT:         # never let prose take the record down: drop the optional objects first
        for k in ("roofline_comm_predicted", "sub_ms_per_step", "scaling_anchor", "dtype_detail"):
            line.pop(k, None)
            txt = json.dumps(line)
            if len(txt) < LINE_LIMIT:
                break
    assert len(txt) < LINE_LIMIT, "bench line is %d bytes" % len(txt)
    return txt


def _print_line(out):
    # RCCL prints its version banner through C stdio: flush it out first so that the JSON line is the LAST
    # line on stdout
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    detail = json.dumps(out)
    for path in (os.path.join(ROOT, DETAIL_FILE), os.path.join(ROOT, "gpurun_out", DETAIL_FILE)):
        if os.environ.get("ARX_BENCH_CHILD"):
            break
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(detail + "\n")
        except OSError:
            pass
    print("BENCH_DETAIL " + detail, flush=True)        # the earlier line: everything; not a '{' line on purpose
    print(compact_line(out), flush=True)


def main_seq_hybrid(args, world, rank, local_rank):
    """`--workload c4h`: BASELINE configs[3] (C4: LSTM d = h = 64, L = 50, S = 1024) on `world` ranks with
    arx.dist.SeqHybridParallel -- every embedding table striped by row (all-to-all lookups, owner-side sparse
    updates), the LSTM weights data-parallel (one packed all-reduce), per-step pool gradients reduce-scattered for the
    clip norm.  args.lstm_batch sequences PER RANK (weak scaling).  One JSON line from rank 0: targets/s over all
    ranks, the max-over-ranks wall time, and the exchanges priced at link rate (roofline_comm_predicted: arithmetic --
    no multi-GPU box was in reach of the builder; N = 1 runs the same code path with every exchange a local copy)."""
    import torch.distributed as dist
    from arx import dist as arx_dist
    from arx.attributes.embed_attribute import EmbeddingAttribute
    from arx.lstm.seqModel import SeqModel
    from arx.utils.synthetic import SyntheticHMF
    one_gpu = bool(os.environ.get("ARX_DIST_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port())), ("RANK", "0"),
                     ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
    backend = os.environ.get("ARX_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    B, L, S, size = args.lstm_batch, 50, args.n_sampled, 64
    t0 = time.time()
    syn = SyntheticHMF(n_users=args.n_users, n_items=args.n_items, permute_logits=False, seed=0)   # the same on every rank
    syn.u_attr.set_model_size(size)
    syn.i_attr.set_model_size(size)
    START = args.n_items
    emb = EmbeddingAttribute(syn.u_attr, syn.i_attr, B, S, L, False, None, syn.logit_ind2item_ind)
    model = SeqModel([L], size, 1, 5.0, B, 0.5, 0.99, emb, loss='mw', use_concat=False, START_ID=START)
    emb.prepare_warp(syn.positives_csr(), syn.positives_csr())
    dp = arx_dist.SeqHybridParallel(model)
    rng = np.random.default_rng(1 + rank)                    # every rank its own sequences
    total = args.steps + args.warmup
    nb = min(total, 8)
    batches = []
    for _ in range(nb):
        users = rng.integers(0, args.n_users, size=B).astype(np.int32)
        tg = np.stack([syn.sample_batch(B, rng)[1] for _ in range(L)], 0).astype(np.int32)
        inp = np.concatenate([np.full((1, B), START, dtype=np.int32), tg[:-1]], 0)
        lens = rng.integers(10, L + 1, size=B)
        w = (np.arange(L)[:, None] < lens[None, :]).astype(np.float32)
        batches.append((torch.from_numpy(users).to(dev), torch.from_numpy(inp).to(dev),
                        torch.from_numpy(tg).to(dev), torch.from_numpy(w).to(dev), float(w.sum())))
    pool = torch.from_numpy(syn.sample_pool(S, np.random.default_rng(5))).to(dev)      # the same pool on every rank
    setup_s = time.time() - t0

    def run(k0, k1):
        tot = 0.0
        for k in range(k0, k1):
            u, i, t, w, ws = batches[k % nb]
            model.step_async(None, u, i, t, w, 0, pool if (k == 0 or k == args.warmup) else None, None)
            tot += ws
        return tot
    run(0, args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.time()
    tot_w = run(args.warmup, total)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.time() - t1
    red = torch.tensor([wall, -wall, tot_w], dtype=torch.float64, device=dev)
    mx = red[:2].clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = red[2:].clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    wall, targets = float(mx[0].item()), float(sm[0].item())
    dense_bytes = sum(p.w.numel() * 4 for p in model.rt.dense.values())
    dist.destroy_process_group()
    if rank != 0:
        return 0
    pred = arx_dist.comm_prediction('seq_hybrid', world, B, S, size, L=L, dense_bytes=dense_bytes)
    pred8 = arx_dist.comm_prediction('seq_hybrid', 8, B, S, size, L=L, dense_bytes=dense_bytes)
    dp8 = arx_dist.comm_prediction('seq_dp', 8, B, S, size, L=L, dense_bytes=dense_bytes)
    out = {"metric": METRIC, "value": targets / wall, "unit": "targets/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "dtype_detail": DTYPE_DETAIL, "data": "synthetic",
           "value_per_gpu": targets / wall / world,
           "config": {"workload": "C4 hybrid (BASELINE configs[3] on %d rank(s)): LSTM d=h=%d, L=%d, %d sequences per "
                                  "rank, %d items, S=%d, 'mw', clip 5.0, Adagrad; tables striped by row "
                                  "(arx.dist.SeqHybridParallel: all-to-all lookups, owner-side K7), LSTM weights "
                                  "all-reduced; eager step, routing on the host per batch (inside the timed region)"
                                  % (world, size, L, B, args.n_items, S),
                      "batch_per_gpu": B, "global_batch": B * world, "n_sampled": S, "dim": size,
                      "parallelism": "row-striped tables x dp%d" % world, "routing_in_timed_region": True,
                      "timestep_rows_per_s": L * B * world * args.steps / wall, "setup_s": setup_s},
           "roofline_comm_predicted": {"this_run": pred, "at_8_ranks": pred8, "seq_data_parallel_at_8_ranks": dp8},
           "cpu_baseline": {"value": None, "unit": "targets/s", "cores": None, "kind": "port", "sample": None,
                            "why": "the sequence model's CPU baseline is not timed (the headline line's is the HMF step)"}}
    _print_line(out)
    return 0


def main_sharded(args, world, rank, local_rank):
    """BASELINE configs[4] (C5) on `world` ranks: one JSON line from rank 0, with the N = 1 anchor beside it."""
    import torch.distributed as dist
    from arx import dist as arx_dist
    if world == 1:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port())), ("RANK", "0"),
                     ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
    out = arx_dist.bench_run(args, world, rank, local_rank)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank != 0:
        return 0
    out["dtype_detail"] = DTYPE_DETAIL
    out["value_per_gpu"] = out["value"] / world
    if world > 1 and not args.no_anchor:
        torch.cuda.empty_cache()
        try:
            out["scaling_anchor"] = scaling_anchor(args)
            out["scaling_efficiency_vs_anchor"] = out["value"] / (world * out["scaling_anchor"]["value"])
            # the contract times the CPU baseline on rank 0 at N = 1 only: the anchor IS that run -- its baseline is
            # the one this line carries (round 4 verdict: the N > 1 line had value null)
            cb = out["scaling_anchor"].pop("cpu_baseline", None)
            if cb and cb.get("value") is not None:
                cb = dict(cb)
                cb["timed_in"] = "the N = 1 anchor of this run (rank 0's child process, same box)"
                out["cpu_baseline"] = cb
        except Exception as e:
            out["scaling_anchor"] = {"error": "%s: %s" % (type(e).__name__, e)}
    elif world == 1:
        out["scaling_anchor"] = {"value": out["value"], "unit": out["unit"], "n_gpus": 1,
                                 "ms_per_step": out["ms_per_step"], "what": "this line IS the anchor"}
        if not args.no_cpu_baseline:
            # the reference algorithm is dense over the whole table (full-table scorer GEMM, dense Adagrad):
            # at 100 M items one step is ~100 GB of host traffic; the bounded sample runs it on a 1 M-item table
            import copy
            from arx.utils.synthetic import SyntheticHMF
            a = copy.copy(args)
            a.n_items = min(args.n_items, 1000000)
            syn = SyntheticHMF(n_users=a.n_users, n_items=a.n_items, permute_logits=False, seed=0,
                               zipf_items=a.zipf_items)
            out["cpu_baseline"] = cpu_baseline(a, syn, "C5 (on a %d-item table)" % a.n_items)
    _print_line(out)
    return 0


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        return self_launch(args)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "c4h":
        if args.n_items is None:
            args.n_items = 1000000
        return main_seq_hybrid(args, world, rank, local_rank)
    if world > 1 or args.workload == "c5":
        if args.n_items is None:
            # configs[4]: 100 M-item dim-128 table, row-sharded (with --sharded-bags: 1 M items, 20 tokens each)
            args.n_items = 1000000 if (args.sharded_bags or args.sharded_rep_tokens) else 100000000
        return main_sharded(args, world, rank, local_rank)
    if args.n_items is None:
        args.n_items = 1000000
    if args.mulhot:
        args.workload = "c3"
    torch.cuda.set_device(0)
    head = run_hmf(args, args.workload, args.steps, args.warmup)
    out = {
        "metric": METRIC, "value": head["value"], "unit": head["unit"], "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "dtype_detail": DTYPE_DETAIL, "data": "synthetic", "config": head["config"],
    }
    for k in ("roofline", "roofline_hbm", "roofline_gather", "traffic_source", "kernels_ms", "kernels",
              "cpu_baseline"):
        if k in head:
            out[k] = head[k]
    subs = [s for s in args.subs.split(",") if s and s != args.workload]
    sub = {}
    _quiesce()
    for s in subs:
        try:
            if s in WORKLOADS:
                r = run_hmf(args, s, args.sub_steps, min(args.warmup, 10))
                r.pop("kernels", None)
            elif s.endswith("mce") and s[:-3] in WORKLOADS:      # C2 / C3 with the sampled softmax (round 6: fused at d = 128)
                r = run_hmf(args, s[:-3], args.sub_steps, min(args.warmup, 10), loss='mce')
                r.pop("kernels", None)
            elif s.endswith("host") and s[:-4] in WORKLOADS:     # the step with its ids handed over as host arrays
                r = run_hmf(args, s[:-4], args.sub_steps, min(args.warmup, 10), host_feed=True)
                r.pop("kernels", None)
            elif s in ("c4", "c4mce"):
                r = run_lstm(args, 'mce' if s == "c4mce" else 'mw', min(args.sub_steps, 30), 5)
            elif s == "c4host":
                r = run_lstm(args, 'mw', min(args.sub_steps, 30), 5, host_feed=True)
                r.pop("roofline", None)
            elif s == "k1":
                r = k1_past_llc(torch.device('cuda', 0), args.dim)
                # the gather's headline is this PHYSICAL measurement (2 GB table, 8x the LLC); the step's own
                # LLC-resident lookup launch stays beside it as `in_step`
                if "error" not in r:
                    ent = dict(r)
                    if "roofline_gather" in out:
                        ent["in_step"] = out["roofline_gather"]
                    out["roofline_gather"] = ent
            elif s == "c5w1":
                r = run_sharded_world1(args)
            elif s == "c3repw1":
                r = run_sharded_world1(args, rep_tokens=True)
            elif s == "topk":
                r = run_topk(args)
            elif s.endswith("_f32mfma") and (s[:-8] in WORKLOADS or s[:-8] == "c5w1"):
                r = run_f32mfma(args, s[:-8])
            else:
                continue
            sub[s] = r
        except Exception as e:      # a sub-result must never take the headline line down
            sub[s] = {"error": "%s: %s" % (type(e).__name__, e)}
        _quiesce()
    if sub:
        out["sub"] = sub
    if not args.no_cpu_baseline:
        # LAST: its 128 BLAS threads keep the host busy for a while after they return, and the
        # eager (host-paced) sub-results measured 2x slower behind it (c5w1: 0.86 vs 0.43 ms/step)
        from arx.utils.synthetic import SyntheticHMF
        syn = SyntheticHMF(n_users=args.n_users, n_items=args.n_items, permute_logits=False, seed=0,
                           zipf_items=args.zipf_items, **WORKLOADS[args.workload][1])
        out["cpu_baseline"] = cpu_baseline(args, syn, args.workload.upper())
    _print_line(out)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
