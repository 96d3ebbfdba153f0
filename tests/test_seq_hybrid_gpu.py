"""Hybrid sequence model (arx.dist.SeqHybridParallel, round 6): embedding tables striped by ROW over the ranks
(owner = row % world), LSTM / input-projection weights data-parallel.  `world` ranks, each fed 1/world of the
sequences, must reproduce the single-process oracle step (oracle.ref_lstm.RefSeqModel) on the GLOBAL batch: loss,
TF-1.0 clip_by_global_norm (inactive and ACTIVE), every dense weight on every rank, and the striped tables put back
together (global_params) -- rows touched from several ranks in one step included.  The ranks are processes sharing
the one GPU of the test box and exchange over gloo (host-staged all-to-all; bench / production: RCCL).
Reference: lstm/run.py:87,221-229 (its only device split), lstm/seqModel.py:87-126,173-182."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_ID = dict(n_users=301, n_items=503, logit_size=503)        # (odd sizes: ragged last stripes)


def _worker(rank, world, port, out_dir, loss, clip, use_concat):
    for p in (ROOT, os.path.join(ROOT, "a-recsys_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from arx.dist import SeqHybridParallel
    from test_lstm_gpu import _build, _batch, RTOL

    size, B_loc, L, S = 64, 16, 5, 128
    B = B_loc * world
    syn, emb, model, _, _ = _build(CFG_ID, loss, size, B_loc, L, S, clip, seed=4, use_concat=use_concat)
    _, _, _, remb, ref = _build(CFG_ID, loss, size, B, L, S, clip, seed=4, use_concat=use_concat)   # the oracle: global batch
    full_rows = {k: v.shape[0] for k, v in emb.get_params().items()}
    dp = SeqHybridParallel(model)
    for t in emb.tables.values():                                     # no rank holds a whole table
        assert t.E.shape[0] == (full_rows[t.name] + world - 1) // world + 1
    rng = np.random.default_rng(7)
    pool = syn.sample_pool(S, rng)
    id2idx = {int(v): i for i, v in enumerate(pool)}
    sl = slice(rank * B_loc, (rank + 1) * B_loc)
    for step in range(3):
        users, inp, tg, w = _batch(syn, rng, L, B)                    # the same global batch on every rank
        if step == 1:
            tg[:, :] = tg[:, :1]                                      # every rank hits the SAME target rows
            inp[1:] = tg[:-1]
        ps = pool if step == 0 else None
        l_ref = ref.step(list(users), inp.tolist(), tg.tolist(), w.tolist(), ps, id2idx)
        l_loc = model.step(None, list(users[sl]), inp[:, sl].tolist(), tg[:, sl].tolist(), w[:, sl].tolist(), 0,
                           ps, id2idx)
        np.testing.assert_allclose(dp.global_loss(l_loc), l_ref, rtol=RTOL, err_msg='loss step %d' % step)
        np.testing.assert_allclose(float(model._gnorm.item()), ref.last['gnorm'], rtol=RTOL,
                                   err_msg='global norm step %d' % step)
        got = dp.global_params()
        for k, v in got.items():
            np.testing.assert_allclose(v, remb.params[k], rtol=RTOL, atol=3e-6, err_msg='%s step %d' % (k, step))
        np.testing.assert_allclose(model.W.w.cpu().numpy(), ref.W, rtol=RTOL, atol=3e-6, err_msg='lstm_w')
        np.testing.assert_allclose(model.b.w.cpu().numpy(), ref.b, rtol=RTOL, atol=3e-6, err_msg='lstm_b')
        if use_concat:
            np.testing.assert_allclose(model.Wi.w.cpu().numpy(), remb.params['w_input_item'], rtol=RTOL, atol=3e-6)
    for t in emb.tables.values():                                     # the padding row never moved
        assert not t.E[t.shard['zero_row']].any()
    slots = dp.global_params(slots=True)
    for k, v in slots.items():
        np.testing.assert_allclose(v, remb.slots[k], rtol=RTOL, atol=3e-6, err_msg='slot ' + k)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,loss,clip,use_concat", [(2, 'mw', 5.0, False), (2, 'mw', 0.5, True),
                                                         (3, 'mce', 0.5, False), (1, 'mw', 5.0, False)])
def test_seq_hybrid_matches_global_oracle(dev, tmp_path, world, loss, clip, use_concat):
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 300) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), loss, clip, use_concat), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))
