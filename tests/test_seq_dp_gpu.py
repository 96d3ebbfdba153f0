"""Data-parallel sequence model (arx.dist.SeqDataParallel) on the real HIP path: `world` replicas of
SeqModel, each fed 1/world of the sequences, must reproduce the single-process oracle step
(oracle.ref_lstm.RefSeqModel) on the GLOBAL batch -- loss, TF-1.0 clip_by_global_norm and every
updated table / dense weight.  The replicas are processes sharing the one GPU of the test box and
exchange over gloo (the collectives are backend-agnostic; bench / production use RCCL)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_ID = dict(n_users=300, n_items=500, logit_size=500)


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
    from arx.dist import SeqDataParallel
    from test_lstm_gpu import _build, _batch, _compare, RTOL

    size, B_loc, L, S = 64, 16, 5, 128
    B = B_loc * world
    syn, emb, model, _, _ = _build(CFG_ID, loss, size, B_loc, L, S, clip, seed=4, use_concat=use_concat)
    _, _, _, remb, ref = _build(CFG_ID, loss, size, B, L, S, clip, seed=4, use_concat=use_concat)   # the oracle: global batch
    dp = SeqDataParallel(model)
    rng = np.random.default_rng(7)
    pool = syn.sample_pool(S, rng) if loss == 'mw' else None
    id2idx = {int(v): i for i, v in enumerate(pool)} if pool is not None else None
    sl = slice(rank * B_loc, (rank + 1) * B_loc)
    for step in range(3):
        users, inp, tg, w = _batch(syn, rng, L, B)                    # the same global batch on every replica
        ps = pool if step == 0 else None
        l_ref = ref.step(list(users), inp.tolist(), tg.tolist(), w.tolist(), ps, id2idx)
        l_loc = model.step(None, list(users[sl]), inp[:, sl].tolist(), tg[:, sl].tolist(), w[:, sl].tolist(), 0,
                           ps, id2idx)
        np.testing.assert_allclose(dp.global_loss(l_loc), l_ref, rtol=RTOL, err_msg='loss step %d' % step)
        np.testing.assert_allclose(float(model._gnorm.item()), ref.last['gnorm'], rtol=RTOL,
                                   err_msg='global norm step %d' % step)
        _compare(emb, model, remb, ref)                               # EVERY replica holds the global update
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,loss,clip,use_concat", [(2, 'mw', 5.0, False), (2, 'mw', 0.5, True),
                                                         (2, 'ce', 5.0, False), (1, 'mw', 5.0, False)])
def test_seq_data_parallel_matches_global_oracle(dev, tmp_path, world, loss, clip, use_concat):
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 300) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), loss, clip, use_concat), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))
