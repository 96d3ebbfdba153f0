"""Host logic of the static token order (arx.ops.BagCSC, csrc/csc.hip): the token-major order of a multi-hot
feature's (token, entity) pairs, built once from the feature CSR of attributes/attribute.py
(embed_attribute.py:265-318 uploads that CSR once and never changes it) -- against a plain loop restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-recsys_amd"))


def _brute(vals, starts, lens, max_len, rows):
    pairs = []
    for r in range(len(lens)):
        for j in range(min(int(lens[r]), max_len)):
            p = int(starts[r]) + j
            if 0 <= p < len(vals) and 0 <= vals[p] < rows:
                pairs.append((int(vals[p]), r, j, p))
    pairs.sort(key=lambda x: (x[0], x[1], x[2]))          # by token; inside a token by (entity, position in bag)
    return pairs


@pytest.mark.parametrize("n_ent,rows,max_len,permute", [(1000, 50, 12, True), (300, 70000, 40, False), (5, 3, 2, False),
                                                        (64, 9, 30, True)])
def test_bag_csc_matches_loop(n_ent, rows, max_len, permute):
    import torch
    from arx import ops
    rng = np.random.default_rng(n_ent + rows)
    lens = rng.integers(0, max_len + 4, size=n_ent).astype(np.int32)        # some bags longer than max_len
    starts = (np.cumsum(lens) - lens).astype(np.int32)
    vals = rng.integers(-1, rows + 2, size=int(lens.sum()) + 3).astype(np.int32)   # some tokens out of range
    if permute:                                         # bag index re-indexed by a table row (graph.py _rider_csr)
        perm = rng.permutation(n_ent)
        st_r, ln_r = np.zeros(n_ent, np.int32), np.zeros(n_ent, np.int32)
        st_r[perm], ln_r[perm] = starts, lens
        starts, lens = st_r, ln_r
    cs = ops.BagCSC(torch.from_numpy(vals), torch.from_numpy(starts), torch.from_numpy(lens), max_len, rows)
    want = _brute(vals, starts, lens, max_len, rows)
    assert cs.ok and cs.nq == len(want)
    qte, qpos = cs.qte.numpy(), cs.qpos.numpy()
    for q, (t, r, _j, p) in enumerate(want):
        assert qte[q, 0] == t and qte[q, 1] == r and qpos[p] == q
    assert int((qpos >= 0).sum()) == cs.nq


def test_bag_csc_rejects_shared_positions():
    """Two bags over the same CSR positions have no place of their own: the caller keeps the per-step sort."""
    import torch
    from arx import ops
    vals = torch.tensor([1, 2, 3, 0], dtype=torch.int32)
    starts = torch.tensor([0, 1], dtype=torch.int32)
    lens = torch.tensor([3, 3], dtype=torch.int32)
    assert not ops.BagCSC(vals, starts, lens, 8, 5).ok
