"""Golden batches of the REAL reference iterator word2vec/data_iterator.py (imports under py3),
seeded through numpy's global RandomState.  Build container only:

    python tests/golden/make_w2v_iterator.py   ->  tests/golden/w2v_iterator.json
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("ARX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "word2vec"))
import data_iterator as ref_di      # noqa: E402


def make_seq(seed, n_users, end_ind):
    rs = np.random.RandomState(seed)
    seq = []
    for u in range(n_users):
        for _ in range(rs.randint(1, 9)):
            seq.append((u, int(rs.randint(0, 40))))
        seq.append((u, end_ind))
    return seq


def main():
    out = []
    end_ind = 40
    for case, (gen, seed, batch, skips, window, sequence, nb) in enumerate([
            ('get_next', 1, 16, 2, 2, False, 5), ('get_next', 2, 8, 1, 1, True, 4),
            ('get_next_sg', 3, 16, 2, 2, False, 5), ('get_next_sg', 4, 12, 3, 3, True, 4),
            ('get_next_cbow', 5, 16, 3, 4, False, 5), ('get_next_cbow', 6, 8, 2, 2, False, 6)]):
        seq = make_seq(100 + case, 25, end_ind)
        it = ref_di.DataIterator(seq, end_ind, batch, skips, window, sequence)
        np.random.seed(seed)
        g = getattr(it, gen)()
        batches = []
        for _ in range(nb):
            u, i, o = next(g)
            batches.append({"users": np.asarray(u).tolist(), "inputs": np.asarray(i).tolist(),
                            "outputs": np.asarray(o).tolist()})
        out.append({"gen": gen, "seed": seed, "batch": batch, "n_skips": skips, "window": window,
                    "sequence": sequence, "seq": seq, "end_ind": end_ind, "batches": batches,
                    "index_after": int(it.index)})
    with open(os.path.join(HERE, "w2v_iterator.json"), "w") as f:
        json.dump(out, f)
    print("cases", len(out))


if __name__ == "__main__":
    main()
