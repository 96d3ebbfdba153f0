"""Generates tests/golden/*.json from the REAL reference helpers that are importable
under Python 3: utils/prepare_train.py (item_frequency, sample_items,
positive_items), attributes/attribute.py (Attributes) and utils/eval_metrics.py.
Run in the build container only (needs /root/reference); the outputs are data --
inputs + expected outputs -- and are committed; the reference source is not.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("ARX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "utils"))
sys.path.insert(0, os.path.join(REF, "attributes"))

import prepare_train as ref_pt      # noqa: E402
import attribute as ref_attr        # noqa: E402
import eval_metrics as ref_em       # noqa: E402


def main():
    rng = np.random.RandomState(7)
    out = {}
    # ---- item_frequency / positive_items on a small interaction log ----
    data_tr = [(int(u), int(i), int(t)) for u, i, t in
               zip(rng.randint(0, 30, 400), rng.zipf(1.6, 400) % 50, rng.randint(0, 1000, 400))]
    data_va = [(int(u), int(i), int(t)) for u, i, t in
               zip(rng.randint(0, 30, 80), rng.zipf(1.6, 80) % 50, rng.randint(0, 1000, 80))]
    cases = []
    for power in (0.5, 1.0, 0.0):
        pop, p = ref_pt.item_frequency(data_tr, power)
        cases.append({"power": power, "item_population": [int(x) for x in pop],
                      "p_item": [float(x) for x in p]})
    out["item_frequency"] = {"data_tr": data_tr, "cases": cases}
    pos, pos_va = ref_pt.positive_items(data_tr, data_va)
    out["positive_items"] = {"data_va": data_va,
                             "train": {str(k): sorted(int(x) for x in v) for k, v in pos.items()},
                             "valid": {str(k): sorted(int(x) for x in v) for k, v in pos_va.items()}}
    # ---- sample_items: legacy RandomState stream is frozen -> reproducible draws ----
    pop, p = ref_pt.item_frequency(data_tr, 0.5)
    samples = []
    for seed, n in ((0, 8), (1, 16), (5, len(pop))):
        np.random.seed(seed)
        s, id2idx = ref_pt.sample_items(pop, n, p)
        samples.append({"seed": seed, "n": n, "sampled": [int(x) for x in s],
                        "id2idx": {str(int(k)): int(v) for k, v in id2idx.items()}})
    np.random.seed(3)
    s, id2idx = ref_pt.sample_items(list(range(40)), 10)
    samples.append({"seed": 3, "n": 10, "uniform_over": 40, "sampled": [int(x) for x in s],
                    "id2idx": {str(int(k)): int(v) for k, v in id2idx.items()}})
    out["sample_items"] = samples
    # ---- Attributes container semantics ----
    a = ref_attr.Attributes(2, [[2, 3, 1], [4, 2, 1]], 1, [[5, 6, 7, 1]], [2], [[0, 2, 3, 4]],
                            [[2, 1, 1]], [5, 6], [9])
    a.set_model_size(12)
    st1 = {"cat": list(a._embedding_size_list_cat), "mulhot": list(a._embedding_size_list_mulhot)}
    a.set_model_size([3, 4])
    a.set_model_size([7], 1)
    st2 = {"cat": list(a._embedding_size_list_cat), "mulhot": list(a._embedding_size_list_mulhot)}
    a.set_target_prediction([[1]], [[2]], [[3]], [[4.0]])
    out["attributes"] = {"after_int": st1, "after_lists": st2,
                         "num_features_cat": a.num_features_cat,
                         "num_features_mulhot": a.num_features_mulhot,
                         "v_cat": list(a._embedding_classes_list_cat),
                         "v_mulhot": list(a._embedding_classes_list_mulhot),
                         "full": [a.full_cat_tr, a.full_values_tr, a.full_segids_tr, a.full_lengths_tr]}
    # ---- ranking metrics (next #4) ----
    R = {u: [int(x) for x in rng.permutation(60)[:30]] for u in range(6)}
    T = {u: [int(x) for x in rng.permutation(60)[:rng.randint(1, 8)]] for u in range(6)}
    res = ref_em.metrics(R, T)
    out["eval_metrics"] = {"R": {str(k): v for k, v in R.items()}, "T": {str(k): v for k, v in T.items()},
                           "result": {k: [float(x) for x in v] for k, v in res.items()}}
    with open(os.path.join(HERE, "reference_helpers.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "reference_helpers.json"))


if __name__ == "__main__":
    main()
