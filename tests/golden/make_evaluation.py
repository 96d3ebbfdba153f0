"""Golden outputs of the REAL reference evaluation harness (utils/evaluate.py, submit.py -- both
import under py3) on tests/golden/ml1m_small.  Build container only:

    python tests/golden/make_evaluation.py  ->  tests/golden/evaluation.json
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

REF = os.environ.get("ARX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "utils"))
import evaluate as ref_ev      # noqa: E402


def make_rec(uids, seed):
    rs = np.random.RandomState(seed)
    return {int(u): [int(x) for x in rs.permutation(3952)[:30] + 1] for u in uids}


def main():
    out = {}
    for test in (False, True):
        d = tempfile.mkdtemp()
        for f in os.listdir(os.path.join(HERE, 'ml1m_small')):
            shutil.copy(os.path.join(HERE, 'ml1m_small', f), d)
        ev = ref_ev.Evaluation(d, test=test)
        files = {}
        for f in ('historical_train.csv', 'res_T.csv', 'res_T_test.csv', 'historical_train_test.csv'):
            files[f] = open(os.path.join(d, f), 'rb').read().decode('latin-1')
        rec = make_rec(ev.get_uids(), 7)
        # seed the recommendations with some true items so that the scores are not all ~0
        for k, u in enumerate(ev.get_uids()):
            if k % 3 == 0 and ev.T[u]:
                rec[u][k % 5] = int(ev.T[u][0])
        ev.eval_on(rec)
        s_self, s_ex = ev.get_scores()
        out[str(test)] = {"files": files, "uids": [int(u) for u in ev.get_uids()],
                          "uinds": [int(u) for u in ev.get_uinds()],
                          "T": {str(k): v for k, v in ev.T.items()},
                          "hist_len": {str(k): len(v) for k, v in ev.hist.items()},
                          "rec": {str(k): v for k, v in rec.items()},
                          "s_self": [float(x) for x in s_self], "s_ex": [float(x) for x in s_ex]}
        shutil.rmtree(d)
    with open(os.path.join(HERE, "evaluation.json"), "w") as f:
        json.dump(out, f)
    print("ok", out["False"]["s_self"][:5])


if __name__ == "__main__":
    main()
