"""Evaluation harness of the runners -- py3 mirror of utils/evaluate.py (class Evaluation):
builds the ground-truth / history files from the interaction logs once, then scores a
recommendation dict {user id: [item ids]} with utils/eval_metrics.py, as is and with the user's
training history filtered out."""
from __future__ import annotations

from os.path import isfile, join

from .eval_metrics import metrics
from .load_data import load_interactions, load_items, load_users
from .submit import combine_sub, format_submit, load_submit


class Evaluation(object):
    def __init__(self, raw_data_dir, test=False):
        res_filename = 'res_T_test.csv' if test else 'res_T.csv'
        if not isfile(join(raw_data_dir, res_filename)):
            print('eval file does not exist. creating ... ')
            self.create_eval_file(raw_data_dir)
        self.T = load_submit(res_filename, submit_dir=raw_data_dir)
        hist_filename = 'historical_train_test.csv' if test else 'historical_train.csv'
        self.hist = load_submit(hist_filename, submit_dir=raw_data_dir)
        self.Iatt, _, self.Iid2ind = load_items(raw_data_dir)
        self.Uatt, _, self.Uid2ind = load_users(raw_data_dir)
        self.Uids = self.get_uids()
        self.Uinds = [self.Uid2ind[v] for v in self.Uids]
        self.combine_sub = combine_sub

    def get_user_n(self):
        return len(self.Uinds)

    def get_uids(self):
        return list(self.T.keys())

    def get_uinds(self):
        return self.Uinds

    def set_uinds(self, uinds):
        self.Uinds = uinds

    def eval_on(self, rec):
        """evaluate.py:35-56: `rec` values are turned into strings in place; scores = the
        flattened metric table (prec, recall, map, ndcg at 2/5/10/20/30)."""
        self.res = rec
        for k in rec:
            rec[k] = [str(v) for v in rec[k]]
        r_ex = self.combine_sub(self.hist, rec, 1, users=self.Uatt)
        self.s_self = [x for row in metrics(rec, self.T).values() for x in row]
        self.s_ex = [x for row in metrics(r_ex, self.T).values() for x in row]

    def get_scores(self):
        return self.s_self, self.s_ex

    @staticmethod
    def _by_user(log, with_time):
        seqs = {}
        for row in log:
            seqs.setdefault(row[0], []).append((row[1], row[2]) if with_time else row[1])
        return seqs

    def create_eval_file(self, raw_data):
        """evaluate.py:64-114: historical_train.csv (training items, newest first), res_T.csv
        (validation items, newest first), res_T_test.csv (test items in log order) and
        historical_train_test.csv (validation + training history)."""
        (tr, va, te), _ = load_interactions(data_dir=raw_data)
        newest_first = lambda v: ','.join(str(p[0]) for p in sorted(v, key=lambda x: x[1], reverse=True))
        seq_tr = {u: newest_first(v) for u, v in self._by_user(tr, True).items()}
        seq_va = {u: newest_first(v) for u, v in self._by_user(va, True).items()}
        seq_te = {u: ','.join(str(p) for p in v) for u, v in self._by_user(te, False).items()}
        format_submit(seq_tr, 'historical_train.csv', submit_dir=raw_data)
        format_submit(seq_va, 'res_T.csv', submit_dir=raw_data)
        format_submit(seq_te, 'res_T_test.csv', submit_dir=raw_data)
        seq_va_tr = seq_va                     # (aliasing as in the reference: res_T is already written)
        for u in seq_tr:
            if u in seq_va:
                seq_va_tr[u] = seq_va[u] + ',' + seq_tr[u]
        format_submit(seq_va_tr, 'historical_train_test.csv', submit_dir=raw_data)
