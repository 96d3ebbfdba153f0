"""Recommendation / ground-truth files of the evaluation harness -- py3 mirror of utils/submit.py
(load_submit, format_submit, combine_sub).  File format: tab-separated, header `user_id  items`,
`items` a comma-joined id list (an empty cell = no items)."""
from __future__ import annotations

from os.path import join

import pandas as pd


def load_submit(sub_id, submit_dir='../submissions/'):
    """submit.py:5-18 -> {user_id: [item id strings]}."""
    data = pd.read_csv(join(submit_dir, sub_id), delimiter='\t', header=0)
    out = {}
    for uid, cell in zip(data['user_id'].tolist(), data['items'].tolist()):
        if isinstance(cell, str):
            out[uid] = cell.split(',')
        elif isinstance(cell, int):
            out[uid] = [str(cell)]               # a single id is parsed as a number
        else:
            out[uid] = []                         # empty cell (NaN)
    return out


def format_submit(X, sub_id, submit_dir='../submissions/'):
    """submit.py:20-38: write {user: [items] | 'a,b,c'} in dict order.  Like the reference this
    joins list values IN PLACE (it stops at the first value that is not a list)."""
    for key in X:
        if not isinstance(X[key], list):
            break
        X[key] = ','.join(str(v) for v in X[key])
    frame = pd.DataFrame(list(X.items()))
    frame.to_csv(path_or_buf=join(submit_dir, sub_id), sep='\t', index=False,
                 header=['user_id', 'items'])


def combine_sub(r1, r2, opt=0, users=None):
    """submit.py:42-61: per user (rows of the users table, first column = id) the items of r1
    followed by the items of r2 not already seen; with opt != 0 the r1 items only BLOCK
    (used to exclude a user's history from a recommendation list)."""
    rec = {}
    for i in range(len(users)):
        uid = users[i, 0]
        if uid not in r1 and uid not in r2:
            continue
        seen, out = set(), []
        for iid in r1.get(uid, ()):
            if iid not in seen:
                seen.add(iid)
                if opt == 0:
                    out.append(iid)
        for iid in r2.get(uid, ()):
            if iid not in seen:
                seen.add(iid)
                out.append(iid)
        rec[uid] = out
    return rec
