"""Raw CSV loader -- py3 mirror of the reference's utils/load_data.py (same function names and
return shapes), the entry of the preprocessing path.

Files in `data_dir` (load_data.py:30-71): `u.csv` / `i.csv` -- tab separated with a header, first
column = raw id, remaining columns = attributes; optional `u_attr.csv` / `i_attr.csv` -- a header
plus ONE row giving each column's type (0 categorical, 1 multi-hot, >1 ignored); `obs_tr.csv`,
`obs_va.csv`, `obs_te.csv` -- `user  item [time [...]]` rows of raw ids.

Differences from the reference, none of them visible in the returned values:
  * files are decoded as Latin-1 (the Python-2 reference read bytes; the ML-1m example `i.csv`
    is not UTF-8), every byte string round-trips;
  * raw id -> row index mapping of the interaction logs is one vectorised lookup instead of a
    Python loop over rows (load_data.py:77-80);
  * a missing file raises FileNotFoundError instead of `exit(1)` / an attribute error on `[]`.
"""
from __future__ import annotations

from os.path import isfile, join

import numpy as np
import pandas as pd

ENCODING = 'latin-1'


def build_index(values):
    """load_data.py:5-15: {raw id (first column) -> row number}; a repeated id keeps its LAST row."""
    values = np.asarray(values, dtype=object)
    keys = values[:, 0] if values.ndim == 2 else values
    return {k: n for n, k in enumerate(keys.tolist())}


def load_csv(filename, indexing=True, sep='\t', header=0):
    """load_data.py:16-28 -> (values [rows, cols] object array, column names[, index])."""
    if not isfile(filename):
        raise FileNotFoundError(filename)
    data = pd.read_csv(filename, delimiter=sep, header=header, encoding=ENCODING)
    values = data.values
    columns = list(data.columns)
    if indexing:
        return values, columns, build_index(values)
    return values, columns


def _load_entities(data_dir, stem):
    values, names, index = load_csv(join(data_dir, stem + '.csv'))
    tfile = join(data_dir, stem + '_attr.csv')
    if isfile(tfile):
        vals, _ = load_csv(tfile, False)
        types = [int(v) for v in np.asarray(vals).flatten().tolist()]
    else:
        types = [0] * len(names)                       # load_data.py:42-43
    return values, (names, types), index


def load_users(data_dir, sep='\t'):
    """load_data.py:36-45 -> (users [N_u, F] object array, (attr names, attr types), {raw id: row})."""
    return _load_entities(data_dir, 'u')


def load_items(data_dir, sep='\t'):
    """load_data.py:47-56."""
    return _load_entities(data_dir, 'i')


def load_interactions(data_dir, sep='\t'):
    """load_data.py:58-72 -> ([tr, va, te] arrays with >= 3 columns (a zero time column is added
    to two-column logs), column names of the training log).  A split whose file is absent comes
    back as an empty [0, 3] array (the reference fails on it)."""
    ints, names = [], []
    for s in ('tr.csv', 'va.csv', 'te.csv'):
        fn = join(data_dir, 'obs_' + s)
        if not isfile(fn):
            ints.append(np.zeros((0, 3), dtype=np.int64))
            names.append(None)
            continue
        a, name = load_csv(fn, False)
        if a.shape[1] < 2:
            raise ValueError("%s: need at least the columns user, item" % fn)
        if a.shape[1] == 2:
            a = np.append(a, np.zeros((a.shape[0], 1), dtype=int), 1)
        ints.append(a)
        names.append(name)
    return ints, names[0]


def _reindex(a, user_index, item_index, what):
    """Columns 0/1: raw ids -> row numbers of u.csv / i.csv (load_data.py:77-80)."""
    if a.shape[0] == 0:
        return a
    u = pd.Series(a[:, 0]).map(user_index)
    i = pd.Series(a[:, 1]).map(item_index)
    if u.isna().any() or i.isna().any():
        bad = int(np.flatnonzero((u.isna() | i.isna()).values)[0])
        raise KeyError("%s row %d: user %r / item %r not in u.csv / i.csv" % (what, bad, a[bad, 0], a[bad, 1]))
    out = np.array(a, dtype=object) if a.dtype == object else a.copy()
    out[:, 0] = u.values.astype(np.int64)
    out[:, 1] = i.values.astype(np.int64)
    return out


def _triples(a):
    return list(zip(a[:, 0].tolist(), a[:, 1].tolist(), a[:, 2].tolist()))


def load_raw_data(data_dir, _submit=0):
    """load_data.py:73-96 -> (users, items, data_tr, data_va, (u names, u types),
    (i names, i types), user_index, item_index); data_* are lists of (user row, item row, time).
    `_submit=1`: train on tr+va, validate on te."""
    users, u_attr, user_index = load_users(data_dir)
    items, i_attr, item_index = load_items(data_dir)
    ints, _ = load_interactions(data_dir)
    tr, va, te = [_reindex(a, user_index, item_index, 'obs_' + s)
                  for a, s in zip(ints, ('tr', 'va', 'te'))]
    if tr.shape[0] == 0:
        raise FileNotFoundError(join(data_dir, 'obs_tr.csv'))
    if _submit == 1:
        tr = np.append(tr, va[:, :tr.shape[1]], 0) if va.shape[0] else tr
        data_tr, data_va = _triples(tr), _triples(te)
    else:
        data_tr, data_va = _triples(tr), _triples(va)
    return users, items, data_tr, data_va, u_attr, i_attr, user_index, item_index
