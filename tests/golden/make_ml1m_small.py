"""Cuts a small slice out of the reference's example dataset (examples/dataset, MovieLens-1m
derived: DATA, not code) and records what the REAL reference loader -- utils/load_data.py, which
imports under Python 3 -- returns for it.  Run in the build container only:

    python tests/golden/make_ml1m_small.py

Outputs (committed): tests/golden/ml1m_small/*.csv and tests/golden/ml1m_small_load_raw_data.json.
The example dataset ships only obs_va / obs_te; the slice uses the first 3/5 of each user's
validation rows as obs_tr so that all three splits exist.
"""
import io
import json
import os
import sys

REF = os.environ.get("ARX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(REF, "examples", "dataset")
DST = os.path.join(HERE, "ml1m_small")
N_USERS = 60


def rows(name):
    with io.open(os.path.join(SRC, name), 'r', encoding='latin-1', newline='') as f:
        lines = f.read().split('\n')
    if lines[-1] == '':
        lines.pop()
    return lines[0], lines[1:]


def write(name, header, body):
    with io.open(os.path.join(DST, name), 'w', encoding='latin-1', newline='') as f:
        f.write('\n'.join([header] + body) + '\n')


def main():
    os.makedirs(DST, exist_ok=True)
    hu, users = rows('u.csv')
    users = users[:N_USERS]
    keep_u = set(u.split('\t')[0] for u in users)
    hva, va = rows('obs_va.csv')
    hte, te = rows('obs_te.csv')
    va = [r for r in va if r.split('\t')[0] in keep_u]
    te = [r for r in te if r.split('\t')[0] in keep_u]
    tr, va2 = [], []
    by_user = {}
    for r in va:
        by_user.setdefault(r.split('\t')[0], []).append(r)
    for u in sorted(by_user, key=int):
        rs = by_user[u]
        k = (3 * len(rs)) // 5
        tr.extend(rs[:k])
        va2.extend(rs[k:])
    keep_i = set(r.split('\t')[1] for r in tr + va2 + te)
    hi, items = rows('i.csv')
    # only ASCII titles, so that the reference loader (UTF-8 under py3) can read the slice too
    items = [r for r in items if r.split('\t')[0] in keep_i]
    bad = set(r.split('\t')[0] for r in items if any(ord(c) > 127 for c in r))
    items = [r for r in items if r.split('\t')[0] not in bad]
    drop = lambda rs: [r for r in rs if r.split('\t')[1] not in bad]
    tr, va2, te = drop(tr), drop(va2), drop(te)
    write('u.csv', hu, users)
    write('i.csv', hi, items)
    write('obs_tr.csv', hva, tr)
    write('obs_va.csv', hva, va2)
    write('obs_te.csv', hte, te)
    for n in ('u_attr.csv', 'i_attr.csv'):
        h, b = rows(n)
        write(n, h, b)

    sys.path.insert(0, os.path.join(REF, "utils"))
    import load_data as ref_ld
    out = {}
    for submit in (0, 1):
        (u, i, data_tr, data_va, u_attr, i_attr, user_index, item_index) = ref_ld.load_raw_data(DST, _submit=submit)
        out[str(submit)] = {
            "data_tr": [[int(x) for x in t] for t in data_tr],
            "data_va": [[int(x) for x in t] for t in data_va],
            "u_attr": [list(u_attr[0]), [int(x) for x in u_attr[1]]],
            "i_attr": [list(i_attr[0]), [int(x) for x in i_attr[1]]],
            "user_index": {str(k): int(v) for k, v in user_index.items()},
            "item_index": {str(k): int(v) for k, v in item_index.items()},
        }
    out["users"] = [[str(x) for x in r] for r in u.tolist()]
    out["items"] = [[str(x) for x in r] for r in i.tolist()]
    with open(os.path.join(HERE, "ml1m_small_load_raw_data.json"), "w") as f:
        json.dump(out, f, sort_keys=True)
    print("users", len(users), "items", len(items), "tr/va/te", len(tr), len(va2), len(te))


if __name__ == "__main__":
    main()
