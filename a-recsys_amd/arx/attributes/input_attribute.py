"""read_data -- py3 mirror of attributes/input_attribute.py:10-72: raw CSVs -> (data_tr, data_va,
u_attr, i_attr, item_ind2logit_ind, logit_ind2item_ind, user_index, item_index), cached in
`data_dir`.  Same arguments and return value; the cache is a flat binary file
(arx.utils.csr_cache, `data.arxcsr`) instead of a pickle of Python objects, so it can be mapped
straight into the attribute maps the kernels read.
"""
from __future__ import annotations

import os

import numpy as np

from . import attribute
from .comb_attribute import HET, MIX
from ..utils import csr_cache
from ..utils.load_data import load_raw_data

CACHE_NAME = 'data.arxcsr'


def _pack_attr(prefix, a, arrays, meta):
    meta[prefix] = {"num_features_cat": int(a.num_features_cat),
                    "num_features_mulhot": int(a.num_features_mulhot),
                    "v_sizes_cat": [int(v) for v in a._embedding_classes_list_cat],
                    "v_sizes_mulhot": [int(v) for v in a._embedding_classes_list_mulhot],
                    "mulhot_max_length": [int(v) for v in (a.mulhot_max_length or [])],
                    "n_full_cat": len(a.full_cat_tr), "n_full_mulhot": len(a.full_values_tr)}
    for i in range(a.num_features_cat):
        arrays['%s/cat/%d' % (prefix, i)] = np.asarray(a.features_cat[i], dtype=np.int32)
    for i in range(a.num_features_mulhot):
        arrays['%s/mulhot/%d/values' % (prefix, i)] = np.asarray(a.features_mulhot[i], dtype=np.int32)
        arrays['%s/mulhot/%d/starts' % (prefix, i)] = np.asarray(a.mulhot_starts[i], dtype=np.int64)
        arrays['%s/mulhot/%d/lengths' % (prefix, i)] = np.asarray(a.mulhot_lengths[i], dtype=np.int64)
    for i in range(len(a.full_cat_tr)):
        arrays['%s/full/cat/%d' % (prefix, i)] = np.asarray(a.full_cat_tr[i], dtype=np.int32)
    for i in range(len(a.full_values_tr)):
        arrays['%s/full/mulhot/%d/values' % (prefix, i)] = np.asarray(a.full_values_tr[i], dtype=np.int32)
        arrays['%s/full/mulhot/%d/segids' % (prefix, i)] = np.asarray(a.full_segids_tr[i], dtype=np.int32)
        arrays['%s/full/mulhot/%d/lengths' % (prefix, i)] = np.asarray(a.full_lengths_tr[i], dtype=np.float64)


def _unpack_attr(prefix, arrays, meta):
    m = meta[prefix]
    nc, nm = m["num_features_cat"], m["num_features_mulhot"]
    a = attribute.Attributes(
        nc, [arrays['%s/cat/%d' % (prefix, i)] for i in range(nc)],
        nm, [arrays['%s/mulhot/%d/values' % (prefix, i)] for i in range(nm)],
        list(m["mulhot_max_length"]),
        [arrays['%s/mulhot/%d/starts' % (prefix, i)] for i in range(nm)],
        [arrays['%s/mulhot/%d/lengths' % (prefix, i)] for i in range(nm)],
        list(m["v_sizes_cat"]), list(m["v_sizes_mulhot"]))
    if m["n_full_cat"] or m["n_full_mulhot"]:
        a.set_target_prediction(
            [arrays['%s/full/cat/%d' % (prefix, i)] for i in range(m["n_full_cat"])],
            [arrays['%s/full/mulhot/%d/values' % (prefix, i)] for i in range(m["n_full_mulhot"])],
            [arrays['%s/full/mulhot/%d/segids' % (prefix, i)] for i in range(m["n_full_mulhot"])],
            [arrays['%s/full/mulhot/%d/lengths' % (prefix, i)] for i in range(m["n_full_mulhot"])])
    return a


def _pack_index(name, index, arrays, meta):
    """{raw id -> row}: stored as the raw ids in row order (ints as an array, anything else as JSON)."""
    keys = [None] * len(index)
    for k, v in index.items():
        keys[v] = k
    if all(isinstance(k, (int, np.integer)) for k in keys):
        arrays[name] = np.asarray(keys, dtype=np.int64)
        meta[name] = 'array'
    else:
        meta[name] = [k if isinstance(k, str) else str(k) for k in keys]


def _unpack_index(name, arrays, meta):
    keys = arrays[name].tolist() if meta[name] == 'array' else meta[name]
    return {k: n for n, k in enumerate(keys)}


def save_cache(filename, data_tr, data_va, u_attr, i_attr, item_ind2logit_ind, logit_ind2item_ind,
               user_index, item_index):
    arrays, meta = {}, {}
    arrays['data_tr'] = np.asarray(data_tr, dtype=np.int64).reshape(-1, 3)
    arrays['data_va'] = np.asarray(data_va, dtype=np.int64).reshape(-1, 3)
    _pack_attr('user', u_attr, arrays, meta)
    _pack_attr('item', i_attr, arrays, meta)
    V = len(logit_ind2item_ind)
    arrays['logit_ind2item_ind'] = np.asarray([logit_ind2item_ind[j] for j in range(V)], dtype=np.int64)
    _pack_index('user_index', user_index, arrays, meta)
    _pack_index('item_index', item_index, arrays, meta)
    csr_cache.save(filename, arrays, meta)


def load_cache(filename):
    arrays, meta = csr_cache.load(filename)
    l2i = arrays['logit_ind2item_ind'].tolist()
    return ([tuple(r) for r in arrays['data_tr'].tolist()], [tuple(r) for r in arrays['data_va'].tolist()],
            _unpack_attr('user', arrays, meta), _unpack_attr('item', arrays, meta),
            {e: k for k, e in enumerate(l2i)}, {k: e for k, e in enumerate(l2i)},
            _unpack_index('user_index', arrays, meta), _unpack_index('item_index', arrays, meta))


def read_data(raw_data_dir='../raw_data/data/', data_dir='../cache/data/', combine_att='mix',
              logits_size_tr='10000', thresh=2, use_user_feature=True, use_item_feature=True,
              no_user_id=False, test=False, mylog=None):
    """input_attribute.py:10-72."""
    if not mylog:
        def mylog(val):
            print(val)
    data_filename = os.path.join(data_dir, CACHE_NAME)
    if os.path.isfile(data_filename):
        mylog("data file {} exists! loading cached data. \nCaution: change cached data dir (--data_dir) "
              "if new data (or new preprocessing) is used.".format(data_filename))
        out = load_cache(data_filename)
    else:
        if combine_att not in ('het', 'mix'):
            raise ValueError("combine_att must be 'het' or 'mix', got %r" % (combine_att,))
        if not os.path.exists(data_dir):
            os.makedirs(data_dir)
        (users, items, data_tr, data_va, user_features, item_features, user_index,
         item_index) = load_raw_data(data_dir=raw_data_dir, _submit=1 if test else 0)
        if not use_user_feature:                                   # id column only (:36-39)
            users = users[:, 0].reshape(len(users), 1)
            user_features = ([user_features[0][0]], [user_features[1][0]])
        if not use_item_feature:
            items = items[:, 0].reshape(len(items), 1)
            item_features = ([item_features[0][0]], [item_features[1][0]])
        if no_user_id:                                             # :43-44
            users = np.array(users, dtype=object)
            users[:, 0] = 0
        logits_size_tr = int(logits_size_tr)
        if combine_att == 'het':
            het = HET(data_dir=data_dir, logits_size_tr=logits_size_tr, threshold=thresh)
            u_attr, i_attr, item_ind2logit_ind, logit_ind2item_ind = het.get_attributes(
                users, items, data_tr, user_features, item_features)
        else:
            mix = MIX(data_dir=data_dir, logits_size_tr=logits_size_tr, threshold=thresh)
            users2, items2, user_features, item_features = mix.mix_attr(users, items, user_features,
                                                                        item_features)
            u_attr, i_attr, item_ind2logit_ind, logit_ind2item_ind = mix.get_attributes(
                users2, items2, data_tr, user_features, item_features)
        mylog("saving data format to data directory")
        out = (data_tr, data_va, u_attr, i_attr, item_ind2logit_ind, logit_ind2item_ind, user_index,
               item_index)
        save_cache(data_filename, *out)
    mylog('length of item_ind2logit_ind: {}'.format(len(out[4])))
    return out
