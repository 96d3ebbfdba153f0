"""Attribute vocabularies and the CSR attribute maps the kernels consume -- py3 mirror of the
reference's utils/preprocess.py (same function names, arguments and outputs; SURVEY 8f #2,
Appendix B), written array-at-a-time instead of as Python loops over entities.

Vocabulary rows: 0 = _UNK, 1 = _START (preprocess.py:7-15).  Layout of the outputs:
  categorical f : int32[N+1]   vocabulary row of entity n, last entry = _START      (:191-199)
  multi-hot f   : values int32[sum len + 1] (last = _START), starts int64[N+2], lengths int64[N+1];
                  tokens outside the vocabulary are dropped, an entity left with none keeps a
                  single _UNK                                                          (:200-231)

Where the reference's result depends on Python-2 dict iteration order (ties between equally
frequent tokens at preprocess.py:93, the order of the `uid` tokens at :145) this module fixes the
order to first appearance; any such order yields a valid vocabulary (the models are invariant to
a permutation of table rows).  Cells are compared by their `str()` (the reference looks tokens
up by `str(value)` too, :196,210).

No TensorFlow gfile, no pickle: vocabularies are plain text files, one token per line, named as
the reference names them (`<prefix>_vocab<i>_<max_size>`), so existing caches stay readable.
"""
from __future__ import annotations

from os import listdir
from os.path import isfile, join

import numpy as np
import pandas as pd

_UNK = "_UNK"
_START = "_START"
UNK_ID = 0
START_ID = 1
_START_VOCAB = [_UNK, _START]
ENCODING = 'latin-1'


# ------------------------------------------------------------------ small helpers
def _cell_str(v):
    if isinstance(v, list):
        return ','.join(str(t) for t in v)
    return v if isinstance(v, str) else str(v)


def _col_str(col):
    """Object column -> pandas string Series (`str(cell)`; list cells are re-joined)."""
    s = pd.Series(np.asarray(col, dtype=object))
    if s.map(type).eq(str).all():
        return s
    return s.map(_cell_str)


def _explode(col_str):
    """Comma-split a string column -> (token array, row number of every token)."""
    parts = col_str.str.split(',')
    lens = parts.str.len().to_numpy(dtype=np.int64)
    rows = np.repeat(np.arange(len(col_str), dtype=np.int64), lens)
    toks = np.fromiter((t for p in parts for t in p), dtype=object, count=int(lens.sum()))
    return toks, rows


def _count_first_seen(tokens, weights=None):
    """-> (distinct tokens in order of first appearance, their (weighted) counts)."""
    codes, uniq = pd.factorize(tokens, sort=False)
    cnt = np.bincount(codes, weights=weights, minlength=len(uniq))
    return np.asarray(uniq, dtype=object), cnt.astype(np.int64)


def _by_count_desc(tokens, counts):
    order = np.argsort(-counts, kind='stable')          # ties keep first-appearance order
    return tokens[order], counts[order]


def _entity_multiplicity(inds, n_rows):
    """Entities in order of first appearance in `inds`, and how often each appears."""
    codes, uniq = pd.factorize(np.asarray(inds, dtype=np.int64), sort=False)
    return np.asarray(uniq, dtype=np.int64), np.bincount(codes, minlength=len(uniq)).astype(np.int64)


def _write_vocab(path, tokens):
    with open(path, 'w', encoding=ENCODING, newline='\n') as f:
        for w in tokens:
            f.write(str(w) + '\n')


def initialize_vocabulary(vocabulary_path):
    """preprocess.py:23-50 -> ({token: row}, [token of row r]); one token per line."""
    if not isfile(vocabulary_path):
        raise ValueError("Vocabulary file %s not found." % vocabulary_path)
    with open(vocabulary_path, 'r', encoding=ENCODING, newline='\n') as f:
        rev_vocab = [line.strip() for line in f.read().split('\n')]
    if rev_vocab and rev_vocab[-1] == '':
        rev_vocab.pop()
    vocab = {x: y for y, x in enumerate(rev_vocab)}      # a repeated token keeps its LAST row
    return vocab, rev_vocab


def _find_vocab(data_dir, prefix, i):
    head = "%s_vocab%d_" % (prefix, i)
    paths = [f for f in listdir(data_dir) if f.startswith(head)]
    if len(paths) != 1:
        raise ValueError("expected exactly one %s* file in %s, found %d -- delete the stale ones"
                         % (head, data_dir, len(paths)))
    return join(data_dir, paths[0])


# ------------------------------------------------------------------ vocabularies
def create_dictionary(data_dir, inds, features, feature_types, feature_names,
                      max_vocabulary_size=50000, logits_size_tr=50000, threshold=2, prefix='user'):
    """preprocess.py:52-119 (HET): one vocabulary per attribute column of type 0/1.  Counts run over
    the TRAINING interactions -- entity row `inds[k]` is counted once per interaction -- tokens
    seen fewer than `threshold` times are dropped, the rest is ordered by count (descending) after
    _UNK, _START and cut to the size limit (items' column 0: logits_size_tr + 2, else
    max_vocabulary_size).  Writes `<prefix>_vocab<i>_<limit>` and `<prefix>_minimum_occurance_<limit>`;
    returns the vocabularies (list of token lists; the reference returns None)."""
    num_f = len(feature_names)
    if len(feature_types) != num_f:
        raise AssertionError('length of feature_types should be the same length of feature_names '
                             '{} vs {}'.format(len(feature_types), num_f))
    features = np.asarray(features, dtype=object)
    ents, mult = _entity_multiplicity(inds, len(features))
    vocabs, minimum_occurance, max_size = [], [], max_vocabulary_size
    for i in range(num_f):
        if feature_types[i] > 1:
            continue
        col = _col_str(features[ents, i])
        if feature_types[i] == 0:
            toks, cnt = _count_first_seen(col.to_numpy(dtype=object), mult.astype(np.float64))
        else:
            t, rows = _explode(col)
            toks, cnt = _count_first_seen(t, mult[rows].astype(np.float64))
        toks, cnt = _by_count_desc(toks, cnt)
        max_size = logits_size_tr + len(_START_VOCAB) if (prefix == 'item' and i == 0) else max_vocabulary_size
        keep = cnt >= threshold
        vocab_list = (_START_VOCAB + toks[keep].tolist())[:max_size]
        if int(keep.sum()) + len(_START_VOCAB) > max_size:
            print("vocabulary {}_{} longer than max_vocabulary_size {}. Truncate the tail".format(
                prefix, int(keep.sum()) + len(_START_VOCAB), max_size))
        _write_vocab(join(data_dir, "%s_vocab%d_%d" % (prefix, i, max_size)), vocab_list)
        n_kept = len(vocab_list) - len(_START_VOCAB)
        minimum_occurance.append(int(cnt[keep][n_kept - 1]) if n_kept > 0 else 0)
        vocabs.append(vocab_list)
    with open(join(data_dir, "%s_minimum_occurance_%d" % (prefix, max_size)), 'w') as f:
        f.write('\n'.join(str(v) for v in minimum_occurance))
    return vocabs


def create_dictionary_mix(data_dir, inds, features, feature_types, feature_names,
                          max_vocabulary_size=50000, logits_size_tr=50000, threshold=2, prefix='user'):
    """preprocess.py:121-166 (MIX): ONE vocabulary over the entity's merged bag of name-prefixed
    tokens (column 0 of `features`, see MIX.mix_attr).  `uid...` tokens come first (un-sorted),
    then all other tokens by descending count; same threshold / size cut."""
    if len(feature_types) != len(feature_names):
        raise AssertionError('length of feature_types should be the same length of feature_names')
    features = np.asarray(features, dtype=object)
    ents, mult = _entity_multiplicity(inds, len(features))
    t, rows = _explode(_col_str(features[ents, 0]))
    toks, cnt = _count_first_seen(t, mult[rows].astype(np.float64))
    is_uid = np.fromiter((str(x).startswith('uid') for x in toks), dtype=bool, count=len(toks))
    o_t, o_c = _by_count_desc(toks[~is_uid], cnt[~is_uid])
    all_t = np.concatenate([toks[is_uid], o_t])
    all_c = np.concatenate([cnt[is_uid], o_c])
    keep = all_c >= threshold
    max_size = max_vocabulary_size
    vocab_list = (_START_VOCAB + all_t[keep].tolist())[:max_size]
    if int(keep.sum()) + len(_START_VOCAB) > max_size:
        print("vocabulary {}_{} longer than max_vocabulary_size {}. Truncate the tail".format(
            prefix, int(keep.sum()) + len(_START_VOCAB), max_size))
    _write_vocab(join(data_dir, "%s_vocab%d_%d" % (prefix, 0, max_size)), vocab_list)
    n_kept = len(vocab_list) - len(_START_VOCAB)
    with open(join(data_dir, "%s_minimum_occurance_%d" % (prefix, max_size)), 'w') as f:
        f.write(str(int(all_c[keep][n_kept - 1]) if n_kept > 0 else 0))
    return [vocab_list]


# ------------------------------------------------------------------ tokenisation
def _lookup(tokens, vocab):
    ids = pd.Series(tokens, dtype=object).map(vocab)
    return ids.fillna(UNK_ID).to_numpy(dtype=np.int64)


def _tokenize_bags(col, vocab):
    """Multi-hot column -> (values int64[sum len], lengths int64[N]): known tokens in order,
    [_UNK] for a bag with none (preprocess.py:205-214)."""
    n = len(col)
    t, rows = _explode(_col_str(col))
    ids = _lookup(t, vocab)
    known = ids != UNK_ID
    lens = np.bincount(rows[known], minlength=n).astype(np.int64)
    empty = np.flatnonzero(lens == 0)
    vals = np.concatenate([ids[known], np.full(len(empty), UNK_ID, dtype=np.int64)])
    r = np.concatenate([rows[known], empty])
    vals = vals[np.argsort(r, kind='stable')]
    lens[empty] = 1
    return vals, lens


def tokenize_attribute_map(data_dir, features, feature_types, max_vocabulary_size,
                           logits_size_tr=50000, prefix='user'):
    """preprocess.py:168-238: entity attribute table -> the arrays of an `Attributes` object,
    with the vocabularies written by create_dictionary*.  Returns (num_features_cat, features_cat,
    num_features_mulhot, features_mulhot, mulhot_max_leng, mulhot_starts, mulhot_lengs,
    v_sizes_cat, v_sizes_mulhot).  `features` is left untouched (the reference overwrites the
    categorical columns of its argument with the token rows, :196)."""
    features = np.asarray(features, dtype=object)
    features_cat, features_mulhot = [], []
    v_sizes_cat, v_sizes_mulhot = [], []
    mulhot_max_leng, mulhot_starts, mulhot_lengs = [], [], []
    for i, ut in enumerate(feature_types):
        if ut > 1:
            continue
        vocab, _ = initialize_vocabulary(_find_vocab(data_dir, prefix, i))
        col = features[:, i]
        if ut == 0:
            v_sizes_cat.append(len(vocab))
            ids = _lookup(_col_str(col).to_numpy(dtype=object), vocab)
            features_cat.append(np.append(ids, START_ID).astype(np.int32))
        else:
            v_sizes_mulhot.append(len(vocab))
            vals, lens = _tokenize_bags(col, vocab)
            mulhot_max_leng.append(int(lens.max()) if len(lens) else 0)
            lens1 = np.append(lens, 1)                                  # the _START entity (:223-226)
            mulhot_starts.append(np.concatenate([[0], np.cumsum(lens1)]).astype(np.int64))
            mulhot_lengs.append(lens1)
            features_mulhot.append(np.append(vals, START_ID).astype(np.int32))
    num_features_cat = sum(v == 0 for v in feature_types)
    num_features_mulhot = sum(v == 1 for v in feature_types)
    return (num_features_cat, features_cat, num_features_mulhot, features_mulhot, mulhot_max_leng,
            mulhot_starts, mulhot_lengs, v_sizes_cat, v_sizes_mulhot)


def _logit_order(logit_ind2item_ind):
    L = len(logit_ind2item_ind)
    if isinstance(logit_ind2item_ind, dict):
        return np.fromiter((logit_ind2item_ind[j] for j in range(L)), dtype=np.int64, count=L)
    return np.asarray(logit_ind2item_ind, dtype=np.int64)


def filter_cat(num_features_cat, features_cat, logit_ind2item_ind):
    """preprocess.py:240-254: categorical maps re-ordered by logit index (row j = item
    logit_ind2item_ind[j]) for the full-vocabulary scorer."""
    order = _logit_order(logit_ind2item_ind)
    return [np.asarray(features_cat[i])[order] for i in range(num_features_cat)]


def filter_mulhot(data_dir, items, feature_types, max_vocabulary_size, logit_ind2item_ind, prefix='item'):
    """preprocess.py:257-326 -> (full_values, full_values_tr, full_segids, full_lengths,
    full_segids_tr, full_lengths_tr), one entry per multi-hot column: the bags of ALL items in
    item order, and of the V logit items in logit order (values int32, segment ids int32
    ascending, lengths float [n, 1])."""
    items = np.asarray(items, dtype=object)
    order = _logit_order(logit_ind2item_ind)
    N, L = len(items), len(order)
    out = ([], [], [], [], [], [])
    for i, ut in enumerate(feature_types):
        if ut != 1:
            continue
        vocab, _ = initialize_vocabulary(_find_vocab(data_dir, prefix, i))
        vals, lens = _tokenize_bags(items[:, i], vocab)
        starts = np.concatenate([[0], np.cumsum(lens)])
        lens_tr = lens[order]
        # gather the bags of the logit items: positions starts[o] .. starts[o] + len
        seg_tr = np.repeat(np.arange(L, dtype=np.int64), lens_tr)
        first = np.repeat(starts[order], lens_tr)
        within = np.arange(int(lens_tr.sum()), dtype=np.int64) - np.repeat(
            np.concatenate([[0], np.cumsum(lens_tr)])[:-1], lens_tr)
        out[0].append(vals.astype(np.int32))
        out[1].append(vals[first + within].astype(np.int32))
        out[2].append(np.repeat(np.arange(N, dtype=np.int64), lens).astype(np.int32))
        out[3].append(lens.astype(np.float64).reshape(N, 1))
        out[4].append(seg_tr.astype(np.int32))
        out[5].append(lens_tr.astype(np.float64).reshape(L, 1))
    return out
