"""Preprocessing -> CSR attribute maps: arx.utils.load_data against the REAL
reference loader's output (golden), arx.utils.preprocess / arx.attributes.comb_attribute /
input_attribute against hand-computed answers and the loop restatement oracle/ref_preprocess.py
on a slice of the reference's ML-1m example dataset (tests/golden/ml1m_small)."""
import json
import os
import shutil

import numpy as np
import pytest

from oracle import ref_preprocess as rp

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, 'golden', 'ml1m_small')


def _golden():
    with open(os.path.join(HERE, 'golden', 'ml1m_small_load_raw_data.json')) as f:
        return json.load(f)


@pytest.mark.parametrize("submit", [0, 1])
def test_load_raw_data_matches_reference_loader(submit):
    from arx.utils.load_data import load_raw_data
    g = _golden()
    users, items, data_tr, data_va, u_attr, i_attr, user_index, item_index = load_raw_data(DATA, submit)
    e = g[str(submit)]
    assert [[str(x) for x in r] for r in users.tolist()] == g["users"]
    assert [[str(x) for x in r] for r in items.tolist()] == g["items"]
    assert [list(map(int, t)) for t in data_tr] == e["data_tr"]
    assert [list(map(int, t)) for t in data_va] == e["data_va"]
    assert [list(u_attr[0]), list(u_attr[1])] == e["u_attr"]
    assert [list(i_attr[0]), list(i_attr[1])] == e["i_attr"]
    assert {str(k): v for k, v in user_index.items()} == e["user_index"]
    assert {str(k): v for k, v in item_index.items()} == e["item_index"]
    assert isinstance(data_tr, list) and isinstance(data_tr[0], tuple)


def test_loader_reads_latin1_and_two_column_logs(tmp_path):
    from arx.utils.load_data import load_raw_data
    d = str(tmp_path)
    with open(os.path.join(d, 'u.csv'), 'wb') as f:
        f.write(b'id\tcity\n7\tParis\n9\tOrl\xe9ans\n')
    with open(os.path.join(d, 'i.csv'), 'wb') as f:
        f.write(b'id\ttitle\n100\tAm\xe9lie,2001\n200\tHeat,1995\n')
    with open(os.path.join(d, 'obs_tr.csv'), 'wb') as f:
        f.write(b'user\titem\n9\t100\n7\t200\n9\t200\n')
    users, items, data_tr, data_va, u_attr, i_attr, ui, ii = load_raw_data(d)
    assert users[1, 1] == 'Orl\xe9ans' and items[0, 1] == 'Am\xe9lie,2001'
    assert data_tr == [(1, 0, 0), (0, 1, 0), (1, 1, 0)]          # zero time column added
    assert data_va == [] and u_attr[1] == [0, 0] and ui == {7: 0, 9: 1}
    with pytest.raises(FileNotFoundError):
        load_raw_data(os.path.join(d, 'nope'))


def test_dictionary_and_tokenizer_known_answer(tmp_path):
    """Hand-computed: counts run over interactions, threshold drops rare tokens, order = count
    descending (ties: first seen), unknown tokens vanish from a bag unless nothing is left."""
    from arx.utils import preprocess as pp
    d = str(tmp_path)
    items = np.array([[10, 'a,b', 'x'],          # row 0
                      [11, 'b,c', 'y'],          # row 1
                      [12, 'zz', 'x'],           # row 2: its only token is rare -> [_UNK]
                      [13, 'c,a,b', 'w']], dtype=object)
    types = [0, 1, 2]                            # third column ignored (type > 1)
    inds = [1, 0, 1, 3, 1, 2]                    # item rows of the training interactions
    # column 0 counts: 11 x3, 10 x1, 13 x1, 12 x1 ; threshold 1, limit logits_size_tr + 2 = 4
    # column 1 counts: b: 3 (row 1) + 1 (row 0) + 1 (row 3) = 5, c: 3 + 1 = 4, a: 1 + 1 = 2, zz: 1
    v = pp.create_dictionary(d, inds, items, types, ['id', 'tags', 'junk'], max_vocabulary_size=50,
                             logits_size_tr=2, threshold=2, prefix='item')
    assert v == [['_UNK', '_START', '11'], ['_UNK', '_START', 'b', 'c', 'a']]
    assert sorted(os.listdir(d)) == ['item_minimum_occurance_50', 'item_vocab0_4', 'item_vocab1_50']
    assert open(os.path.join(d, 'item_vocab1_50')).read() == '_UNK\n_START\nb\nc\na\n'
    assert open(os.path.join(d, 'item_minimum_occurance_50')).read() == '3\n2'
    (nc, cat, nm, mul, mx, starts, lens, vc, vm) = pp.tokenize_attribute_map(d, items, types, 50, 2, 'item')
    assert (nc, nm, vc, vm, mx) == (1, 1, [3], [5], [3])
    assert cat[0].tolist() == [0, 2, 0, 0, 1]                     # only '11' is in the vocabulary; last = _START
    assert mul[0].tolist() == [4, 2, 2, 3, 0, 3, 4, 2, 1]         # a,b | b,c | [_UNK] | c,a,b | _START
    assert starts[0].tolist() == [0, 2, 4, 5, 8, 9] and lens[0].tolist() == [2, 2, 1, 3, 1]
    assert items[0, 0] == 10                                       # argument not overwritten
    l2i = {0: 3, 1: 1}
    assert [a.tolist() for a in pp.filter_cat(1, cat, l2i)] == [[0, 2]]
    fv, fv_tr, fs, fl, fs_tr, fl_tr = pp.filter_mulhot(d, items, types, 50, l2i)
    assert fv[0].tolist() == [4, 2, 2, 3, 0, 3, 4, 2] and fs[0].tolist() == [0, 0, 1, 1, 2, 3, 3, 3]
    assert fv_tr[0].tolist() == [3, 4, 2, 2, 3] and fs_tr[0].tolist() == [0, 0, 0, 1, 1]
    assert fl_tr[0].tolist() == [[3.0], [2.0]] and fl[0].shape == (4, 1)


def test_mix_bags_known_answer():
    from arx.attributes.comb_attribute import MIX
    users = np.array([[5, 'F', 'x,y'], [6, 'M', 'y']], dtype=object)
    items = np.array([[1], [2]], dtype=object)
    u2, i2, uf, itf = MIX('unused').mix_attr(users, items, (['id', 'g', 'tags'], [0, 0, 1]), (['id'], [0]))
    assert u2[:, 0].tolist() == ['uid5,gF,tagsx,tagsy', 'uid6,gM,tagsy']
    assert i2[:, 0].tolist() == ['id1', 'id2']
    assert uf == (['mix'], [1]) and itf == (['mix'], [0])          # a lone categorical stays categorical


def _raw():
    from arx.utils.load_data import load_raw_data
    return load_raw_data(DATA, 0)


def _as_lists(a):
    return [np.asarray(x).tolist() for x in a]


def test_het_matches_loop_restatement(tmp_path):
    from arx.attributes.comb_attribute import HET
    users, items, data_tr, data_va, uf, itf, _, _ = _raw()
    V = 300
    het = HET(str(tmp_path), logits_size_tr=V, threshold=1)
    u_attr, i_attr, i2l, l2i = het.get_attributes(users, items, data_tr, uf, itf)
    u_inds, i_inds = [p[0] for p in data_tr], [p[1] for p in data_tr]
    for prefix, feats, inds, (names, types), attr in (('user', users, u_inds, uf, u_attr),
                                                     ('item', items, i_inds, itf, i_attr)):
        vocabs = rp.vocab_het(inds, feats.tolist(), types, V, 50000, 1, prefix)
        cat, mul = rp.tokenize(feats.tolist(), types, vocabs)
        assert attr.num_features_cat == len(cat) and attr.num_features_mulhot == len(mul)
        assert _as_lists(attr.features_cat) == cat
        assert attr._embedding_classes_list_cat == [len(vocabs[i]) for i, t in enumerate(types) if t == 0]
        assert attr._embedding_classes_list_mulhot == [len(vocabs[i]) for i, t in enumerate(types) if t == 1]
        for k, (vals, starts, lens, mx) in enumerate(mul):
            assert attr.features_mulhot[k].tolist() == vals
            assert attr.mulhot_starts[k].tolist() == starts
            assert attr.mulhot_lengths[k].tolist() == lens
            assert attr.mulhot_max_length[k] == mx
    order = rp.index_mapping_het(cat[0], len(items))
    assert len(order) == V and [l2i[j] for j in range(V)] == order
    assert all(i2l[e] == k for k, e in enumerate(order))
    assert _as_lists(i_attr.full_cat_tr) == [[c[i] for i in order] for c in cat]
    for k, m in enumerate(mul):
        v_tr, seg_tr, len_tr = rp.full_mulhot(m, order)
        assert i_attr.full_values_tr[k].tolist() == v_tr
        assert i_attr.full_segids_tr[k].tolist() == seg_tr
        assert i_attr.full_lengths_tr[k].tolist() == len_tr
    # the vocabulary cut really bites: more training items than logits
    assert len(set(i_inds)) > V
    with pytest.raises(AssertionError):
        HET(str(tmp_path / 'x'), logits_size_tr=10 ** 6).index_mapping(np.asarray(cat[0]), i_inds, len(items))


def test_mix_matches_loop_restatement(tmp_path):
    from arx.attributes.comb_attribute import MIX
    users, items, data_tr, data_va, uf, itf, _, _ = _raw()
    V = 250
    mix = MIX(str(tmp_path), logits_size_tr=V, threshold=2)
    u2, i2, uf2, itf2 = mix.mix_attr(users, items, (list(uf[0]), uf[1]), itf)
    names_u = ['uid'] + list(uf[0][1:])
    assert u2[:, 0].tolist() == rp.mix_bags(users.tolist(), names_u, uf[1])
    assert i2[:, 0].tolist() == rp.mix_bags(items.tolist(), itf[0], itf[1])
    u_attr, i_attr, i2l, l2i = mix.get_attributes(u2, i2, data_tr, uf2, itf2)
    u_inds, i_inds = [p[0] for p in data_tr], [p[1] for p in data_tr]
    for feats, inds, attr in ((u2, u_inds, u_attr), (i2, i_inds, i_attr)):
        vocab = rp.vocab_mix(inds, feats.tolist(), 500000, 2)
        cat, mul = rp.tokenize(feats.tolist(), [1], {0: vocab})
        assert attr.num_features_cat == 0 and attr.num_features_mulhot == 1
        assert attr._embedding_classes_list_mulhot == [len(vocab)]
        vals, starts, lens, mx = mul[0]
        assert attr.features_mulhot[0].tolist() == vals and attr.mulhot_starts[0].tolist() == starts
        assert attr.mulhot_lengths[0].tolist() == lens and attr.mulhot_max_length == [mx]
    order = rp.index_mapping_mix(i_inds, V)
    assert [l2i[j] for j in range(V)] == order and len(i2l) == V
    v_tr, seg_tr, len_tr = rp.full_mulhot(mul[0], order)
    assert i_attr.full_values_tr[0].tolist() == v_tr and i_attr.full_segids_tr[0].tolist() == seg_tr
    assert i_attr.full_lengths_tr[0].tolist() == len_tr
    # user ids seen fewer than `threshold` times fall out of the vocabulary -> bag without a uid token
    uid_rows = {w: k for k, w in enumerate(rp.vocab_mix(u_inds, u2.tolist(), 500000, 2))}
    assert any(('uid%s' % users[n, 0]) not in uid_rows for n in range(len(users)))


@pytest.mark.parametrize("comb", ['het', 'mix'])
def test_read_data_cache_round_trip(tmp_path, comb):
    from arx.attributes.input_attribute import read_data, CACHE_NAME
    from arx.utils import csr_cache
    cache = str(tmp_path / 'cache')
    log = []
    a = read_data(DATA, cache, comb, 200, 1, mylog=log.append)
    assert os.path.isfile(os.path.join(cache, CACHE_NAME))
    b = read_data(DATA, cache, comb, 200, 1, mylog=log.append)
    assert any('loading cached data' in m for m in log)
    assert a[0] == b[0] and a[1] == b[1] and a[4] == b[4] and a[5] == b[5] and a[6] == b[6] and a[7] == b[7]
    for x, y in ((a[2], b[2]), (a[3], b[3])):
        assert x.num_features_cat == y.num_features_cat and x.num_features_mulhot == y.num_features_mulhot
        for name in ('features_cat', 'features_mulhot', 'mulhot_starts', 'mulhot_lengths', 'full_cat_tr',
                     'full_values_tr', 'full_segids_tr', 'full_lengths_tr'):
            assert _as_lists(getattr(x, name)) == _as_lists(getattr(y, name)), name
        assert list(x.mulhot_max_length) == list(y.mulhot_max_length)
        assert x._embedding_classes_list_cat == y._embedding_classes_list_cat
    arrays, meta = csr_cache.load(os.path.join(cache, CACHE_NAME), mmap=True)
    assert all(int(getattr(v, 'offset', 0)) % 64 == 0 for v in arrays.values())
    assert arrays['data_tr'].shape[1] == 3 and meta['item']['n_full_mulhot'] == a[3].num_features_mulhot
    # switches: id-only tables, and no_user_id collapses every user onto one id row
    c = read_data(DATA, str(tmp_path / 'c2'), comb, 200, 1, use_user_feature=False, use_item_feature=False,
                  no_user_id=True, mylog=log.append)
    nfeat = c[2].num_features_cat + c[2].num_features_mulhot
    assert nfeat == 1
    col = c[2].features_cat[0][:-1] if c[2].num_features_cat else c[2].features_mulhot[0][:-1]
    assert len(set(np.asarray(col).tolist())) == 1
