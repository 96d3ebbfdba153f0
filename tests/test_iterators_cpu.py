"""Batch iterators / bucketing of the runners (callers of the hot path): bit-exact against the
REAL reference classes where they import under Python 3 (word2vec/data_iterator.py; the
deterministic half of lstm/data_iterator.py), hand-computed for lstm/best_buckets.py (Python-2
only)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_w2v_iterator_bit_exact_with_reference_stream():
    from arx.word2vec.data_iterator import DataIterator, batch_major
    cases = json.load(open(os.path.join(HERE, 'golden', 'w2v_iterator.json')))
    assert {c['gen'] for c in cases} == {'get_next', 'get_next_sg', 'get_next_cbow'}
    for c in cases:
        seq = [tuple(x) for x in c['seq']]
        it = DataIterator(seq, c['end_ind'], c['batch'], c['n_skips'], c['window'], c['sequence'])
        np.random.seed(c['seed'])
        g = getattr(it, c['gen'])()
        for k, b in enumerate(c['batches']):
            u, i, o = next(g)
            assert np.asarray(u).tolist() == b['users'], (c['gen'], k)
            assert np.asarray(i).tolist() == b['inputs'], (c['gen'], k)
            assert np.asarray(o).tolist() == b['outputs'], (c['gen'], k)
        assert it.index == c['index_after']
    assert batch_major([[1, 2, 3], [4, 5, 6]], 2, 3) == [[1, 4], [2, 5], [3, 6]]


class _Recorder(object):
    def __init__(self, sizes, batch):
        self.sizes, self.batch, self.calls = sizes, batch, []

    def get_batch(self, data_set, bucket_id, start_id=None):
        self.calls.append(['train', bucket_id, start_id])
        if start_id is None:
            return ('u', bucket_id, None), 'i', 'o', 'w', False
        return ('u', bucket_id, start_id), 'i', 'o', 'w', start_id + self.batch >= self.sizes[bucket_id]

    def get_batch_recommend(self, data_set, bucket_id, start_id=None):
        self.calls.append(['rec', bucket_id, start_id])
        return ('r', bucket_id, start_id), 'i', 'o', 'w', start_id + self.batch >= self.sizes[bucket_id]


def test_lstm_iterator_matches_reference_sweep():
    from arx.lstm.data_iterator import DataIterator
    for c in json.load(open(os.path.join(HERE, 'golden', 'lstm_iterator.json'))):
        m = _Recorder(c['sizes'], c['batch'])
        it = DataIterator(m, None, len(c['sizes']), c['batch'], [1.0])
        ys = []
        for k, y in enumerate(it.next_sequence(stop=c['stop'], recommend=c['recommend'])):
            ys.append([list(y[0]), y[4]])
            if k + 1 >= c['take']:
                break
        assert m.calls == c['calls'] and ys == c['yields']


def test_lstm_iterator_next_random_follows_bucket_shares():
    """data_iterator.py:14-22: bucket = first whose cumulative share exceeds a uniform draw."""
    from arx.lstm.data_iterator import DataIterator
    m = _Recorder([1, 1, 1], 2)
    it = DataIterator(m, None, 3, 2, [0.2, 0.5, 1.0])
    np.random.seed(3)
    draws = np.random.random_sample(2000)
    np.random.seed(3)
    g = it.next_random()
    got = [next(g)[4] for _ in range(2000)]
    exp = [int(np.searchsorted([0.2, 0.5, 1.0], x, side='right')) for x in draws]
    assert got == exp
    assert abs(got.count(0) / 2000.0 - 0.2) < 0.03 and abs(got.count(2) / 2000.0 - 0.5) < 0.04


def test_best_buckets_known_answers():
    from arx.lstm.best_buckets import calculate_buckets
    seqs = [(0, [0] * l) for l in [1, 1, 1, 2, 2, 5, 5, 5, 5, 9]]
    # cumulative counts (1,3) (2,5) (5,9) (9,10): first boundary = the longest length; its best
    # split saves (9-5) * (9-3) = 24 paddings at length 5 (length 2 would save 7*2 = 14)
    assert calculate_buckets(seqs, 10, 2) == [9, 5]
    assert calculate_buckets(seqs, 10, 3) == [9, 5, 2]      # next: inside (1..5): (5-2)*(5-3) = 6
    assert calculate_buckets(seqs, 10, 8) == [1, 2, 5, 9]   # fewer distinct lengths than buckets
    assert calculate_buckets(seqs, 6, 2) == [5, 2]          # lengths above max_length are cut off


def test_evaluation_harness_matches_reference(tmp_path):
    """utils/evaluate.py::Evaluation + utils/submit.py on the ML-1m slice: the four files it
    writes, the parsed truth / history and the flattened scores of a fixed recommendation dict --
    all against the real reference classes (tests/golden/evaluation.json)."""
    import shutil
    from arx.utils.evaluate import Evaluation
    from arx.utils.submit import combine_sub, load_submit
    g = json.load(open(os.path.join(HERE, 'golden', 'evaluation.json')))
    src = os.path.join(HERE, 'golden', 'ml1m_small')
    for test in (False, True):
        e = g[str(test)]
        d = str(tmp_path / ('t%d' % test))
        shutil.copytree(src, d)
        ev = Evaluation(d, test=test)
        for name, content in e['files'].items():
            assert open(os.path.join(d, name), 'rb').read().decode('latin-1') == content, name
        assert [int(u) for u in ev.get_uids()] == e['uids'] and [int(u) for u in ev.get_uinds()] == e['uinds']
        assert {str(k): v for k, v in ev.T.items()} == e['T']
        assert {str(k): len(v) for k, v in ev.hist.items()} == e['hist_len']
        rec = {int(k): list(v) for k, v in e['rec'].items()}
        ev.eval_on(rec)
        s_self, s_ex = ev.get_scores()
        np.testing.assert_allclose(s_self, e['s_self'], rtol=1e-12, atol=0)
        np.testing.assert_allclose(s_ex, e['s_ex'], rtol=1e-12, atol=0)
        assert ev.get_user_n() == len(e['uids'])
        # second construction reuses the files
        assert load_submit('res_T.csv', submit_dir=d) == Evaluation(d, test=False).T
    users = np.array([[1], [2], [3]], dtype=object)
    assert combine_sub({1: ['a', 'b']}, {1: ['b', 'c'], 2: ['x', 'x']}, 0, users) == {1: ['a', 'b', 'c'], 2: ['x']}
    assert combine_sub({1: ['a', 'b']}, {1: ['b', 'c'], 2: ['x']}, 1, users) == {1: ['c'], 2: ['x']}


class _Batcher(object):
    """SeqBatching needs only these attributes (the GPU model supplies them in production)."""

    def __init__(self, buckets, batch_size, start_id, pad_id, user_pad=0):
        from arx.lstm.batching import SeqBatching
        self.__class__ = type('B', (SeqBatching,), {})
        self.buckets, self.batch_size = buckets, batch_size
        self.START_ID, self.PAD_ID, self.USER_PAD_ID = start_id, pad_id, user_pad


def test_seq_get_batch_and_get_batch_recommend_known_answers():
    """lstm/seqModel.py:356-452 (TensorFlow module, not importable): hand-computed batches.
    Bucket of length 4, batch of 3, two real examples then padding; START = PAD = 99."""
    b = _Batcher([4], 3, 99, 99)
    data = [[(7, [11, 12, 13]), (8, [21])]]
    users, inp, tgt, w, fin = b.get_batch(data, 0, start_id=0)
    assert users == [7, 8, 0] and fin is True
    # per example: [START, s0, s1, PAD] / [START, PAD, PAD, PAD] / empty slot -> START + PAD; time-major
    assert inp == [[99, 99, 99], [11, 99, 99], [12, 99, 99], [99, 99, 99]]
    assert tgt == [[11, 21, 99], [12, 99, 99], [13, 99, 99], [99, 99, 99]]
    assert w == [[1.0, 1.0, 0.0], [1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
    users, inp, pos, valid, fin = b.get_batch_recommend(data, 0, start_id=0)
    assert users == [7, 8, 0] and pos == [2, 0, 3] and valid == [1, 1, 0] and fin is True
    assert inp == [[11, 21, 99], [12, 99, 99], [13, 99, 99], [99, 99, 99]]
    # not finished while examples remain; random draws come from the bucket, one per slot
    b2 = _Batcher([2, 4], 1, 5, 5)
    data2 = [[(1, [3])], [(2, [4, 6, 8]), (3, [9, 9, 9, 9])]]
    assert b2.get_batch(data2, 1, start_id=0)[4] is False and b2.get_batch(data2, 1, start_id=1)[4] is True
    import random
    random.seed(3)
    picks = [random.choice(data2[1]) for _ in range(4)]
    random.seed(3)
    for p in picks:
        users, inp, pos, valid, fin = b2.get_batch_recommend(data2, 1)
        assert users == [p[0]] and pos == [len(p[1]) - 1] and valid == [1] and fin is False


def test_lstm_iterator_recommend_sweep_through_batching():
    """DataIterator.next_sequence(recommend=True) over the real batching (the
    iterator called a method the model did not have)."""
    from arx.lstm.data_iterator import DataIterator
    b = _Batcher([3], 2, 1, 0)
    data = [[(5, [2, 3]), (6, [4]), (7, [8, 9, 10])]]
    got = list(DataIterator(b, data, 1, 2, [1.0]).next_sequence(stop=True, recommend=True))
    assert [g[0] for g in got] == [[5, 6], [7, 0]]
    assert [g[2] for g in got] == [[1, 0], [2, 2]] and [g[3] for g in got] == [[1, 1], [1, 0]]
