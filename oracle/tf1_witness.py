"""Auto-upgrade hook of the oracle: when a TensorFlow with the v1 graph API is importable, the
two semantics the restatement could not verify here -- TF-1.0's LSTMCell (lstm/seqModel.py:99-103) and
AdagradOptimizer with duplicate IndexedSlices (hmf/hmf_model.py:146-151) -- are checked against the REAL ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  TensorFlow is NOT in this image: `available()` is False here and on the
GPU box, `kind()` stays "port" and bench.py's cpu_baseline keeps labelling itself so.  Nothing in the product path
imports this module.
"""
from __future__ import annotations

import numpy as np

_tf = None


def _load():
    global _tf
    if _tf is None:
        try:
            import tensorflow.compat.v1 as tf     # noqa: F401  (absent in this image)
            tf.disable_v2_behavior()
            _tf = tf
        except Exception:
            _tf = False
    return _tf


def available():
    return bool(_load())


def kind():
    """Label for bench.py's cpu_baseline: "tf1" once the real graph can run, else "port"."""
    return "tf1" if available() else "port"


def check_lstm(ref_lstm, L=5, B=4, din=6, h=8, seed=3):
    """static_rnn over tf.nn.rnn_cell.LSTMCell(forget_bias=1.0) == ref_lstm.lstm_fwd, and tf.gradients == lstm_bwd."""
    tf = _load()
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((L, B, din))
    W = rng.standard_normal((din + h, 4 * h)) * 0.3
    b = rng.standard_normal(4 * h) * 0.1
    dhs = rng.standard_normal((L, B, h))
    hs, cs, gates = ref_lstm.lstm_fwd(x, W, b, 1.0)
    dz, dx, dW, db = ref_lstm.lstm_bwd(x, W, hs, cs, gates, dhs)
    g = tf.Graph()
    with g.as_default():
        cell = tf.nn.rnn_cell.LSTMCell(h, forget_bias=1.0, state_is_tuple=True, dtype=tf.float64)
        xs = [tf.constant(x[t]) for t in range(L)]
        outs, _ = tf.nn.static_rnn(cell, xs, dtype=tf.float64)
        H = tf.stack(outs)
        kern, bias = cell.trainable_variables
        loss = tf.reduce_sum(H * tf.constant(dhs))
        gk, gb = tf.gradients(loss, [kern, bias])
        with tf.Session(graph=g) as s:
            s.run(tf.global_variables_initializer())
            s.run([kern.assign(W), bias.assign(b)])
            Hn, gkn, gbn = s.run([H, gk, gb])
    np.testing.assert_allclose(Hn, hs, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gkn, dW, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gbn, db, rtol=1e-9, atol=1e-12)


def check_adagrad(rg, V=11, d=3, lr=0.7, seed=5):
    """AdagradOptimizer(lr) (initial accumulator 0.1) on an embedding_lookup gradient with duplicate ids ==
    the oracle's sum-then-apply."""
    tf = _load()
    rng = np.random.default_rng(seed)
    p0 = rng.standard_normal((V, d))
    ids = np.array([2, 2, 7, 0, 2, 7])
    w = rng.standard_normal((len(ids), d))
    g = tf.Graph()
    with g.as_default():
        E = tf.Variable(p0)
        loss = tf.reduce_sum(tf.nn.embedding_lookup(E, ids) * tf.constant(w))
        step = tf.train.AdagradOptimizer(lr).minimize(loss)
        with tf.Session(graph=g) as s:
            s.run(tf.global_variables_initializer())
            s.run(step)
            En = s.run(E)
    p, acc = p0.copy(), np.full((V, d), 0.1)
    gsum = np.zeros((V, d))
    np.add.at(gsum, ids, w)
    rg.adagrad_apply(p, acc, gsum, lr)
    np.testing.assert_allclose(En, p, rtol=1e-12, atol=1e-14)
