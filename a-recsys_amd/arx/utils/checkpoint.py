"""Flat {name -> array} checkpoints carrying the reference's variable names.

Replaces tf.train.Saver(tf.global_variables()) (hmf_model.py:156, seqModel.py:184):
tables 'userembed_cat_0', 'itemembed_mulhot_0', biases 'item_bias_cat_0' ([Vf,1]),
dense weights, and the Adagrad slots as '<name>/Adagrad'."""
from __future__ import annotations

import os

import numpy as np
import torch


class Saver(object):
    def __init__(self, model):
        self.model = model

    def _state(self):
        m = self.model
        rt = m.rt
        st = {}
        for t in m.att_emb.tables.values():
            st[t.name] = t.E.cpu().numpy()
            st[t.name + '/Adagrad'] = t.acc.cpu().numpy()
            if t.bias is not None:
                st[t.bias_name] = t.bias.cpu().numpy().reshape(-1, 1)
                st[t.bias_name + '/Adagrad'] = t.bias_acc.cpu().numpy().reshape(-1, 1)
        for p in rt.dense.values():
            st[p.name] = p.w.cpu().numpy()
            st[p.name + '/Adagrad'] = p.acc.cpu().numpy()
        st['global_step'] = np.asarray(rt.global_step, dtype=np.int64)
        st['learning_rate'] = np.asarray(rt.lr_host, dtype=np.float32)
        return st

    def save(self, session, path, global_step=None, write_meta_graph=False):
        if global_step is not None:
            path = '%s-%d' % (path, global_step)
        d = os.path.dirname(path)
        if d and not os.path.isdir(d):
            os.makedirs(d)
        np.savez(path + '.npz', **self._state())
        return path

    def restore(self, session, path):
        if not path.endswith('.npz'):
            path = path + '.npz'
        z = np.load(path)
        m = self.model
        rt = m.rt
        for t in m.att_emb.tables.values():
            t.E.copy_(torch.from_numpy(z[t.name]))
            t.acc.copy_(torch.from_numpy(z[t.name + '/Adagrad']))
            if t.bias is not None:
                t.bias.copy_(torch.from_numpy(z[t.bias_name].reshape(-1)))
                t.bias_acc.copy_(torch.from_numpy(z[t.bias_name + '/Adagrad'].reshape(-1)))
        for p in rt.dense.values():
            p.w.copy_(torch.from_numpy(z[p.name]))
            p.acc.copy_(torch.from_numpy(z[p.name + '/Adagrad']))
        rt.global_step = int(z['global_step'])
        rt.set_learning_rate(float(z['learning_rate']))
