"""Flat {name -> array} checkpoints carrying the reference's variable names.

Replaces tf.train.Saver(tf.global_variables()) (hmf_model.py:156, seqModel.py:184):
tables 'userembed_cat_0', 'itemembed_mulhot_0', biases 'item_bias_cat_0' ([Vf,1]),
dense weights, and the Adagrad slots as '<name>/Adagrad'.

Like tf.train.Saver, save() also maintains a small text file `checkpoint` next to the data file
naming the latest checkpoint, so runner code written as
    ckpt = get_checkpoint_state(dir);  saver.restore(sess, ckpt.model_checkpoint_path)
(run_hmf.py:131-137, lstm/run.py:345-353) finds it: `get_checkpoint_state` / `latest_checkpoint`
below."""
from __future__ import annotations

import os

import numpy as np
import torch


INDEX_NAME = 'checkpoint'


class CheckpointState(object):
    """The two fields of tf.train.get_checkpoint_state()'s result the runners read."""

    def __init__(self, model_checkpoint_path, all_model_checkpoint_paths):
        self.model_checkpoint_path = model_checkpoint_path
        self.all_model_checkpoint_paths = all_model_checkpoint_paths


def get_checkpoint_state(checkpoint_dir):
    """None when `checkpoint_dir` holds no index file (the runners then initialise fresh)."""
    idx = os.path.join(checkpoint_dir, INDEX_NAME)
    if not os.path.isfile(idx):
        return None
    latest, every = None, []
    for line in open(idx):
        key, _, val = line.strip().partition(': ')
        val = val.strip('"')
        if not os.path.isabs(val):
            val = os.path.join(checkpoint_dir, val)
        if key == 'model_checkpoint_path':
            latest = val
        elif key == 'all_model_checkpoint_paths':
            every.append(val)
    return CheckpointState(latest, every) if latest else None


def latest_checkpoint(checkpoint_dir):
    st = get_checkpoint_state(checkpoint_dir)
    return st.model_checkpoint_path if st else None


def _update_index(path):
    d = os.path.dirname(path) or '.'
    st = get_checkpoint_state(d)
    every = [os.path.relpath(p, d) for p in (st.all_model_checkpoint_paths if st else [])]
    rel = os.path.relpath(path, d)
    every = [p for p in every if p != rel] + [rel]
    with open(os.path.join(d, INDEX_NAME), 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % rel)
        for p in every:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)


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
        _update_index(path)
        return path

    def restore(self, session, path):
        if not path.endswith('.npz'):
            path = path + '.npz'
        z = np.load(path)
        m = self.model
        rt = m.rt
        # every variable (and slot) of THIS model must be in the file, with its shape: a checkpoint
        # of a differently configured model (nonlinear / use_concat / other attribute set) is refused
        # before anything is overwritten
        want = {}
        for t in m.att_emb.tables.values():
            want[t.name] = want[t.name + '/Adagrad'] = tuple(t.E.shape)
            if t.bias is not None:
                want[t.bias_name] = want[t.bias_name + '/Adagrad'] = (int(t.bias.shape[0]), 1)
        for p in rt.dense.values():
            want[p.name] = want[p.name + '/Adagrad'] = tuple(p.w.shape)
        missing = sorted(k for k in want if k not in z.files)
        if missing:
            raise KeyError("checkpoint %s lacks %d variable(s) of this model: %s%s"
                           % (path, len(missing), ', '.join(missing[:6]), ' ...' if len(missing) > 6 else ''))
        bad = sorted(k for k, shp in want.items() if tuple(z[k].shape) != shp)
        if bad:
            raise ValueError("checkpoint %s: shape mismatch for %s (file %s, model %s)"
                             % (path, bad[0], tuple(z[bad[0]].shape), want[bad[0]]))
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
