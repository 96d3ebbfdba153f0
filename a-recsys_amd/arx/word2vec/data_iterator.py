"""Batch iterator of the word2vec-style recommenders -- py3 mirror of word2vec/data_iterator.py
(same class, constructor and generator names; run_w2v.py:261-266 picks `get_next_sg` for the
skip-gram model and `get_next_cbow` for CBOW).

`seq` is the training log flattened to (user, item) pairs in user-major, time order, every user's
run terminated by an (user, end_ind) marker.  The generators draw from numpy's GLOBAL legacy
RandomState in the same order as the reference does, so a seeded run yields the same batches bit
for bit (pinned by tests/golden/w2v_iterator.json, produced by the real reference class).
"""
from __future__ import annotations

import collections

import numpy as np


def batch_major(l, m, n):
    """[m][n] -> [n][m] (data_iterator.py:171-178)."""
    return [[l[j][i] for j in range(m)] for i in range(n)]


class DataIterator(object):
    def __init__(self, seq, end_ind, batch_size, n_skips, window, sequence):
        self.seq = seq
        self.l_seq = len(seq)
        self.end_ind = end_ind
        self.batch_size = batch_size
        self.num_skips = n_skips
        self.skip_window = window
        self.index = 0
        self.sequence = sequence

    # ---- window bookkeeping shared by the three generators -------------------------------
    def _push(self, win, count=1):
        for _ in range(count):
            win.append(self.seq[self.index])
            self.index = (self.index + 1) % self.l_seq

    def _reseed(self, win, span):
        self.index = np.random.randint(0, self.l_seq)
        self._push(win, span)

    def _pairs(self, win, center, hi_when_end, out_users, out_in, out_out):
        """Slide the window until one batch of (user, centre item, context item) triples is
        complete: per centre, up to num_skips distinct context positions are drawn; a draw is
        wasted when it lands on another user's event or on an end marker (data_iterator.py:34-54)."""
        span = win.maxlen
        filled = 0
        while filled < self.batch_size:
            user, item = win[center]
            hi = hi_when_end if (hi_when_end is not None and item == self.end_ind) else span
            taken = [center]
            for _ in range(self.num_skips):
                t = np.random.randint(0, hi)
                while t in taken:
                    t = np.random.randint(0, hi)
                ctx_user, ctx_item = win[t]
                if ctx_user != user or ctx_item == self.end_ind:
                    continue
                taken.append(t)
                out_users[filled], out_in[filled], out_out[filled] = user, item, ctx_item
                filled += 1
                if filled >= self.batch_size:
                    break
            self._push(win)

    def _two_sided(self, center_of, hi_when_end_of, wrap_input):
        span = 2 * self.skip_window + 1
        users = np.ndarray(shape=[self.batch_size], dtype=np.int32)
        i_items = np.ndarray(shape=[self.batch_size], dtype=np.int32)
        o_items = np.ndarray(shape=[self.batch_size], dtype=np.int32)
        win = collections.deque(maxlen=span)
        self._push(win, span)
        while True:
            if self.sequence:
                self._reseed(win, span)
            self._pairs(win, center_of(span), hi_when_end_of(span), users, i_items, o_items)
            yield users, ([i_items] if wrap_input else i_items), o_items

    def get_next(self):
        """Symmetric window around the centre item (data_iterator.py:15-57); for a centre that is
        an end marker only the positions before it are candidates.  The three arrays are reused
        from batch to batch."""
        return self._two_sided(lambda span: self.skip_window, lambda span: self.skip_window, False)

    def get_next_sg(self):
        """Skip-gram, history -> future only: the centre is the OLDEST event of the window and the
        contexts are drawn from the whole window (data_iterator.py:59-103).  Yields the inputs as
        a one-element list (the model's [n_input][mb] layout)."""
        return self._two_sided(lambda span: 0, lambda span: None, True)

    def get_next_cbow(self):
        """CBOW: the newest event of a (window + 1)-long history is the target, num_skips of the
        `window` older positions are sampled as inputs -- with replacement while the user's own
        history is shorter than num_skips (data_iterator.py:105-169)."""
        span = self.skip_window + 1
        center = span - 1
        users = np.ndarray(shape=[self.batch_size], dtype=np.int32)
        i_items = [[0] * self.num_skips] * self.batch_size
        o_items = np.ndarray(shape=[self.batch_size], dtype=np.int32)
        win = collections.deque(maxlen=span)
        self._push(win, span)
        user, item = win[center]
        if item == self.end_ind:
            own = 0
        else:
            own = -1 + sum(1 for k in range(center) if win[k][0] == user)
        while True:
            if self.sequence:
                raise NotImplementedError('error: not implemented')      # data_iterator.py:132-134
            filled = 0
            while filled < self.batch_size:
                user, item = win[center]
                if item == self.end_ind:                                  # a new user's run starts
                    self._push(win)
                    own = 0
                    continue
                own = min(own + 1, center)
                picks = np.random.choice(center, self.num_skips, own < self.num_skips)
                i_items[filled] = [win[j][1] for j in picks]
                users[filled] = user
                o_items[filled] = item
                filled += 1
                self._push(win)
            yield users, batch_major(i_items, self.batch_size, self.num_skips), o_items
