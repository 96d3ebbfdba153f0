"""Host-side batch assembly of the sequence model (lstm/seqModel.py:356-452), GPU-free so that
it can be tested on its own.  `SeqBatching` is mixed into arx.lstm.seqModel.SeqModel; it needs
`buckets`, `batch_size`, `START_ID`, `PAD_ID`, `USER_PAD_ID` on the instance.

Row layout returned to the caller (and expected by SeqModel.step / step_recommend): time-major
lists, entry t = the batch_size values of time step t.
"""
from __future__ import annotations

import random


class SeqBatching(object):
    def _draw(self, data_set, bucket_id, start_id, slot):
        """(user, item sequence, is a real example) for batch slot `slot`: a random example of
        the bucket when start_id is None (training; one random.choice per slot, in slot order),
        else the start_id + slot-th one, or padding past the end of the bucket."""
        rows = data_set[bucket_id]
        if start_id is None:
            user, seq = random.choice(rows)
            return user, list(seq), True
        if start_id + slot < len(rows):
            user, seq = rows[start_id + slot]
            return user, list(seq), True
        return self.USER_PAD_ID, [], False

    def _time_major(self, per_example, length):
        return [[per_example[b][t] for b in range(self.batch_size)] for t in range(length)]

    def _finished(self, data_set, bucket_id, start_id):
        return start_id is not None and start_id + self.batch_size >= len(data_set[bucket_id])

    def get_batch(self, data_set, bucket_id, start_id=None):
        """seqModel.py:356-404 -> (users, inputs, targets, weights, finished).  Example with items
        s_0..s_{k-1} in a bucket of length L: input = [START, s_0..s_{k-2}] + PAD, target =
        s_0..s_{k-1} + PAD, weight = 1 on the k real targets.  An empty slot feeds START then
        PAD with all-zero weights."""
        L = self.buckets[bucket_id]
        users, inputs, targets, weights = [], [], [], []
        for slot in range(self.batch_size):
            user, seq, _ = self._draw(data_set, bucket_id, start_id, slot)
            k = len(seq)
            shifted = [self.START_ID] + seq[:k - 1] if k else [self.START_ID]
            users.append(user)
            inputs.append(shifted + [self.PAD_ID] * (L - len(shifted)))
            targets.append(seq + [self.PAD_ID] * (L - k))
            weights.append([1.0] * k + [0.0] * (L - k))
        return (users, self._time_major(inputs, L), self._time_major(targets, L),
                self._time_major(weights, L), self._finished(data_set, bucket_id, start_id))

    def get_batch_recommend(self, data_set, bucket_id, start_id=None):
        """seqModel.py:407-452 -> (users, inputs, positions, valids, finished): the sequence itself
        (no START shift) padded to the bucket length; positions[b] = index of the example's last
        item (the step whose top-k is the recommendation), L-1 for an empty slot; valids[b] = 0
        for padding slots past the end of the bucket."""
        L = self.buckets[bucket_id]
        users, inputs, positions, valids = [], [], [], []
        for slot in range(self.batch_size):
            user, seq, real = self._draw(data_set, bucket_id, start_id, slot)
            users.append(user)
            inputs.append(seq + [self.PAD_ID] * (L - len(seq)))
            positions.append(len(seq) - 1 if real else L - 1)
            valids.append(1 if real else 0)
        return (users, self._time_major(inputs, L), positions, valids,
                self._finished(data_set, bucket_id, start_id))
