"""Bucketed batch iterator of the LSTM runner -- py3 mirror of lstm/data_iterator.py (same class
and method names).  `model` is anything with get_batch(data_set, bucket_id, start_id=None) ->
(users, inputs, outputs, weights, finished) -- arx.lstm.seqModel.SeqModel.get_batch."""
from __future__ import annotations

import numpy as np

PAD_ID = 0
START_ID = 1


class DataIterator(object):
    def __init__(self, model, data_set, n_bucket, batch_size, train_buckets_scale):
        self.data_set = data_set
        self.n_bucket = n_bucket
        self.batch_size = batch_size
        self.train_buckets_scale = train_buckets_scale
        self.model = model

    def next_random(self):
        """Endless training stream: a bucket drawn in proportion to its share of the data
        (train_buckets_scale = cumulative shares), then a random batch of it
        (data_iterator.py:14-22)."""
        scale = self.train_buckets_scale
        while True:
            x = np.random.random_sample()
            bucket_id = min(i for i in range(len(scale)) if scale[i] > x)
            users, inputs, outputs, weights, _ = self.model.get_batch(self.data_set, bucket_id)
            yield users, inputs, outputs, weights, bucket_id

    def next_sequence(self, stop=False, recommend=False):
        """Every bucket front to back in batch_size strides (evaluation / recommendation);
        `stop` ends after one sweep, otherwise the sweep repeats (data_iterator.py:24-42)."""
        fetch = self.model.get_batch_recommend if recommend else self.model.get_batch
        while True:
            for bucket_id in range(self.n_bucket):
                start_id = 0
                while True:
                    users, inputs, outputs, weights, finished = fetch(self.data_set, bucket_id,
                                                                      start_id=start_id)
                    yield users, inputs, outputs, weights, bucket_id
                    if finished:
                        break
                    start_id += self.batch_size
            if stop:
                return
