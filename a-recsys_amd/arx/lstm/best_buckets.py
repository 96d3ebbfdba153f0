"""Sequence-length buckets for the LSTM runner -- py3 mirror of lstm/best_buckets.py
(calculate_buckets).  Greedy: start from the cumulative length histogram cut at max_length as
one segment; repeatedly take the segment whose best split saves the most padding
(gain of cutting at length l = (segment's longest length - l) x (sequences of the segment no
longer than l, minus its first cumulative count)) and emit that split point as a bucket
boundary, until max_buckets boundaries exist."""
from __future__ import annotations


def calculate_buckets(array, max_length, max_buckets):
    counts = {}
    for _, seq in array:
        counts[len(seq)] = counts.get(len(seq), 0) + 1
    running, s = [], 0
    for length in sorted(counts):
        s += counts[length]
        running.append((length, s))                      # (length, sequences no longer than it)

    def best_point(seg):
        base = seg[0][1]
        index, maxv = 0, 0
        for i, (l, n) in enumerate(seg):
            v = (seg[-1][0] - l) * (n - base)
            if v > maxv:
                maxv, index = v, i
        return index, maxv

    end_index = 0
    for i in range(len(running) - 1, -1, -1):
        if running[i][0] <= max_length:
            end_index = i + 1
            break
    if end_index <= max_buckets:
        return [x[0] for x in running[:end_index]]
    buckets = []
    states = [(running[:end_index], 0, end_index - 1)]    # (segment, gain of its best split, split index)
    while len(buckets) < max_buckets:
        k = max(range(len(states)), key=lambda j: (states[j][1], -j))   # first of the largest gains
        seg, _, split = states.pop(k)
        buckets.append(seg[split][0])
        for part in (seg[:split + 1], seg[split + 1:]):
            if part:
                idx, gain = best_point(part)
                states.append((part, gain, idx))
    return buckets
