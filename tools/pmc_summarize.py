"""Reduce a rocprofv3 counter_collection.csv to per-kernel averages of one counter.

usage: pmc_summarize.py <counter_collection.csv> <COUNTER>  ->  CSV on stdout:
kernel, dispatches, mean, min, max  (counter value per dispatch, raw units as rocprofv3
reports them: FETCH_SIZE / WRITE_SIZE are in KiB).
"""
import csv
import sys
from collections import defaultdict


def main():
    path, counter = sys.argv[1], sys.argv[2]
    agg = defaultdict(list)
    per_dispatch = defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter:
                continue
            # one row per (dispatch, counter[, dimension instance]): sum instances
            per_dispatch[(r['Kernel_Name'], r['Dispatch_Id'])] += float(r['Counter_Value'])
    for (k, _), v in per_dispatch.items():
        agg[k].append(v)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'dispatches', 'mean_' + counter, 'min', 'max'])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k[:160], len(v), '%.3f' % (sum(v) / len(v)), '%.3f' % min(v), '%.3f' % max(v)])


if __name__ == '__main__':
    main()
