"""Per-step busy / idle accounting from a rocprofv3 kernel trace.

usage: python tools/trace_gaps.py <kernel_trace.csv> <marker substring> [skip_steps]
Steps are delimited by successive launches of the marker kernel (one per step); prints, for
the median step, wall time, summed kernel time, idle time between kernels and the kernel list."""
import csv
import sys

f, marker = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
steps = []
for a, b in zip(marks[skip:-1], marks[skip + 1:]):
    seg = rows[a:b]
    wall = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    # overlap-aware busy: union of intervals
    cur_s, cur_e, uni = seg[0][0], seg[0][1], 0
    for s, e, _ in seg[1:]:
        if s > cur_e:
            uni += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    uni += cur_e - cur_s
    if len(seg) >= 10:
        steps.append((wall, busy, uni, seg))
steps.sort(key=lambda t: t[0])
wall, busy, uni, seg = steps[len(steps) // 2]
print("steps %d  median wall %.1f us  kernel sum %.1f us  busy(union) %.1f us  idle %.1f us  launches %d"
      % (len(steps), wall / 1e3, busy / 1e3, uni / 1e3, (wall - uni) / 1e3, len(seg)))
prev_e = None
for s, e, n in seg:
    gap = 0 if prev_e is None else (s - prev_e) / 1e3
    print("  %+7.1f gap  %8.1f us  %s" % (gap, (e - s) / 1e3, n[:90].replace('void ', '')))
    prev_e = max(prev_e, e) if prev_e else e
