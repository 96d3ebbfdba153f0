"""Print a rocprofv3 kernel_stats.csv as: calls, average us, total us, short kernel name."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[:top]:
    name = re.sub(r'\(.*$', '', r['Name'])
    name = re.sub(r'^void ', '', name).replace('arx::', '').replace('(anonymous namespace)::', '')
    print('%6d  %9.1f us  %10.1f us  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3,
                                          float(r['TotalDurationNs']) / 1e3, name[:100]))
