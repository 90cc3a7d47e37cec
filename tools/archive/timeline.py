"""Developer helper: overlap analysis of a rocprofv3 kernel trace (CSV) of bench.py.

Prints, for the steady state, the mean start-to-start period per kernel and how the
kernels of different lanes overlap in time."""
import csv
import glob
import sys

import numpy as np

path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[-1]
rows = []
with open(path) as stream:
    for row in csv.DictReader(stream):
        name = row['Kernel_Name']
        kind = ('contract' if 'contract' in name else 'occ' if 'occ_' in name else
                'finalize' if 'finalize' in name else None)
        if kind:
            rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), kind))
rows.sort()
rows = rows[len(rows) // 3: -30]          # steady state
t0 = rows[0][0]
by = {}
for s, e, k in rows:
    by.setdefault(k, []).append((s - t0, e - t0))
for k, v in by.items():
    v = np.array(v)
    print('%-9s n=%d duration %.1f us, period %.1f us' % (
        k, len(v), np.mean(v[:, 1] - v[:, 0]) / 1e3, np.mean(np.diff(v[:, 0])) / 1e3))
# time with no contraction running
c = np.array(by['contract'])
gaps = c[1:, 0] - np.maximum.accumulate(c[:-1, 1])
print('gap between contractions: mean %.1f us (negative = overlap)' % (np.mean(gaps) / 1e3))
print('first 12 events (us):')
for s, e, k in rows[:12]:
    print('  %-9s %8.1f -> %8.1f' % (k, (s - t0) / 1e3, (e - t0) / 1e3))
