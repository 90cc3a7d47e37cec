#!/usr/bin/env python3
"""Where the pipelined step spends its time: from a rocprofv3 kernel trace of
tools/archive/r03_clustered.py (device-resident steps only), for a steady window: how long 0 / 1 / 2+
contraction kernels are running, the per-queue chain (kernel durations and the gaps between a
kernel's end and the next kernel's start on the same queue), and a sample of the timeline."""
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1]
path = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for row in csv.DictReader(open(path)):
    name = row['Kernel_Name']
    kind = 'C' if 'contract' in name else 'O' if 'occ_' in name else 'F' if 'finalize' in name else None
    if kind:
        rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), kind,
                     row.get('Queue_Id', row.get('Stream_Id', '?'))))
rows.sort()
# steady window: steps 6000-9000 of the first draw set (4000 warm-up + 10000 timed steps)
n_calls = len(rows) // 3
lo, hi = 3 * 6000, 3 * 9000
window = rows[lo:hi]
t0, t1 = window[0][0], window[-1][1]
span = t1 - t0
print('window: %d kernels, %.1f us per step' % (len(window), span / 1e3 / (len(window) / 3)))
edges = []
for s, e, k, q in window:
    if k == 'C':
        edges.append((s, 1))
        edges.append((e, -1))
edges.sort()
level, last, time_at = 0, t0, {}
for t, d in edges:
    time_at[level] = time_at.get(level, 0) + (t - last)
    level += d
    last = t
print('contractions running: ' + ', '.join('%d: %.1f %%' % (k, 100.0 * v / span)
                                           for k, v in sorted(time_at.items())))
for kind in 'OCF':
    d = np.array([e - s for s, e, k, q in window if k == kind]) / 1e3
    print('%s duration mean %.1f us (p10 %.1f, p90 %.1f)' % (kind, d.mean(), np.percentile(d, 10),
                                                           np.percentile(d, 90)))
by_queue = {}
for s, e, k, q in window:
    by_queue.setdefault(q, []).append((s, e, k))
for q, items in sorted(by_queue.items()):
    gaps = {}
    for (s0, e0, k0), (s1, e1, k1) in zip(items[:-1], items[1:]):
        gaps.setdefault(k0 + '->' + k1, []).append((s1 - e0) / 1e3)
    cycle = (items[-1][0] - items[0][0]) / 1e3 / (len(items) / 3)
    print('queue %s: %d kernels, cycle %.1f us; gaps ' % (q, len(items), cycle) +
          ', '.join('%s %.1f' % (k, np.mean(v)) for k, v in sorted(gaps.items())))
print('sample (us):')
base = window[3000][0]
for s, e, k, q in window[3000:3036]:
    print('  %s q%s %8.1f -> %8.1f (%.1f)' % (k, q, (s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3))
