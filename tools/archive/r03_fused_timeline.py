#!/usr/bin/env python3
"""Kernel timeline of the pipelined step with predict_fused_kernel (one kernel per call): from
a rocprofv3 kernel trace of tools/archive/r03_clustered.py, for a steady window: per-queue kernel
durations, the gaps between consecutive kernels of a queue, and how many kernels run at once."""
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1]
path = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for row in csv.DictReader(open(path)):
    if 'predict_fused' in row['Kernel_Name']:
        rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']),
                     row.get('Queue_Id', row.get('Stream_Id', '?'))))
rows.sort()
window = rows[6000:9000]
t0, t1 = window[0][0], window[-1][1]
span = t1 - t0
print('window: %d kernels, %.2f us per step' % (len(window), span / 1e3 / len(window)))
d = np.array([e - s for s, e, q in window]) / 1e3
print('duration mean %.1f us (p10 %.1f, p50 %.1f, p90 %.1f)' %
      (d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
edges = []
for s, e, q in window:
    edges += [(s, 1), (e, -1)]
edges.sort()
level, last, time_at = 0, t0, {}
for t, step in edges:
    time_at[level] = time_at.get(level, 0) + (t - last)
    level += step
    last = t
print('kernels running: ' + ', '.join('%d: %.1f %%' % (k, 100.0 * v / span)
                                      for k, v in sorted(time_at.items())))
by_queue = {}
for s, e, q in window:
    by_queue.setdefault(q, []).append((s, e))
for q, items in sorted(by_queue.items()):
    gaps = [(s1 - e0) / 1e3 for (s0, e0), (s1, e1) in zip(items[:-1], items[1:])]
    print('queue %s: %d kernels, cycle %.1f us, gap between kernels %.2f us' %
          (q, len(items), (items[-1][0] - items[0][0]) / 1e3 / (len(items) - 1), np.mean(gaps)))
print('sample (us):')
base = window[1000][0]
for s, e, q in window[1000:1016]:
    print('  q%s %8.1f -> %8.1f (%.1f)' % (q, (s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3))
