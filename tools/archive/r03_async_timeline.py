#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/archive/r03_async.py:
durations of the kernels and copies of the last calls, how much of the wall time each
engine is busy, and one call's sequence of commands with the gaps between them."""
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1]
kernels = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
copies = glob.glob(os.path.join(root, '**', '*memory_copy_trace.csv'), recursive=True)
events = []
for row in csv.DictReader(open(kernels[0])):
    name = row['Kernel_Name'].split('(')[0].replace('void tc::', '')[:40]
    events.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), 'K ' + name,
                   row.get('Queue_Id', '')))
if copies:
    for row in csv.DictReader(open(copies[0])):
        events.append((int(row['Start_Timestamp']), int(row['End_Timestamp']),
                       'C ' + row.get('Direction', row.get('Kind', 'copy')),
                       row.get('Stream_Id', '')))
events.sort()
# the last 40 % of the run: steady state
t_lo = events[int(len(events) * 0.6)][0]
tail = [e for e in events if e[0] >= t_lo]
span = tail[-1][1] - tail[0][0]
print('steady window: %.1f us, %d commands' % (span / 1e3, len(tail)))
by_name = {}
for start, end, name, queue in tail:
    by_name.setdefault(name, []).append((end - start) / 1e3)
for name, values in sorted(by_name.items()):
    print('%-46s n=%5d mean=%8.2f us  sum=%5.1f %% of window' %
          (name, len(values), np.mean(values), 100.0 * np.sum(values) * 1e3 / span))


def busy(prefix):
    intervals = sorted((s, e) for s, e, n, q in tail if n.startswith(prefix))
    total, cur_s, cur_e = 0, None, None
    for s, e in intervals:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        total += cur_e - cur_s
    return 100.0 * total / span


print('any kernel running: %.1f %% of window; any copy running: %.1f %%' %
      (busy('K '), busy('C ')))
print('\nlast 40 commands (start us relative, duration us, what, queue/stream):')
base = tail[-40][0]
for start, end, name, queue in tail[-40:]:
    print('%9.1f %8.2f  %-44s %s' % ((start - base) / 1e3, (end - start) / 1e3, name, queue))
