"""Developer helper: the pipelined stretch of a rocprofv3 kernel trace (rocpd database) of
bench.py: mean duration per kernel kind while the lanes rotate, and a sample of the timeline."""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute('select name, start, end, queue_id from kernels order by start'))
kinds = (('C', 'contract'), ('O', 'occ_'), ('F', 'finalize'))
rows = [(s, e, next((k for k, word in kinds if word in n), None), q) for n, s, e, q in rows]
rows = [r for r in rows if r[2]]
index = [i for i in range(len(rows) - 12) if len({r[3] for r in rows[i:i + 12]}) >= 3]
best, seg = [], []
for i in index:
    if seg and i != seg[-1] + 1:
        best, seg = (seg if len(seg) > len(best) else best), []
    seg.append(i)
best = seg if len(seg) > len(best) else best
rows = rows[best[0] + len(best) // 4: best[0] + 3 * len(best) // 4]
t0 = rows[0][0]
for kind, _ in kinds:
    v = np.array([(s, e) for s, e, k, q in rows if k == kind])
    if len(v):
        print('  %s n=%d duration %.1f us (p10 %.1f p90 %.1f)  period %.1f us' % (
            kind, len(v), np.mean(v[:, 1] - v[:, 0]) / 1e3,
            np.percentile(v[:, 1] - v[:, 0], 10) / 1e3, np.percentile(v[:, 1] - v[:, 0], 90) / 1e3,
            np.mean(np.diff(v[:, 0])) / 1e3))
if len(sys.argv) > 2:
    for s, e, k, q in rows[:int(sys.argv[2])]:
        print('   %s q%-2s %8.1f -> %8.1f  (%.1f)' % (k, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
