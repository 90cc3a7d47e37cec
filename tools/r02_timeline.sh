#!/bin/bash
# Developer loop: kernel timeline of the pipelined regime (four lanes) from a rocprofv3 trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- \
  python3 bench.py --steps 600 --warmup 100 --cpu-seconds 0 --other-configs 0 --settle-seconds 0.05 > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob
path = glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True)[-1]
rows = []
for row in csv.DictReader(open(path)):
    n = row['Kernel_Name']
    kind = 'C' if 'contract' in n else 'O' if 'occ_' in n else 'F' if 'finalize' in n else None
    if kind: rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), kind, row.get('Stream_Id', row.get('Queue_Id', '?'))))
rows.sort()
# the timed region: a long stretch of back-to-back steps; take 48 records from the middle of the first 2000
rows = rows[900:948]
t0 = rows[0][0]
for s, e, k, q in rows: print('%s q%s %8.1f -> %8.1f  (%.1f)' % (k, q, (s-t0)/1e3, (e-t0)/1e3, (e-s)/1e3))
# fraction of time with a contraction kernel running
cs = [(s, e) for s, e, k, q in rows if k == 'C']
span = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = cs[0]
for s, e in cs[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('contraction kernels cover %.1f %% of the span; %d contractions in %.1f us' % (100.0 * busy / span, len(cs), span / 1e3))
PY
