#!/bin/bash
# Developer loop: serialised per-kernel rocprofv3 stats and the 4-lane timeline of bench.py.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/qp_l1 gpurun_out/qp_l4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qp_l1 -- \
  python3 bench.py --lanes 1 --steps 1500 --warmup 200 --cpu-seconds 0 > gpurun_out/qp_l1.log 2>&1
cat gpurun_out/qp_l1/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110 | head -6
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qp_l4 -- \
  python3 bench.py --steps 600 --warmup 100 --cpu-seconds 0 > gpurun_out/qp_l4.log 2>&1
python3 tools/archive/timeline.py gpurun_out/qp_l4
python3 - <<'PY'
import csv, glob
path = glob.glob('gpurun_out/qp_l4/**/*kernel_trace.csv', recursive=True)[-1]
rows = []
for row in csv.DictReader(open(path)):
    n = row['Kernel_Name']
    kind = 'C' if 'contract' in n else 'O' if 'occ_' in n else 'F' if 'finalize' in n else None
    if kind: rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), kind, row.get('Stream_Id', row.get('Queue_Id', '?'))))
rows.sort()
rows = rows[len(rows)//2: len(rows)//2 + 36]
t0 = rows[0][0]
for s, e, k, q in rows: print('%s q%s %8.1f -> %8.1f  (%.1f)' % (k, q, (s-t0)/1e3, (e-t0)/1e3, (e-s)/1e3))
PY
