#!/bin/bash
# Kernel timeline of a 20-step burst (tools/archive/r03_burst.py) under rocprofv3.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/burst
rm -rf $OUT && mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/archive/r03_burst.py "$@"
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- \
  python3 $GRAFT_REPO_ROOT/tools/archive/r03_burst.py "$@" > $OUT/run.log 2>&1
cat $OUT/run.log | tail -1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob('$OUT/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'predict_fused' in row['Kernel_Name'] or 'contract_quad' in row['Kernel_Name']:
            rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), row.get('Queue_Id', '?')))
rows.sort()
burst = rows[-20:]            # the last burst
t0 = burst[0][0]
for s, e, q in burst:
    print('  q%s %8.1f -> %8.1f us (%.1f)' % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print('burst span %.1f us = %.2f us per step' % ((burst[-1][1] - t0) / 1e3, (burst[-1][1] - t0) / 1e3 / 20))
PY
rm -rf $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
