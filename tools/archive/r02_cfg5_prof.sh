#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for opt in "quad_order=2" "quad_order=0"; do
rm -rf gpurun_out/c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5 -- python3 tools/archive/cfg5.py float32 $opt > gpurun_out/c5.log 2>&1
echo "== $opt"; grep "ms per" gpurun_out/c5.log
python3 - <<'PY'
import csv, glob
for row in csv.DictReader(open(glob.glob('gpurun_out/c5/*/*kernel_stats.csv')[0])):
    if 'rocclr' in row['Name']: continue
    print('  %-60s calls %6s  avg %9.2f us' % (row['Name'][:60], row['Calls'], float(row['AverageNs'])/1e3))
PY
done
