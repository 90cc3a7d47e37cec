#!/bin/bash
# Kernel times of the pair counter at tabulation scale (host sorting excluded).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pc_prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pc -- \
  python3 $GRAFT_REPO_ROOT/tools/archive/paircount_bench.py --no-oracle > $OUT/run.log 2>&1
cat $OUT/run.log | grep points
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'pair_count' in row['Kernel_Name']:
            print('%-60s %10.2f ms  grid %s' % (row['Kernel_Name'][:60], (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6, row.get('Grid_Size_X', row.get('Grid_Size', ''))))
PY
