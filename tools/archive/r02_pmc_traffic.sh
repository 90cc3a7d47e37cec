#!/bin/bash
# Developer loop: HBM-side traffic of the step's kernels (two PMC passes) + serialised stats.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
FAST="--cpu-seconds 0 --other-configs 0 --settle-seconds 0.05"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- \
    python3 bench.py --lanes 1 --steps 50 --warmup 5 $FAST > $OUT/pmc_$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_1 $OUT/pmc_2 | grep -v rocclr
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l1 -- python3 bench.py --lanes 1 --steps 1000 --warmup 100 $FAST > $OUT/l1.log 2>&1
python3 - <<'PY'
import csv, glob
for row in csv.DictReader(open(glob.glob('gpurun_out/pmc_traffic/l1/*/*kernel_stats.csv')[0])):
    print('  %-60s calls %6s  avg %9.2f us' % (row['Name'][:60], row['Calls'], float(row['AverageNs'])/1e3))
PY
