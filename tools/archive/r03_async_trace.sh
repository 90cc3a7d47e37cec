#!/bin/bash
# Kernel + memory-copy timeline of the asynchronous host-to-host path (no counters).
#   gpurun -- bash tools/archive/r03_async_trace.sh [predict:4:400]
cd /tmp && export TMPDIR=/tmp
ONLY=${1:-predict:4:400}
OUT=$GRAFT_REPO_ROOT/gpurun_out/async_trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o trace -- \
  python3 $GRAFT_REPO_ROOT/tools/archive/r03_async.py --only $ONLY > $OUT/run.log 2>&1
tail -3 $OUT/run.log
find $OUT -name '*.csv' | head
python3 $GRAFT_REPO_ROOT/tools/archive/r03_async_timeline.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
