#!/bin/bash
# Kernel timeline of the device-resident pipelined step only (tools/archive/r03_clustered.py: 14 000
# steps of 10^4 draws on four lanes), analysed by tools/r03_pipeline_timeline.py.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pipe_trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- \
  python3 $GRAFT_REPO_ROOT/tools/archive/r03_clustered.py "$@" > $OUT/run.log 2>&1
cat $OUT/run.log | grep "us per step"
python3 $GRAFT_REPO_ROOT/tools/archive/r03_pipeline_timeline.py $OUT
