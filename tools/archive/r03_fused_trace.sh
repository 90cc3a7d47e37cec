#!/bin/bash
# Kernel timeline of the pipelined step with predict_fused_kernel (tools/archive/r03_fused_timeline.py).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fused_trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- \
  python3 $GRAFT_REPO_ROOT/tools/archive/r03_clustered.py "$@" > $OUT/run.log 2>&1
grep "us per step" $OUT/run.log
python3 $GRAFT_REPO_ROOT/tools/archive/r03_fused_timeline.py $OUT
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
