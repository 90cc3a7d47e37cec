#!/bin/bash
# Does rocprofv3 --kernel-trace change a kernel's duration?  Same box, same command, with and
# without the profiler: in-process per-launch events (bench.py) against the profiler's record.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for tag in cfg5f64 cfg5f32; do
  for i in 1 2; do
    python3 bench.py --only-config $tag --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print('plain      ', k, round(v['kernel_us'],1), 'us, step', round(1e6*v['draws_per_call']/v['device_calls_per_sec'],1), 'us')"
    rm -rf gpurun_out/po; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/po -- python3 bench.py --only-config $tag --cpu-seconds 0 2>/dev/null | grep -E '^\{' | tail -1 | python3 -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print('under trace', k, round(v['kernel_us'],1), 'us, step', round(1e6*v['draws_per_call']/v['device_calls_per_sec'],1), 'us')"
    grep contract gpurun_out/po/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
  done
done
