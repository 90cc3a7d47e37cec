#!/bin/bash
# Developer loop: un-batched latencies and where the interpolator's 150 us go.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python tools/archive/latency.py
python tools/archive/latency_interp.py
rm -rf gpurun_out/lat_interp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lat_interp -- python3 tools/archive/latency_interp_trace.py > /dev/null 2>&1
cat gpurun_out/lat_interp/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
