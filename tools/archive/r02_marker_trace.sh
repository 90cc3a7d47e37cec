#!/bin/bash
# roctx ranges of the library (upload / occupation / contraction + finalisation / download /
# gather) as rocprofv3 --marker-trace sees them, next to the kernel trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/markers
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/markers -- \
  python3 tools/archive/host_sizes.py > gpurun_out/markers.log 2>&1
ls gpurun_out/markers/*/ | head -20
for f in gpurun_out/markers/*/*marker*stats*.csv gpurun_out/markers/*/*domain_stats.csv; do echo "== $f"; head -12 "$f" | cut -c1-160; done
