#!/bin/bash
# Developer A/B: the likelihood API in the pipelined bench (no communicator) against the plain
# prediction, with kernel timelines.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for mode in "" "--gather chi2"; do
  python bench.py --cpu-seconds 0 --other-configs 0 --steps 6000 --warmup 300 $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-16s ms/step %.4f  enqueue %.0f us of %.0f' % ('$mode', d['ms_per_step'], d['timed_region_breakdown_us']['enqueue'], d['ms_per_step'] * 1e3 * d['steps']))"
  rocprofv3 --kernel-trace -d gpurun_out/tl_chi2 -o t -- python3 bench.py --cpu-seconds 0 --other-configs 0 --steps 600 --warmup 100 $mode > /dev/null 2>&1
  python tools/archive/timeline_db.py gpurun_out/tl_chi2/t_results.db 16
  rm -rf gpurun_out/tl_chi2
done
