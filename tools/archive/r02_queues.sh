#!/bin/bash
# Developer diagnosis (knob build build/ab/dev.so): does the number of hardware queues change
# how the lanes overlap?  Then kernel timelines with and without the occupation kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
run() {
  env "$@" python bench.py --cpu-seconds 0 --other-configs 0 --steps 6000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-60s ms/step %.4f contract %.4f (overlapped %.4f)' % ('$*', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['overlapped_launch_ms']))"
}
run A=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run TC_SKIP_OCC=1
run TC_SKIP_OCC=1 GPU_MAX_HW_QUEUES=8
run TC_SKIP_FINALIZE=1 GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
for mode in A TC_SKIP_OCC TC_SKIP_FINALIZE; do
  export $mode=1
  rocprofv3 --kernel-trace -d gpurun_out/queues_$mode -o t -- python3 bench.py --cpu-seconds 0 --other-configs 0 --steps 600 --warmup 100 > /dev/null 2>&1
  echo "== $mode"; python tools/archive/timeline.py gpurun_out/queues_$mode | head -40
  unset $mode
done
