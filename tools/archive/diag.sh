#!/bin/bash
# Developer diagnosis on one GPU box with the knob-enabled build (build/ab/dev.so):
# what the pipelined step costs without the occupation / finalisation kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
run() {
  env "$@" python bench.py --cpu-seconds 0 --other-configs 0 --steps 6000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-46s ms/step %.4f contract %.4f (overlapped %.4f)' % ('$*', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['overlapped_launch_ms']))"
}
run A=0
run TC_SKIP_OCC=1
run TC_SKIP_FINALIZE=1
run TC_SKIP_OCC=1 TC_SKIP_FINALIZE=1
run TC_LANES=1
run TC_LANES=2
run TC_LANES=4
run TC_LANES=1 TC_SKIP_OCC=1 TC_SKIP_FINALIZE=1
run TC_LANES=2 TC_SKIP_OCC=1 TC_SKIP_FINALIZE=1
