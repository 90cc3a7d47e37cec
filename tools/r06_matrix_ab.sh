#!/bin/bash
# Round 6: the latency form's matrix phase alone (developer build, TC_FUSED_SKIP=1: no occupations)
# and the fixed part (TC_FUSED_SKIP=3), for A/B runs of kernel variants.
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_LIBRARY=build/ab/dev.so
F="--cpu-seconds 0 --detail 0 --steps 3000 --warmup 300 --lanes 1 --option fused=2 --option fused_draws=40"
for skip in 0 1 3; do
  echo -n "TC_FUSED_SKIP=$skip: "
  TC_FUSED_SKIP=$skip python bench.py $F 2>/dev/null | tail -1 | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f us' % (r['ms_per_step']*1e3))"
done
