#!/bin/bash
# Round 6: what the phases of predict_fused_kernel cost when a launch has the chip to itself
# (developer build build/ab/dev.so from tools/build_dev.sh; TC_FUSED_SKIP 1: no occupations,
# 2: no matrix phase, 3: neither -- prologue, reductions and stores only).
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_LIBRARY=build/ab/dev.so
F="--cpu-seconds 0 --detail 0 --steps 2000 --warmup 200 --lanes 1 --option fused=2"
for draws in 40 64 32; do
  for skip in 0 1 2 3; do
    n=10000; [ $draws = 32 ] && n=8192
    echo -n "alone, $draws draws per workgroup, $n draws, TC_FUSED_SKIP=$skip: "
    TC_FUSED_SKIP=$skip python bench.py $F --draws $n --option fused_draws=$draws 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.1f us' % (r['ms_per_step']*1e3), r['roofline']['kernel'])"
  done
done
