#!/bin/bash
# Developer A/B on one GPU box: default bench with build/ab/${OLD:-old}.so vs the in-tree library,
# alternating.  Extra arguments are passed to bench.py.
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export TABCORR_AMD_LIBRARY=$PWD/build/ab/${OLD:-old}.so; else unset TABCORR_AMD_LIBRARY; fi
    python bench.py --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$which calls/s %.4g ms/step %.4f contract %.4f' % (d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms']))"
  done
done
