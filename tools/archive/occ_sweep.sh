#!/bin/bash
# Developer sweep of the occupation kernel's decomposition (serialized steps: differences
# in ms/step are differences in the occupation kernel).
cd "$GRAFT_REPO_ROOT" || exit 1
for per_cu in 4; do
  for splits in 1 2 3 5 6 5 13; do
    TC_OCC_SPLITS=$splits TC_OCC_PER_CU=$per_cu python bench.py --cpu-seconds 0 --steps 200 \
      2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('per_cu $per_cu splits $splits ms/step %.4f' % d['ms_per_step'])"
  done
done
