#!/bin/bash
# Developer loop: the driver's short run (--steps 20 --warmup 5), repeated.
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3 4 5 6; do
python bench.py --gpus 1 --cpu-seconds 0 --other-configs 0 --steps 20 --warmup 5 "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('%.4g calls/s  %.2f us/step  settle %d  timing %s' % (d['value'], d['ms_per_step'] * 1e3, d['settle_steps'], d.get('timed_region_breakdown_us')))"
done
