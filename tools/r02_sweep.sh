#!/bin/bash
# Developer sweep on one box: wave priorities of the three kernels and the number of lanes.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --cpu-seconds 0 --other-configs 0 --steps 6000 --warmup 300 --settle-seconds 0.1 "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-70s %.4g calls/s  %.2f us/step  overlapped kernel %.2f' % ('$*', d['value'], d['ms_per_step'] * 1e3, r['overlapped_launch_ms'] * 1e3))"; }
run
run --option prio_occ=1
run --option prio_occ=2
run --option prio_occ=3
run --option prio_occ=2 --option prio_contract=0
run --option prio_occ=1 --option prio_contract=0
run --lanes 6
run --lanes 8
run --lanes 6 --option prio_occ=2
run --lanes 8 --option prio_occ=2
run --lanes 3 --option prio_occ=2
run
