#!/bin/bash
# Developer sweep on one box: launch geometry of the occupation / finalisation kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --cpu-seconds 0 --other-configs 0 --steps 6000 --warmup 300 --settle-seconds 0.1 "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-70s %.4g calls/s  %.2f us/step  overlapped kernel %.2f' % ('$*', d['value'], d['ms_per_step'] * 1e3, r['overlapped_launch_ms'] * 1e3))"; }
run
for sp in 1 3 4 5 7 10 13; do run --option occ_splits=$sp; done
for pc in 2 3 6 8; do run --option occ_per_cu=$pc; done
run --option occ_splits=5 --option occ_per_cu=8
run --option finalize_threads=512
run --option finalize_threads=1024
run --option finalize_row_blocks=2
run --option prio_finalize=2
run --option prio_finalize=1
run
