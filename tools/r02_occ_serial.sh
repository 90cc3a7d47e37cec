#!/bin/bash
# Developer A/B: occupation launch geometry against the SERIAL chain (host-buffer API calls:
# one lane, occupation -> contraction -> finalisation).
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  python bench.py --cpu-seconds 0 --other-configs 0 --steps 2000 --warmup 200 $1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-44s step %.2f us  serial step %.2f us  host-to-host %.1f us per call' % ('$1', d['ms_per_step'] * 1e3, r['serialised_step_ms'] * 1e3, d['host_to_host']['ms_per_call'] * 1e3))"
}
run ""
run "--option occ_splits=4 --option occ_per_cu=4"
run "--option occ_splits=5 --option occ_per_cu=4"
run "--option occ_splits=8 --option occ_per_cu=4"
run "--option occ_splits=8 --option occ_per_cu=8"
run "--option occ_splits=13 --option occ_per_cu=8"
run ""
