#!/bin/bash
# Developer A/B: launch geometries against the SERIAL chain (host-buffer API calls: one lane,
# occupation -> contraction -> finalisation).
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  python bench.py --cpu-seconds 0 --other-configs 0 --steps 2000 --warmup 200 $1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-64s step %.2f us  serial step %.2f us  host-to-host %.1f us per call' % ('$1', d['ms_per_step'] * 1e3, r['serialised_step_ms'] * 1e3, d['host_to_host']['ms_per_call'] * 1e3))"
}
run ""
run "--option finalize_row_blocks=2"
run "--option finalize_row_blocks=4"
run "--option finalize_row_blocks=4 --option finalize_threads=512"
run "--option finalize_row_blocks=2 --option finalize_threads=1024"
run "--option finalize_threads=1024"
run ""
