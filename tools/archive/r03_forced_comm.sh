#!/bin/bash
# bench.py with the communicator forced on at one rank: the driver's short command in both
# gather modes, three / four lanes, gathers every 5 steps or once at the end.
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_FORCE_COMM=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python bench.py --gpus 1 --cpu-seconds 0 --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1])
print('%-58s %.4g calls/s  %.2f us/step  every %d lanes %s' % ('$*', d['value'], d['ms_per_step'] * 1e3, d['config']['gather_every_steps'], d['config']['lanes']))"; }
for rep in 1 2; do
for gather in chi2 full; do
run --steps 20 --warmup 5 --gather $gather
run --steps 20 --warmup 5 --gather $gather --lanes 4
run --steps 20 --warmup 5 --gather $gather --gather-every 20 --lanes 4
run --steps 20 --warmup 5 --gather $gather --gather-every 10 --lanes 4
done
done
