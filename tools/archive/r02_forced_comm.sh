#!/bin/bash
# Developer loop: bench.py with the communicator forced on (one rank: RCCL gather to itself),
# short and long timed regions, gather block sizes, three or four lanes (the communicator's
# stream is a fifth stream on the runtime's four hardware queues).
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_FORCE_COMM=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python bench.py --gpus 1 --cpu-seconds 0 --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1])
print('%-52s %.4g calls/s  %.2f us/step  gather %s every %d' % ('$*', d['value'], d['ms_per_step'] * 1e3, d['config']['gather'], d['config']['gather_every_steps']))"; }
for lanes in 4 3; do
run --steps 20 --warmup 5 --lanes $lanes
run --steps 20 --warmup 5 --gather-every 32 --lanes $lanes
run --steps 4000 --warmup 200 --lanes $lanes
run --steps 4000 --warmup 200 --gather-every 8 --lanes $lanes
run --steps 4000 --warmup 200 --gather chi2 --lanes $lanes
done
