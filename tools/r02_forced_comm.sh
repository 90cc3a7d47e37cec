#!/bin/bash
# Developer loop: bench.py with the communicator forced on (one rank: RCCL gather to itself),
# short and long timed regions, gather block sizes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_FORCE_COMM=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python bench.py --gpus 1 --cpu-seconds 0 --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1])
print('%-52s %.4g calls/s  %.2f us/step  gather %s every %d  breakdown %s' % ('$*', d['value'], d['ms_per_step'] * 1e3, d['config']['gather'], d['config']['gather_every_steps'], {k: round(v) for k, v in d['timed_region_breakdown_us'].items()}))"; }
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --gather-every 32
run --steps 20 --warmup 5 --gather-every 2
run --steps 20 --warmup 5 --gather chi2
run --steps 4000 --warmup 200
run --steps 4000 --warmup 200 --gather chi2
