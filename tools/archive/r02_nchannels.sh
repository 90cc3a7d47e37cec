#!/bin/bash
# Developer A/B: RCCL channel count (workgroups of its copy kernels) against the gather's
# interference with the pipeline (one rank, communicator forced on).
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_FORCE_COMM=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python bench.py --gpus 1 --cpu-seconds 0 --other-configs 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1])
print('NCHANNELS=%-4s %-44s %.4g calls/s  %.2f us/step  every %d lanes %s' % ('$NCCL_MAX_NCHANNELS', '$*', d['value'], d['ms_per_step'] * 1e3, d['config']['gather_every_steps'], d['config'].get('lanes')))"; }
for n in "" 1 2 4 8; do
  if [ -z "$n" ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS; else export NCCL_MAX_NCHANNELS=$n NCCL_MIN_NCHANNELS=1; fi
  run --steps 20 --warmup 5
  run --steps 20 --warmup 5 --lanes 4
  run --steps 4000 --warmup 200 --gather-every 8 --lanes 4
  run --steps 4000 --warmup 200
done
