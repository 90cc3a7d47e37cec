#!/bin/bash
# predict_fused_kernel: wave priorities of its phases (device-resident step of bench.py,
# BASELINE configs[1]).  Usage (GPU box): bash tools/archive/r03_fused_sweep.sh > gpurun_out/fused_sweep.txt
run() {
  local label="$1"; shift
  python bench.py --cpu-seconds 0 --other-configs 0 "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %8.2f us/step  async %.1f  chi2 %.1f' % ('$label', d['ms_per_step'] * 1e3, d['host_to_host_pipelined']['us_per_call'], d['host_to_host_chi2']['us_per_call']))"
}
for occ in 0 1 2 3; do for con in 0 1 2; do
  run "occ=$occ con=$con out=3" --option prio_fused_occ=$occ --option prio_fused=$con
done; done
run "occ=2 con=1 out=1" --option prio_fused_out=1
run "fused=0" --option fused=0
