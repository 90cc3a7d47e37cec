#!/bin/bash
# Developer A/B on one box: alternate two option sets.
cd "$GRAFT_REPO_ROOT" || exit 1
A="$1"; B="$2"
for i in 1 2 3; do
for opt in "$A" "$B"; do
python bench.py --cpu-seconds 0 --other-configs 0 --steps 8000 --warmup 500 $opt | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-28s %.4g calls/s  %.2f us/step  kernel %.2f us (frac %.3f)  overlapped %.2f  serial step %.2f' % ('$opt',
    d['value'], d['ms_per_step'] * 1e3, r['mean_launch_ms'] * 1e3, r['frac'], r['overlapped_launch_ms'] * 1e3, r['serialised_step_ms'] * 1e3))"
done; done
