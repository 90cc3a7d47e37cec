#!/bin/bash
# Developer loop: quick bench (no CPU baselines, no other configs), a few times.
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
python bench.py --cpu-seconds 0 --other-configs 0 "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%.4g calls/s  %.2f us/step  kernel %.2f us (frac %.3f)  overlapped %.2f  serial step %.2f  h2h %.3g  unbatched %s' % (
    d['value'], d['ms_per_step'] * 1e3, r['mean_launch_ms'] * 1e3, r['frac'], r['overlapped_launch_ms'] * 1e3,
    r['serialised_step_ms'] * 1e3, d.get('host_to_host', {}).get('value', 0), d.get('unbatched_us')))"
done
