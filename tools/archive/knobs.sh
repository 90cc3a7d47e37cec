#!/bin/bash
# Developer: default bench under a list of environment settings, one per argument.
cd "$GRAFT_REPO_ROOT" || exit 1
for setting in "$@"; do
  env $setting python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s calls/s %.4g ms/step %.4f contract %.4f' % ('$setting', d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms']))"
done
