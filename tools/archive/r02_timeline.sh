#!/bin/bash
# Developer diagnosis: kernel timelines of the pipelined step (rocprofv3 --kernel-trace, rocpd
# database) for option sets given as arguments, e.g. "occ_per_cu=1" "occ_splits=16".
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
i=0
for opts in "$@"; do
  i=$((i + 1))
  args=""
  for o in $opts; do [ "$o" != "default" ] && args="$args --option $o"; done
  python bench.py --cpu-seconds 0 --other-configs 0 --steps 6000 --warmup 300 $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s ms/step %.4f contract %.4f (overlapped %.4f)' % ('$opts', d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['overlapped_launch_ms']))"
  rocprofv3 --kernel-trace -d gpurun_out/tl_$i -o t -- python3 bench.py --cpu-seconds 0 --other-configs 0 --steps 600 --warmup 100 $args > /dev/null 2>&1
  python tools/archive/timeline_db.py gpurun_out/tl_$i/t_results.db ${TL_SAMPLE:-0}
  rm -rf gpurun_out/tl_$i
done
