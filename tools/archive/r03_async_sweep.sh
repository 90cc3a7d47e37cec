#!/bin/bash
# A/B of the asynchronous host path on one box: lanes, copy engines against direct access.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_async_sweep.jsonl
: > $O
run() { python3 $R/tools/archive/r03_async.py --device-step 0 --seconds 0.4 "$@" >> $O 2>> $R/gpurun_out/r03_async_sweep.err || exit 1; }
run --depths 4,6 --option async_direct_in=1 --option async_direct_out=2
run --depths 4,6,8 --option async_direct_in=1 --option async_direct_out=2 --option async_out_stream=1
run --lanes 3 --depths 3,6 --option async_direct_in=1 --option async_direct_out=2 --option async_out_stream=1
run --depths 4,6,8 --option async_direct_in=1 --option async_out_stream=1
run --depths 4,8 --option async_out_stream=1
run --lanes 3 --depths 6 --option async_out_stream=1
python3 - <<PY
import json
for line in open('$O'):
    d = json.loads(line)
    head = 'lanes=%s adj=%s %s' % (d['lanes'], d['adjacent'], ' '.join(d['options']))
    cells = ['%s %.1f' % (k, v['us_per_call']) for k, v in d.items() if isinstance(v, dict)]
    print(head, '|', ' | '.join(cells))
PY
