#!/bin/bash
# configs[4] host-to-host rates (fresh and kept result arrays) out of bench.py's own leg.
# On the GPU box: bash tools/r05_cfg5_host.sh  ->  gpurun_out/r05_cfg5_host.log
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for tag in cfg5f32 cfg5f64; do
  python bench.py --only-config $tag --cpu-seconds 0 > gpurun_out/r05_$tag.log 2>&1 || exit 1
  python - "$tag" <<'PY'
import json, sys
def find(o, path=''):
    if isinstance(o, dict):
        for k, v in o.items():
            if 'host_to_host' in k:
                print('%s %s %s: %.4g' % (sys.argv[1], path, k, v), flush=True)
            find(v, path + '/' + k)
find(json.load(open('bench_detail.json')))
PY
done | tee gpurun_out/r05_cfg5_host.log
