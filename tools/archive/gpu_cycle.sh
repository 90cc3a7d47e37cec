#!/bin/bash
# Developer loop on a GPU box: parity suite, default bench, serialized per-kernel stats.
# Usage (through gpurun): bash tools/archive/gpu_cycle.sh [skip-tests]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "$1" != "skip-tests" ]; then
  timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
  grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | tail -5
fi
python bench.py --cpu-seconds 0 > gpurun_out/bench.json 2> gpurun_out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("calls/s %.4g  ms/step %.4f  contract %.4f ms  parity %.2g" % (
    d["value"], d["ms_per_step"], d["roofline"]["mean_launch_ms"],
    d["parity_max_rel_vs_oracle"]))
PY
export TMPDIR=/tmp
rm -rf gpurun_out/prof_l1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l1 -- \
  python3 bench.py --lanes 1 --steps 300 --warmup 30 --cpu-seconds 0 > gpurun_out/prof_l1.log 2>&1
cat gpurun_out/prof_l1/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110 | head -6
