#!/bin/bash
# Copy the artefacts tools/profile_round.sh collected (gpurun_out/profile) into profiles/.
# Usage: tools/install_profiles.sh [round tag, default r02]
cd "$(dirname "$0")/.." || exit 1
R=${1:-r02}
P=gpurun_out/profile
cp $P/bench.json profiles/${R}_bench.json
cp $P/bench_driver_command.json profiles/${R}_bench_driver_command.json
cp $P/kernel_stats_lanes1.csv profiles/${R}_bench_kernel_stats_lanes1.csv
cp $P/kernel_stats_pipelined.csv profiles/${R}_bench_kernel_stats_pipelined.csv
cp $P/bench_under_rocprof_lanes1.log profiles/${R}_bench_under_rocprof_lanes1.log
cp $P/bench_under_rocprof_pipelined.log profiles/${R}_bench_under_rocprof_pipelined.log
(echo "# rocprofv3 --pmc passes (separate runs, --kernel-trace only): python3 bench.py --lanes 1 --steps 50 --warmup 5 --cpu-seconds 0 --other-configs 0 (tools/profile_round.sh)"
 echo "# FETCH_SIZE / WRITE_SIZE in KB per launch (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950); other counters raw"
 grep -v copyBuffer $P/pmc_summary.txt) > profiles/${R}_pmc_counters.txt
