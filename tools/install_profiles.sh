#!/bin/bash
# Copy the artefacts tools/profile_round.sh collected (gpurun_out/profile) into profiles/.
cd "$(dirname "$0")/.." || exit 1
P=gpurun_out/profile
cp $P/bench.json profiles/r01_bench.json
cp $P/kernel_stats_lanes1.csv profiles/r01_bench_kernel_stats_lanes1.csv
cp $P/kernel_stats_pipelined.csv profiles/r01_bench_kernel_stats_pipelined.csv
cp $P/bench_under_rocprof_lanes1.log profiles/r01_bench_under_rocprof_lanes1.log
cp $P/bench_under_rocprof_pipelined.log profiles/r01_bench_under_rocprof_pipelined.log
(echo "# rocprofv3 --pmc passes (separate runs), TC_LANES=1 python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 (tools/profile_round.sh)"
 echo "# FETCH_SIZE / WRITE_SIZE in KB per launch (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950); other counters raw"
 grep -v copyBuffer $P/pmc_summary.txt) > profiles/r01_pmc_counters.txt
