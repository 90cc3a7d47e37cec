#!/bin/bash
# Copy the artefacts tools/profile_round.sh collected (gpurun_out/profile) into profiles/.
# Usage: tools/install_profiles.sh [round tag, default r03]
cd "$(dirname "$0")/.." || exit 1
R=${1:-r06}
P=gpurun_out/profile
HEAD1="# rocprofv3 --pmc passes (separate runs, --kernel-trace only; tools/profile_round.sh): python3 bench.py"
HEAD2="# FETCH_SIZE / WRITE_SIZE in KB per launch (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950); other counters raw"
[ -f $P/bench.json ] && cp $P/bench.json profiles/${R}_bench.json
[ -f $P/bench_detail.json ] && cp $P/bench_detail.json profiles/${R}_bench_detail.json
[ -f $P/bench_driver_command.json ] && cp $P/bench_driver_command.json profiles/${R}_bench_driver_command.json
for mode in lanes1 pipelined fused_alone; do
  [ -f $P/kernel_stats_$mode.csv ] && cp $P/kernel_stats_$mode.csv profiles/${R}_bench_kernel_stats_$mode.csv
  [ -f $P/kernel_stats_$mode.json ] && cp $P/kernel_stats_$mode.json profiles/${R}_bench_under_rocprof_$mode.log
done
if [ -f $P/pmc_summary.txt ]; then
  (echo "$HEAD1 --steps 50 --warmup 5 --cpu-seconds 0 --detail 0"; echo "$HEAD2"
   grep -v copyBuffer $P/pmc_summary.txt) > profiles/${R}_pmc_counters.txt
fi
for tag in cfg3 cfg4 cfg5f32 cfg5f64 ds4 ds1 wp db; do
  for mode in lanes1 pipelined; do
    [ -f $P/kernel_stats_${tag}_$mode.csv ] && cp $P/kernel_stats_${tag}_$mode.csv profiles/${R}_${tag}_kernel_stats_$mode.csv
    [ -f $P/kernel_stats_${tag}_$mode.json ] && cp $P/kernel_stats_${tag}_$mode.json profiles/${R}_${tag}_under_rocprof_$mode.json
  done
  if [ -f $P/pmc_summary_${tag}.txt ]; then
    (echo "$HEAD1 --only-config $tag --cpu-seconds 0"; echo "$HEAD2"
     grep -v copyBuffer $P/pmc_summary_${tag}.txt) > profiles/${R}_pmc_counters_${tag}.txt
  fi
done
ls profiles/ | grep "^${R}_"
