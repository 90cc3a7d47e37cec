#!/bin/bash
# Collect the round's profile artefacts on a GPU box into gpurun_out/profile/ (copied to
# profiles/ afterwards by tools/install_profiles.sh): bench line, rocprofv3 kernel stats
# (kernels serialised with --lanes 1, and pipelined), PMC passes (each in its own run, with
# --kernel-trace only).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/profile
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# the driver's own command (short timed region), three times
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $OUT/bench_driver_command.json
done
FAST="--cpu-seconds 0 --other-configs 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lanes1 -- \
  python3 bench.py --lanes 1 --steps 2000 --warmup 200 $FAST > $OUT/bench_under_rocprof_lanes1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipelined -- \
  python3 bench.py --steps 2000 --warmup 200 $FAST > $OUT/bench_under_rocprof_pipelined.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "TA_TA_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- \
    python3 bench.py --lanes 1 --steps 50 --warmup 5 $FAST > $OUT/pmc_$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $OUT/pmc_4 $OUT/pmc_5 $OUT/pmc_6 > $OUT/pmc_summary.txt
cp $OUT/lanes1/*/*kernel_stats.csv $OUT/kernel_stats_lanes1.csv
cp $OUT/pipelined/*/*kernel_stats.csv $OUT/kernel_stats_pipelined.csv
grep -h '"metric"' $OUT/bench_under_rocprof_lanes1.log > $OUT/l1.json; mv $OUT/l1.json $OUT/bench_under_rocprof_lanes1.log
grep -h '"metric"' $OUT/bench_under_rocprof_pipelined.log > $OUT/p.json; mv $OUT/p.json $OUT/bench_under_rocprof_pipelined.log
rm -rf $OUT/lanes1 $OUT/pipelined $OUT/pmc_[0-9] $OUT/pmc_[0-9].log
cat $OUT/bench.json; cut -c1-140 $OUT/kernel_stats_lanes1.csv | head -6; cat $OUT/pmc_summary.txt
