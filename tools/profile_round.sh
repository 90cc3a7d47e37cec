#!/bin/bash
# Collect the round's profile artefacts on a GPU box into gpurun_out/profile/ (copied to
# profiles/ afterwards by tools/install_profiles.sh): bench line, rocprofv3 kernel stats
# (kernels serialised with --lanes 1, and pipelined), PMC passes (each in its own run, with
# --kernel-trace only) -- for the headline configuration and, with the same recipe, for
# BASELINE configs[2], [3], [4] float32 / float64 (bench.py --only-config TAG).
# The program always follows `--` directly (python3 bench.py ...).
#   gpurun --timeout 1200 -- bash tools/profile_round.sh [main|configs|all]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
WHAT=${1:-all}
ROUND=${ROUND:-r06}
OUT=gpurun_out/profile
mkdir -p $OUT
FAST="--cpu-seconds 0 --detail 0"

pmc_passes() {   # $1 = output stem, rest = bench arguments
  local stem=$1; shift
  local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES" \
             "TCC_HIT_sum TCC_MISS_sum" \
             "TA_TA_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
             "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
    i=$((i+1))
    [ "$PMC_SHORT" = 1 ] && [ $i -gt 3 ] && break
    rm -rf $OUT/pmc_$i
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- \
      python3 bench.py "$@" > $OUT/pmc_$i.log 2>&1
    echo "pmc pass $i of $stem done"
  done
  python3 tools/pmc_summary.py $OUT/pmc_[0-9] > $OUT/${stem}.txt
  rm -rf $OUT/pmc_[0-9] $OUT/pmc_[0-9].log
  # into profiles/ of THIS copy of the tree as well (as tools/install_profiles.sh will at home):
  # the bench runs that follow quote `traffic` from these passes, not from last round's
  local tag=${stem#pmc_summary}
  (echo "# rocprofv3 --pmc passes (separate runs, --kernel-trace only; tools/profile_round.sh): python3 bench.py $*"
   echo "# FETCH_SIZE / WRITE_SIZE in KB per launch (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950); other counters raw"
   grep -v copyBuffer $OUT/${stem}.txt) > profiles/${ROUND}_pmc_counters${tag}.txt
}

kernel_stats() {   # $1 = output stem, rest = bench arguments
  local stem=$1; shift
  rm -rf $OUT/trace
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 bench.py "$@" > $OUT/${stem}.log 2>&1
  cp $OUT/trace/*/*kernel_stats.csv $OUT/${stem}.csv
  grep -h -E '^\{' $OUT/${stem}.log | tail -1 > $OUT/${stem}.json
  rm -rf $OUT/trace $OUT/${stem}.log
  echo "kernel stats $stem done"
}

if [ $WHAT = main ] || [ $WHAT = all ]; then
  # (default lanes: the profiler serialises the dispatches itself; the timed region then runs
  # predict_fused_kernel, bench.py's serialised pass the three kernels -- both are counted)
  pmc_passes pmc_summary --steps 50 --warmup 5 $FAST
  # (the last line of stdout is the record; everything else sits in the sidecar)
  python bench.py > $OUT/bench.log 2> $OUT/bench.err
  tail -1 $OUT/bench.log > $OUT/bench.json
  cp bench_detail.json $OUT/bench_detail.json
  echo "default bench done"
  # the driver's own command (short timed region), three times
  : > $OUT/bench_driver_command.json
  for i in 1 2 3; do
    python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $OUT/bench_driver_command.json
  done
  # --lanes 1: every call alone on its lane = the three-kernel path, kernels serialised;
  # default lanes: predict_fused_kernel, one launch per step, four launches overlapping;
  # fused=2 with one lane: that kernel alone on the chip
  kernel_stats kernel_stats_lanes1 --lanes 1 --steps 2000 --warmup 200 $FAST
  kernel_stats kernel_stats_pipelined --steps 2000 --warmup 200 $FAST
  kernel_stats kernel_stats_fused_alone --lanes 1 --option fused=2 --steps 2000 --warmup 200 $FAST
fi
if [ $WHAT = configs ] || [ $WHAT = all ]; then
  # (TAGS="cfg3 ds4" ... selects; the PMC passes run with the default lanes so that the
  # one-launch forms are what the pipelined calls take -- the profiler serialises the
  # dispatches itself --, and bench.py's own serialised pass adds the three kernels)
  # (sync_chunks=-1: the host-call leg of these runs launches the same kernels for CHUNKS of the
  # batch otherwise, and the per-launch means below would mix two launch sizes)
  SERIAL="--option sync_chunks=-1"
  for tag in ${TAGS:-cfg3 cfg4 cfg5f32 cfg5f64 ds4 ds1 wp db}; do
    PMC_SHORT=${PMC_SHORT:-1} pmc_passes pmc_summary_${tag} --only-config $tag --cpu-seconds 0 $SERIAL
    kernel_stats kernel_stats_${tag}_lanes1 --only-config $tag --lanes 1 --cpu-seconds 0 $SERIAL
    kernel_stats kernel_stats_${tag}_pipelined --only-config $tag --cpu-seconds 0 $SERIAL
  done
fi
ls -la $OUT
