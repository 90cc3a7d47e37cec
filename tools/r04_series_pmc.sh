#!/bin/bash
# SQ_INSTS_VALU / matrix-pipe counters of the headline kernel with the expansions on and off
# (one rocprofv3 --pmc run each): gpurun -- bash tools/r04_series_pmc.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_series
mkdir -p $OUT
for on in 1 0; do
  rm -rf $OUT/pass
  rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_SALU \
    --output-format csv -d $OUT/pass -- \
    python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --other-configs 0 --option series=$on \
    > $OUT/log_$on.txt 2>&1
  echo "series=$on"
  python3 tools/pmc_summary.py $OUT/pass | grep predict_fused
done > gpurun_out/r04_series_pmc.txt 2>&1
rm -rf $OUT
cat gpurun_out/r04_series_pmc.txt
