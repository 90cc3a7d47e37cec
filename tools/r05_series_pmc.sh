#!/bin/bash
# SQ_INSTS_VALU / SALU / wait counters of the headline kernel for several library builds and
# options (one rocprofv3 --pmc run each): gpurun -- bash tools/r05_series_pmc.sh "name=lib[,opt=val]" ...
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_series
mkdir -p $OUT
KERNEL=${KERNEL:-predict_fused}
ARGS=${ARGS:---steps 50 --warmup 5 --cpu-seconds 0 --detail 0}
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; opts=""
  if [ "$rest" != "$lib" ]; then for o in $(echo ${rest#*,} | tr ',' ' '); do opts="$opts --option $o"; done; fi
  if [ "$lib" = tree ]; then unset TABCORR_AMD_LIBRARY; else export TABCORR_AMD_LIBRARY=$GRAFT_REPO_ROOT/$lib; fi
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    rm -rf $OUT/pass
    rocprofv3 --pmc $set --output-format csv -d $OUT/pass -- \
      python3 bench.py $ARGS $opts > $OUT/log.txt 2>&1
    echo "== $name"
    python3 tools/pmc_summary.py $OUT/pass | grep "$KERNEL"
  done
done > gpurun_out/r05_series_pmc.txt 2>&1
rm -rf $OUT
cat gpurun_out/r05_series_pmc.txt
