#!/bin/bash
# Counter sets of one kernel for several builds / options (one rocprofv3 --pmc run per set):
#   KERNEL=predict_cross_fused ARGS="--only-config ds4 --cpu-seconds 0" SETS="a b c|d e" \
#   gpurun -- bash tools/r05_pmc.sh name=lib[,opt=val] ...
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_r05
mkdir -p $OUT
KERNEL=${KERNEL:-predict_fused}
ARGS=${ARGS:---steps 50 --warmup 5 --cpu-seconds 0 --detail 0}
SETS=${SETS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS|SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES}
NAME=${NAME:-r05_pmc}
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; opts=""
  if [ "$rest" != "$lib" ]; then for o in $(echo ${rest#*,} | tr ',' ' '); do opts="$opts --option $o"; done; fi
  if [ "$lib" = tree ]; then unset TABCORR_AMD_LIBRARY; else export TABCORR_AMD_LIBRARY=$GRAFT_REPO_ROOT/$lib; fi
  echo "== $name ($lib$opts)"
  echo "$SETS" | tr '|' '\n' | while read -r set; do
    rm -rf $OUT/pass
    rocprofv3 --pmc $set --output-format csv -d $OUT/pass -- \
      python3 bench.py $ARGS $opts > $OUT/log.txt 2>&1
    python3 tools/pmc_summary.py $OUT/pass | grep "$KERNEL"
  done
done > gpurun_out/$NAME.txt 2>&1
rm -rf $OUT
cat gpurun_out/$NAME.txt
