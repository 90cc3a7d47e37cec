#!/bin/bash
# PMC passes (each its own run, --pmc only) of one shape of tools/r04_grouped.py at one batch
# size: gpurun -- bash tools/r04_pmc.sh <shape> [stem]      (R04_* environment: see that script)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
SHAPE=${1:-ds}
STEM=${2:-r04_pmc_$SHAPE}
OUT=gpurun_out/pmc
mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $OUT/pass_$i
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass_$i -- \
    python3 tools/r04_grouped.py $SHAPE > $OUT/pass_$i.log 2>&1 || { tail -5 $OUT/pass_$i.log; }
  echo "pmc pass $i done"
done
python3 tools/pmc_summary.py $OUT/pass_[0-9] > gpurun_out/$STEM.txt
rm -rf $OUT
cat gpurun_out/$STEM.txt
