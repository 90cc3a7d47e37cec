#!/bin/bash
# PMC passes for predict_fused_kernel (bench.py's timed region with default lanes; the profiler
# serialises the dispatches).  gpurun -- bash tools/archive/r03_fused_pmc.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/fused_pmc
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "TA_TA_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- \
    python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --other-configs 0 "$@" > $OUT/pmc_$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_[0-9] | grep -i "fused\|quad\|occ_\|finalize" > $OUT/summary.txt
rm -rf $OUT/pmc_[0-9] $OUT/pmc_[0-9].log
cat $OUT/summary.txt
