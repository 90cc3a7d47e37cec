#!/bin/bash
# Developer PMC passes for the float32 contraction (cfg5).
cd "$GRAFT_REPO_ROOT" || exit 1
# (the TC_* knobs are read by developer builds only: tools/build_dev.sh)
export TMPDIR=/tmp TC_LANES=1 TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
rm -rf gpurun_out/pmc_*
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "TA_TA_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 tools/archive/sweep.py cfg5one > gpurun_out/pmc_$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmc_1 gpurun_out/pmc_2 gpurun_out/pmc_3 | grep -E "contract"
