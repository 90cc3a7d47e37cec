#!/bin/bash
# Developer PMC passes (serialized kernels) for the contraction kernel's bottleneck.
cd "$GRAFT_REPO_ROOT" || exit 1
# (the TC_* knobs are read by developer builds only: tools/build_dev.sh)
export TMPDIR=/tmp TC_LANES=1 TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
rm -rf gpurun_out/pmc_*
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_COEXEC_CYCLES" \
           "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 > gpurun_out/pmc_$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmc_1 gpurun_out/pmc_2 gpurun_out/pmc_3 gpurun_out/pmc_4 gpurun_out/pmc_5 | grep contract
