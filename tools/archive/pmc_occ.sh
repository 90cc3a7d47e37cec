#!/bin/bash
# Developer PMC passes (serialized kernels) for the occupation kernel's bottleneck.
cd "$GRAFT_REPO_ROOT" || exit 1
# (the TC_* knobs are read by developer builds only: tools/build_dev.sh)
export TMPDIR=/tmp TC_LANES=1 TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
rm -rf gpurun_out/pmc_*
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 > gpurun_out/pmc_$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmc_1 gpurun_out/pmc_2 gpurun_out/pmc_3 gpurun_out/pmc_4 | grep -v copyBuffer
