#!/bin/bash
# Round 6: one PMC pass (vector / matrix instruction counts) of a configuration of other_configs,
# e.g.  gpurun -- bash tools/r06_pmc_quick.sh wp
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-wp}; shift
OUT=gpurun_out/pmc_quick_$TAG
rm -rf $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU \
  --output-format csv -d $OUT -- python3 bench.py --only-config $TAG --cpu-seconds 0 \
  --option sync_chunks=-1 "$@" > $OUT.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -v copyBuffer > gpurun_out/pmc_quick_$TAG.txt
rm -rf $OUT
cat gpurun_out/pmc_quick_$TAG.txt
