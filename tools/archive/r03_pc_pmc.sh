#!/bin/bash
# PMC passes of the pair counter at tabulation scale (tools/archive/paircount_bench.py).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pc_pmc
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "WRITE_SIZE FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- \
    python3 tools/archive/paircount_bench.py --no-oracle > $OUT/pmc_$i.log 2>&1
done
python3 - <<PY
import csv, glob
from collections import defaultdict
best = defaultdict(float)
for path in glob.glob('$OUT/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        key = (row['Kernel_Name'][:44], row['Counter_Name'])
        best[key] = max(best[key], float(row['Counter_Value']))    # the 10^6-point launch
for (kernel, counter), value in sorted(best.items()):
    if 'pair_count' in kernel:
        print('%-46s %-24s %.4g' % (kernel, counter, value))
PY
rm -rf $OUT/pmc_[0-9]
