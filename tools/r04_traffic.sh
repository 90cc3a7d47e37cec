#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one other_configs entry with an option: 
#   gpurun -- bash tools/r04_traffic.sh <tag> <option=value>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; OPT=$2
OUT=gpurun_out/pmc_traffic
mkdir -p $OUT
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pass
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass -- \
    python3 bench.py --only-config $TAG --cpu-seconds 0 --option $OPT > $OUT/log.txt 2>&1
  python3 tools/pmc_summary.py $OUT/pass | grep -v copyBuffer
  grep -o '"us_per_step": [0-9.]*' $OUT/log.txt | head -1
done
rm -rf $OUT
