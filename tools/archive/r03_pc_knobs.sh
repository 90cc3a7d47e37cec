#!/bin/bash
# Labelled pair count: label-block width, unroll and unit count (developer build).
cd /tmp && export TMPDIR=/tmp
export TABCORR_AMD_LIBRARY=$GRAFT_REPO_ROOT/build/ab/dev.so
for knobs in "TC_PAIR_BLOCK2=8 TC_PAIR_UNROLL=4" "TC_PAIR_BLOCK2=8 TC_PAIR_UNROLL=8" "TC_PAIR_BLOCK2=4 TC_PAIR_UNROLL=4" "TC_PAIR_BLOCK2=4 TC_PAIR_UNROLL=8" "TC_PAIR_BLOCK2=2 TC_PAIR_UNROLL=8" "TC_PAIR_BLOCK2=8 TC_PAIR_UNROLL=8 TC_PAIR_UNITS=16384" "TC_PAIR_BLOCK2=16 TC_PAIR_UNROLL=8"; do
  echo "== $knobs"
  env $knobs python3 $GRAFT_REPO_ROOT/tools/archive/paircount_bench.py --no-oracle 2>&1 | grep "bin pairs"
done
