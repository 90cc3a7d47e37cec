#!/bin/bash
# Build the library of another revision for A/B runs in ONE gpurun call (boxes of the pool differ
# by a few per cent): tools/ab_build.sh <git-rev> <name>  ->  tools/ab/libtabcorr_hip_<name>.so,
# selected with TABCORR_AMD_LIBRARY=tools/ab/libtabcorr_hip_<name>.so (the *.so is git-ignored
# and travels with gpurun).  <git-rev> = WORK builds the working tree's sources.
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/tabcorr_ab_XXXX)
if [ "$REV" = WORK ]; then
  mkdir -p $TMP/tabcorr_amd && cp -r $ROOT/tabcorr_amd/csrc $TMP/tabcorr_amd/ && cp -r $ROOT/include $TMP/
else
  git -C $ROOT archive $REV tabcorr_amd/csrc include | tar -x -C $TMP
fi
mkdir -p $ROOT/tools/ab
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function ${EXTRA_FLAGS}"
cd $TMP/tabcorr_amd/csrc
for f in $(ls *.hip *.cpp); do
  # (inst_single.hip: see its header)
  UNIT_FLAGS=$([ $f = inst_single.hip ] && echo -ffp-contract=on || true)
  /opt/rocm/bin/hipcc $FLAGS $UNIT_FLAGS -c $f -o $TMP/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -Wl,-z,now -Wl,-rpath,/opt/rocm/lib \
  -o $ROOT/tools/ab/libtabcorr_hip_$NAME.so $TMP/*.o
rm -rf $TMP
echo $ROOT/tools/ab/libtabcorr_hip_$NAME.so
