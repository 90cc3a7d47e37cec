#!/bin/bash
# Build build/ab/dev.so: the library with the developer knobs (TC_SKIP_OCC, TC_SKIP_FINALIZE).
cd "$(dirname "$0")/.." || exit 1
mkdir -p build/ab/obj
for f in $(cd tabcorr_amd/csrc && ls *.hip *.cpp); do
  # (inst_single.hip: see its header)
  UNIT_FLAGS=$([ $f = inst_single.hip ] && echo -ffp-contract=on)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -DTC_DEVELOPER_KNOBS $UNIT_FLAGS -c tabcorr_amd/csrc/$f -o build/ab/obj/${f%.*}.o 2>build/ab/obj/${f%.*}.err || echo "build_dev.sh: $f failed (build/ab/obj/${f%.*}.err)" &
done
wait
hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -Wl,-z,now -Wl,-rpath,/opt/rocm/lib -o build/ab/dev.so build/ab/obj/*.o
