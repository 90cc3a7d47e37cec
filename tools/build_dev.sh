#!/bin/bash
# Build build/ab/dev.so: the library with the developer knobs (TC_SKIP_OCC, TC_SKIP_FINALIZE).
cd "$(dirname "$0")/.." || exit 1
mkdir -p build/ab/obj
for f in launch.hip paircount.hip table.cpp interp.cpp comm.cpp runtime.cpp hostmath.cpp; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -DTC_DEVELOPER_KNOBS -c tabcorr_amd/csrc/$f -o build/ab/obj/${f%.*}.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -Wl,-z,now -Wl,-rpath,/opt/rocm/lib -o build/ab/dev.so build/ab/obj/*.o
