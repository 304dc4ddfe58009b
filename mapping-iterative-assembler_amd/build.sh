#!/bin/bash
# Builds libmia_hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$here/build"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function \
  -save-temps=obj -o "$here/build/libmia_hip.so" "$here/csrc/mia_hip.hip" "$here/csrc/mia_comm.hip" "$@"
cp "$here/build/libmia_hip.so" "$here/libmia_hip.so"
# the same library with every alternative route and debug switch compiled in (-DMIA_HIP_ALT_PATHS): what the differential tests and
# the profiling tools load (the Python binding picks it for a context made while a non-release MIA_HIP_* variable is set)
mkdir -p "$here/build/alt"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -DMIA_HIP_ALT_PATHS \
  -o "$here/build/alt/libmia_hip_alt.so" "$here/csrc/mia_hip.hip" "$here/csrc/mia_comm.hip" "$@" &
alt_pid=$!
# host program (mia command line on top of the C ABI)
g++ -O2 -std=c++17 -Wall -pthread -o "$here/mia_hip" "$here/host/mia_main.cpp" -L"$here" -lmia_hip -Wl,-rpath,'$ORIGIN'
g++ -O2 -std=c++17 -Wall -pthread -o "$here/ma_hip" "$here/host/ma_main.cpp" -L"$here" -lmia_hip -Wl,-rpath,'$ORIGIN'
g++ -O2 -std=c++17 -Wall -pthread -o "$here/ccheck_hip" "$here/host/ccheck_main.cpp" -L"$here" -lmia_hip -Wl,-rpath,'$ORIGIN'
wait $alt_pid
cp "$here/build/alt/libmia_hip_alt.so" "$here/libmia_hip_alt.so"
