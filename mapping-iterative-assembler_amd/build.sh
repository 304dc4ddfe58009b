#!/bin/bash
# Builds libmia_hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$here/build"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function \
  -save-temps=obj -o "$here/build/libmia_hip.so" "$here/csrc/mia_hip.hip" "$here/csrc/mia_comm.hip" "$@"
cp "$here/build/libmia_hip.so" "$here/libmia_hip.so"
# host program (mia command line on top of the C ABI)
g++ -O2 -std=c++17 -Wall -pthread -o "$here/mia_hip" "$here/host/mia_main.cpp" -L"$here" -lmia_hip -Wl,-rpath,'$ORIGIN'
g++ -O2 -std=c++17 -Wall -pthread -o "$here/ma_hip" "$here/host/ma_main.cpp" -L"$here" -lmia_hip -Wl,-rpath,'$ORIGIN'
g++ -O2 -std=c++17 -Wall -pthread -o "$here/ccheck_hip" "$here/host/ccheck_main.cpp" -L"$here" -lmia_hip -Wl,-rpath,'$ORIGIN'
