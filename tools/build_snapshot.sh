#!/bin/bash
# Development build from a frozen copy of the sources (the tree may be edited while hipcc runs: a file that changes under the
# compiler's lexer crashed it once).  Same commands as mapping-iterative-assembler_amd/build.sh; the products are copied back.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
snap=/tmp/mia_build_snap
rm -rf "$snap"; mkdir -p "$snap/mapping-iterative-assembler_amd"
cp -r "$root/include" "$snap/include"
cp -r "$root/mapping-iterative-assembler_amd/csrc" "$root/mapping-iterative-assembler_amd/host" "$root/mapping-iterative-assembler_amd/build.sh" "$snap/mapping-iterative-assembler_amd/"
bash "$snap/mapping-iterative-assembler_amd/build.sh" "$@"
pkg="$root/mapping-iterative-assembler_amd"
mkdir -p "$pkg/build"
cp "$snap/mapping-iterative-assembler_amd/"{libmia_hip.so,libmia_hip_alt.so,mia_hip,ma_hip,ccheck_hip} "$pkg/"
cp "$snap/mapping-iterative-assembler_amd/build/"*.s "$pkg/build/" 2>/dev/null || true
echo "snapshot build done"
