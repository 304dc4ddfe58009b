#!/bin/bash
# device assembly only (no host side, no alt build): registers / spills of every kernel in ~1.5 min.  usage: tools/asm_only.sh [out.s]
root="$(cd "$(dirname "$0")/.." && pwd)"
out=${1:-/tmp/mia_dev.s}
snap=/tmp/mia_asm_snap; rm -rf $snap; mkdir -p $snap/mapping-iterative-assembler_amd; cp -r "$root/include" $snap/; cp -r "$root/mapping-iterative-assembler_amd/csrc" $snap/mapping-iterative-assembler_amd/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function --cuda-device-only -S -o "$out" $snap/mapping-iterative-assembler_amd/csrc/mia_hip.hip 2>&1 | grep -v "loop not unrolled\|warnings generated" | head -20
for k in "$@"; do :; done
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1))
    v, vs, ss, sc = g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size")
    if vs or sc or "bxl" in name or "k_bx_plan" in name:
        short = re.sub(r"^_ZN3mia\d+", "", name)[:60]
        print("%-62s vgpr %3d vgpr-spills %3d sgpr-spills %3d scratch %4d" % (short, v, vs, ss, sc))
PY
