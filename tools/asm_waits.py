#!/usr/bin/env python3
"""Per kernel of the gfx950 assembly: loads (global / LDS / scratch) that are waited for within four instructions of their issue --
the pattern of a load whose latency nothing hides.  Round 4 found the band DPs' eight LDS reads per row, the plan's table
walks, the traceback's trace words and the tally's records "a pass ahead" all compiled that way.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o /tmp/dev.s csrc/mia_hip.hip
       python3 tools/asm_waits.py /tmp/dev.s [kernel-name-substring]"""
import re
import subprocess
import sys

LOAD = re.compile(r"\b(global_load|ds_read|scratch_load)")
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
stats, kern, cur = {}, None, []


def flush():
    if not (kern and cur):
        return
    n_imm = n_ld = 0
    for i, l in enumerate(cur):
        if not LOAD.search(l):
            continue
        n_ld += 1
        for k in range(1, 5):
            if i + k >= len(cur):
                break
            m = cur[i + k]
            if "s_waitcnt" in m and ("vmcnt(0)" in m or "lgkmcnt(0)" in m):
                n_imm += 1
                break
            if LOAD.search(m):
                break
    info = " ".join(x.strip("; ") for x in cur if re.search(r"; (NumVgprs|ScratchSize|Occupancy):", x))
    stats[kern] = (len(cur), n_ld, n_imm, info)


for l in lines:
    m = re.match(r"^(_ZN3mia\w+):", l)
    if m:
        flush()
        kern, cur = m.group(1), []
    elif kern is not None:
        cur.append(l)
        if "; Occupancy" in l:
            flush()
            kern, cur = None, []
for k, (n, ld, imm, info) in sorted(stats.items(), key=lambda x: -x[1][2]):
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.split("(")[0]
    if want in name:
        print(f"{imm:5d} waited for on the spot / {ld:5d} loads / {n:6d} lines  {name[:60]:60s} {info}")
